"""`stable-diffusion-prompt-processor` (threestudio/models/prompt_processors/base.py:170-520 +
stable_diffusion_prompt_processor.py): view-dependent prompt strings, the text-embedding cache, and the PromptUtils object the
guidance consumes.  The cache format is the reference's — `.threestudio_cache/text_embeddings/<md5(f"{model}-{prompt}")>.pt`
holding one [77, 1024] tensor per prompt (base.py:19-23, 349-420) — so embeddings computed by a ScaleDreamer installation are
picked up unchanged.  No text encoder ships with this repository (it runs once, off the step path): a missing cache entry is
computed by `encode_fn(prompts) -> [n, 77, 1024]` when one is given, otherwise it is the reference's FileNotFoundError.
"""
from __future__ import annotations

import hashlib
import json
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch

from .base import BaseObject
from .guidance import PromptUtils
from .registry import info, register


def hash_prompt(model: str, prompt: str) -> str:
    return hashlib.md5(f"{model}-{prompt}".encode()).hexdigest()


_DIRECTIONS = ("side", "front", "back", "overhead")   # order of text_embeddings_vd (base.py:231-300)


@register("stable-diffusion-prompt-processor")
class StableDiffusionPromptProcessor(BaseObject):
    @dataclass
    class Config(BaseObject.Config):
        prompt: str = "a hamburger"
        prompt_front: Optional[str] = None
        prompt_side: Optional[str] = None
        prompt_back: Optional[str] = None
        prompt_overhead: Optional[str] = None
        negative_prompt: str = ""
        pretrained_model_name_or_path: str = "runwayml/stable-diffusion-v1-5"
        overhead_threshold: float = 60.0
        front_threshold: float = 45.0
        back_threshold: float = 45.0
        view_dependent_prompt_front: bool = False
        use_cache: bool = True
        spawn: bool = True
        use_perp_neg: bool = False
        perp_neg_f_sb: Tuple[float, float, float] = (1, 0.5, -0.606)
        perp_neg_f_fsb: Tuple[float, float, float] = (1, 0.5, +0.967)
        perp_neg_f_fs: Tuple[float, float, float] = (4, 0.5, -2.426)
        perp_neg_f_sf: Tuple[float, float, float] = (4, 0.5, -2.426)
        use_prompt_debiasing: bool = False
        pretrained_model_name_or_path_prompt_debiasing: str = "bert-base-uncased"
        prompt_debiasing_mask_ids: Optional[List[int]] = None

    cfg: Config

    def configure(self, cache_dir: str = ".threestudio_cache/text_embeddings", encode_fn: Optional[Callable] = None,
                  prompt_library_path: str = "load/prompt_library.json") -> None:
        self._cache_dir, self.encode_fn = cache_dir, encode_fn
        self.prompt_library = {}
        if os.path.exists(prompt_library_path):
            with open(prompt_library_path, "r") as f:
                self.prompt_library = json.load(f)
        if self.cfg.use_prompt_debiasing:
            raise NotImplementedError("prompt debiasing needs a BERT masked-LM (base.py:455-510): not on the step path")
        self.prompt = self.preprocess_prompt(self.cfg.prompt)
        self.negative_prompt = self.cfg.negative_prompt
        info(f"Using prompt [{self.prompt}] and negative prompt [{self.negative_prompt}]")
        fmt = (lambda d, s: f"{'backside' if d == 'back' else d} view of {s}") if self.cfg.view_dependent_prompt_front else \
              (lambda d, s: f"{s}, {d} view")
        self.prompts_vd = [getattr(self.cfg, f"prompt_{d}", None) or fmt(d, self.prompt) for d in _DIRECTIONS]
        self.negative_prompts_vd = [self.negative_prompt for _ in _DIRECTIONS]
        info("Using view-dependent prompts " + " ".join(f"[{d}]:[{p}]" for d, p in zip(_DIRECTIONS, self.prompts_vd)))
        self.prepare_text_embeddings()
        self.load_text_embeddings()

    def preprocess_prompt(self, prompt: str) -> str:
        if not prompt.startswith("lib:"):
            return prompt
        keywords = prompt[4:].lower().split("_")
        matches = [p for p in self.prompt_library.get("dreamfusion", []) if all(k in p.lower() for k in keywords)]
        if len(matches) > 1:
            raise ValueError(f"Multiple prompts matched with keywords {keywords} in library")
        if not matches:
            raise ValueError(f"Cannot find prompt with keywords {keywords} in library")
        info("Find matched prompt in library: " + matches[0])
        return matches[0]

    def _cache_path(self, prompt: str) -> str:
        return os.path.join(self._cache_dir, f"{hash_prompt(self.cfg.pretrained_model_name_or_path, prompt)}.pt")

    def prepare_text_embeddings(self) -> None:
        todo = [p for p in dict.fromkeys([self.prompt, self.negative_prompt] + self.prompts_vd + self.negative_prompts_vd)
                if not (self.cfg.use_cache and os.path.exists(self._cache_path(p)))]
        if not todo or self.encode_fn is None:
            return
        os.makedirs(self._cache_dir, exist_ok=True)
        emb = self.encode_fn(todo)
        for p, e in zip(todo, emb):
            torch.save(e.detach().cpu(), self._cache_path(p))

    def load_from_cache(self, prompt: str) -> torch.Tensor:
        path = self._cache_path(prompt)
        if not os.path.exists(path):
            raise FileNotFoundError(f"Text embedding file {path} for model {self.cfg.pretrained_model_name_or_path} and prompt [{prompt}] not found.")
        return torch.load(path, map_location=self.device)

    def load_text_embeddings(self) -> None:
        self.text_embeddings = self.load_from_cache(self.prompt)[None, ...]
        self.uncond_text_embeddings = self.load_from_cache(self.negative_prompt)[None, ...]
        self.text_embeddings_vd = torch.stack([self.load_from_cache(p) for p in self.prompts_vd], dim=0)
        self.uncond_text_embeddings_vd = torch.stack([self.load_from_cache(p) for p in self.negative_prompts_vd], dim=0)

    def __call__(self) -> PromptUtils:
        c = self.cfg
        return PromptUtils(self.text_embeddings_vd, self.uncond_text_embeddings_vd, self.text_embeddings, self.uncond_text_embeddings,
                           use_perp_neg=c.use_perp_neg, overhead_threshold=c.overhead_threshold, front_threshold=c.front_threshold,
                           back_threshold=c.back_threshold, perp_neg_f_sb=tuple(c.perp_neg_f_sb), perp_neg_f_fsb=tuple(c.perp_neg_f_fsb),
                           perp_neg_f_fs=tuple(c.perp_neg_f_fs), perp_neg_f_sf=tuple(c.perp_neg_f_sf))
