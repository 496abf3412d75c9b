"""Multi-prompt (amortized) training on the HIP path:
  `multiprompt-radience-field-generator-system`  custom/amortized/systems/multiprompt_radience_field_generator.py:18-222
  `multiprompt-camera-datamodule`                custom/amortized/data/multiprompt.py:22-186
  prompt side                                    custom/amortized/models/prompt_processors/base.py:379-560
                                                 (MultiPromptProcessorOutput: per-prompt global / view-dependent embeddings)
Each rank trains on its shard `library[rank::n_ranks]` of the prompt library (multiprompt.py:177-186); per step the
datamodule draws `batch_size` prompts, the system maps their global embeddings [B, 1024] through the geometry's and the
background's hypernetworks and renders one random view per prompt.
"""
from __future__ import annotations

import json
import os
import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from .base import get_rank
from .config import parse_structured
from .data import RandomCameraDataModuleConfig, RandomCameraIterableDataset, RandomMultiviewCameraIterableDataset
from .guidance import shift_azimuth_deg, shifted_expotional_decay
from .registry import find, register
from .system import StableDreamer, binary_cross_entropy, dot


@dataclass
class MultiPromptUtils:
    """MultiPromptProcessorOutput (prompt_processors/base.py:407-560): one entry per batch element."""
    global_text_embeddings: List[torch.Tensor]          # B x [1024]
    local_text_embeddings: List[torch.Tensor]           # B x [77, 1024]
    uncond_text_embeddings: torch.Tensor                # [77, 1024]
    text_embeddings_vd: List[torch.Tensor]              # B x [4, 77, 1024]  side / front / back / overhead
    uncond_text_embeddings_vd: torch.Tensor             # [4, 77, 1024]
    use_perp_neg: bool = True
    overhead_threshold: float = 60.0
    front_threshold: float = 45.0
    back_threshold: float = 45.0
    perp_neg_f_sb: Tuple[float, float, float] = (1, 0.5, -0.606)
    perp_neg_f_fsb: Tuple[float, float, float] = (1, 0.5, +0.967)
    perp_neg_f_fs: Tuple[float, float, float] = (4, 0.5, -2.426)
    perp_neg_f_sf: Tuple[float, float, float] = (4, 0.5, -2.426)
    use_local_text_embeddings: bool = False

    def direction_idx(self, elevation, azimuth, camera_distances):
        """side 0 / front 1 / back 2 / overhead 3, later rules overriding earlier ones (prompt_processors/base.py:262-294) — as selections, so
        that device tensors are never read back (same values as the reference's masked assignments)"""
        azi = shift_azimuth_deg(azimuth)
        one, two, three = (torch.full_like(elevation, k, dtype=torch.long) for k in (1, 2, 3))
        idx = torch.where((azi > -self.front_threshold) & (azi < self.front_threshold), one, torch.zeros_like(one))
        idx = torch.where((azi > 180 - self.back_threshold) | (azi < -180 + self.back_threshold), two, idx)
        return torch.where(elevation > self.overhead_threshold, three, idx)

    def get_global_text_embeddings(self) -> torch.Tensor:
        return torch.stack(self.local_text_embeddings if self.use_local_text_embeddings else self.global_text_embeddings, dim=0)

    def get_text_embeddings(self, elevation, azimuth, camera_distances, view_dependent_prompting: bool = True):
        B = len(self.global_text_embeddings)
        if view_dependent_prompting:
            idx = self.direction_idx(elevation, azimuth, camera_distances)
            # (one gather with tensor indices: `self.text_embeddings_vd[i][idx[i]]` reads every idx[i] back to the host on a device tensor)
            text = torch.stack(list(self.text_embeddings_vd), dim=0)[torch.arange(B, device=idx.device), idx]
            uncond = self.uncond_text_embeddings_vd[idx]
        else:
            text = torch.stack(list(self.local_text_embeddings), dim=0)
            uncond = self.uncond_text_embeddings.unsqueeze(0).expand(B, -1, -1)
        return torch.cat([text, uncond], dim=0)

    def get_text_embeddings_perp_neg(self, elevation, azimuth, camera_distances, view_dependent_prompting: bool = True,
                                     guidance_scale_neg: Optional[float] = None):
        assert view_dependent_prompting, "Perp-Neg only works with view-dependent prompting"
        B = len(self.global_text_embeddings)
        gs = -1 if guidance_scale_neg is None else guidance_scale_neg
        if elevation.is_cuda and os.environ.get("ASD_PERP_NEG_ON_DEVICE", "1") != "0":      # =0: the branching form (same-box A/B)
            return self._perp_neg_on_device(elevation, azimuth, gs)
        idx = self.direction_idx(elevation, azimuth, camera_distances)
        pos, neg, uncond, weights = [], [], [], []
        for b in range(B):
            side, front, back, overhead = (self.text_embeddings_vd[b][k] for k in range(4))
            i = int(idx[b])
            azi = shift_azimuth_deg(azimuth[b])
            uncond.append(self.uncond_text_embeddings_vd[i])
            if i == 3:
                pos.append(overhead)
                neg += [self.uncond_text_embeddings_vd[i], self.uncond_text_embeddings_vd[i]]
                weights += [0.0, 0.0]
            elif torch.abs(azi) < 90:
                r = 1 - torch.abs(azi) / 90
                pos.append(r * front + (1 - r) * side)
                neg += [front, side]
                weights += [shifted_expotional_decay(*self.perp_neg_f_fs, r) * gs, shifted_expotional_decay(*self.perp_neg_f_sf, 1 - r) * gs]
            else:
                r = 2.0 - torch.abs(azi) / 90
                pos.append(r * side + (1 - r) * back)
                neg += [side, front]
                weights += [shifted_expotional_decay(*self.perp_neg_f_sb, r) * gs, shifted_expotional_decay(*self.perp_neg_f_fsb, r) * gs]
        text = torch.cat([torch.stack(pos, 0), torch.stack(uncond, 0), torch.stack(neg, 0)], dim=0)
        return text, torch.as_tensor(weights, device=elevation.device).reshape(B, 2)


def _perp_neg_on_device(self, elevation, azimuth, gs):
    """get_text_embeddings_perp_neg for device tensors WITHOUT reading the angles back: the reference (prompt_processors/base.py:470-533)
    branches on `int(idx[b])` / `torch.abs(azi) < 90` per batch element — one host synchronisation per element and step, after which the
    device idles until the host has caught up.  Here the three cases are evaluated with the same elementwise expressions on [B, 1, 1]
    coefficients and selected with torch.where: bit-identical embeddings and weights (tests/test_gpu_asd_glue.py)."""
    vd = torch.stack(list(self.text_embeddings_vd), dim=0)                     # [B, 4, 77, D]: side / front / back / overhead
    side, front, back, overhead = vd[:, 0], vd[:, 1], vd[:, 2], vd[:, 3]
    azi = shift_azimuth_deg(azimuth)
    a = torch.abs(azi)
    idx = self.direction_idx(elevation, azimuth, None)
    unc = self.uncond_text_embeddings_vd[idx]                                   # [B, 77, D]
    is_over = (idx == 3)
    near = a < 90
    r1 = (1 - a / 90).view(-1, 1, 1)                                            # front half: r * front + (1 - r) * side
    r2 = (2.0 - a / 90).view(-1, 1, 1)                                          # back half:  r * side + (1 - r) * back
    o3, n3 = is_over.view(-1, 1, 1), near.view(-1, 1, 1)
    pos = torch.where(o3, overhead, torch.where(n3, r1 * front + (1 - r1) * side, r2 * side + (1 - r2) * back))
    neg0 = torch.where(o3, unc, torch.where(n3, front, side))
    neg1 = torch.where(o3, unc, torch.where(n3, side, front))
    q1, q2 = r1.view(-1), r2.view(-1)
    w0 = torch.where(near, shifted_expotional_decay(*self.perp_neg_f_fs, q1) * gs, shifted_expotional_decay(*self.perp_neg_f_sb, q2) * gs)
    w1 = torch.where(near, shifted_expotional_decay(*self.perp_neg_f_sf, 1 - q1) * gs, shifted_expotional_decay(*self.perp_neg_f_fsb, q2) * gs)
    zero = torch.zeros_like(w0)
    weights = torch.stack([torch.where(is_over, zero, w0), torch.where(is_over, zero, w1)], dim=1)
    neg = torch.stack([neg0, neg1], dim=1).reshape(-1, *neg0.shape[1:])         # (b, k) order: two negatives per sample, interleaved
    return torch.cat([pos, unc, neg], dim=0), weights


MultiPromptUtils._perp_neg_on_device = _perp_neg_on_device


class SyntheticMultiPromptProcessor:
    """Stand-in for `stable-diffusion-multi-prompt-processor` when no text encoder / embedding cache exists offline:
    N(0,1) embeddings per prompt name, generated once per (seed, prompt) — "synthetic random prompts" (BASELINE.json)."""

    def __init__(self, prompts: List[str], seed: int = 1234, device="cpu", ctx_dim: int = 1024, global_dim: Optional[int] = None, **utils_kw):
        self.prompt_library = list(prompts)
        self.device, self.kw = device, utils_kw
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: torch.randn(*s, generator=g).to(device)
        self.uncond_vd = mk(1, 77, ctx_dim).expand(4, -1, -1).contiguous()
        self.table = {p: (mk(global_dim or ctx_dim), mk(77, ctx_dim), mk(4, 77, ctx_dim)) for p in self.prompt_library}

    def __call__(self, prompt: Union[str, List[str]]) -> MultiPromptUtils:
        prompt = [prompt] if isinstance(prompt, str) else prompt
        for p in prompt:
            if p not in self.table:
                raise ValueError(f"Prompt [{p}] is not in the prompt library.")
        return MultiPromptUtils([self.table[p][0] for p in prompt], [self.table[p][1] for p in prompt], self.uncond_vd[0],
                                [self.table[p][2] for p in prompt], self.uncond_vd, **self.kw)


@dataclass
class MultipromptRandomCameraDataModuleConfig(RandomCameraDataModuleConfig):
    dim_gaussian: int = 512
    prompt_library: Any = "magic3d_prompt_library"      # name of <prompt_library_dir>/<name>.json, or an in-memory dict
    prompt_library_dir: str = "load"
    prompt_library_format: str = "json"
    eval_prompt: Optional[str] = None
    target_prompt: Optional[str] = None
    eval_fix_camera: Optional[int] = None


@register("multiprompt-camera-datamodule")
class MultipromptRandomCameraIterableDataset(RandomCameraIterableDataset):
    def __init__(self, cfg: Any, prompt_library: Optional[Dict[str, List[str]]] = None, rank: Optional[int] = None,
                 n_ranks: Optional[int] = None) -> None:
        cfg_mp = parse_structured(MultipromptRandomCameraDataModuleConfig, cfg)
        super().__init__({k: getattr(cfg_mp, k) for k in RandomCameraDataModuleConfig.__dataclass_fields__})
        self.cfg = cfg_mp
        if prompt_library is None:
            if isinstance(cfg_mp.prompt_library, dict):
                prompt_library = cfg_mp.prompt_library
            else:
                path = os.path.join(cfg_mp.prompt_library_dir, cfg_mp.prompt_library) + "." + cfg_mp.prompt_library_format
                with open(path, "r") as f:
                    prompt_library = json.load(f)
        rank = get_rank() if rank is None else rank
        n_ranks = int(os.environ.get("WORLD_SIZE", max(1, torch.cuda.device_count()))) if n_ranks is None else n_ranks
        assert "train" in prompt_library, "prompt library must contain train split"
        self.prompt_library = list(prompt_library["train"])[rank::n_ranks]     # multiprompt.py:177-186

    def cameras(self) -> Dict[str, Any]:
        """camera draws, then the generator noise and the prompt choice (custom/amortized/data/multiprompt.py:62-83); collate() adds
        the device rays"""
        out = super().cameras()
        out["noise"] = torch.randn(self.batch_size, self.cfg.dim_gaussian)
        if len(self.prompt_library) < self.batch_size:
            out["prompt"] = random.choices(self.prompt_library, k=self.batch_size)
        else:
            out["prompt"] = random.sample(self.prompt_library, k=self.batch_size)
        return out


@dataclass
class MultiviewMultipromptRandomCameraDataModuleConfig(RandomCameraDataModuleConfig):
    dim_gaussian: int = 512
    prompt_library: Any = "magic3d_prompt_library"
    prompt_library_dir: str = "load"
    prompt_library_format: str = "json"
    eval_prompt: Optional[str] = None
    target_prompt: Optional[str] = None
    eval_fix_camera: Optional[int] = None
    relative_radius: bool = True
    n_view: int = 1
    zoom_range: Tuple[float, float] = (1.0, 1.0)


@register("multiprompt-multiview-camera-datamodule")
class MultiviewMultipromptRandomCameraIterableDataset(RandomMultiviewCameraIterableDataset):
    """custom/amortized/data/multiview_multiprompt.py:36-75: groups of n_view cameras, one prompt (and noise vector) per group."""

    def __init__(self, cfg: Any, prompt_library: Optional[Dict[str, List[str]]] = None, rank: Optional[int] = None,
                 n_ranks: Optional[int] = None) -> None:
        cfg_mp = parse_structured(MultiviewMultipromptRandomCameraDataModuleConfig, cfg)
        from .data import RandomMultiviewCameraDataModuleConfig
        super().__init__({k: getattr(cfg_mp, k) for k in RandomMultiviewCameraDataModuleConfig.__dataclass_fields__})
        self.mp_cfg = cfg_mp
        self.n_view = cfg_mp.n_view
        if prompt_library is None:
            if isinstance(cfg_mp.prompt_library, dict):
                prompt_library = cfg_mp.prompt_library
            else:
                path = os.path.join(cfg_mp.prompt_library_dir, cfg_mp.prompt_library) + "." + cfg_mp.prompt_library_format
                with open(path, "r") as f:
                    prompt_library = json.load(f)
        rank = get_rank() if rank is None else rank
        n_ranks = int(os.environ.get("WORLD_SIZE", max(1, torch.cuda.device_count()))) if n_ranks is None else n_ranks
        assert "train" in prompt_library, "prompt library must contain train split"
        self.prompt_library = list(prompt_library["train"])[rank::n_ranks]

    def cameras(self) -> Dict[str, Any]:
        groups = self.batch_size // self.n_view
        out = super().cameras()
        out["noise"] = torch.randn(groups, self.mp_cfg.dim_gaussian)
        if len(self.prompt_library) < groups:
            out["prompt"] = random.choices(self.prompt_library, k=groups)
        else:
            out["prompt"] = random.sample(self.prompt_library, k=groups)
        return out


@register("multiprompt-radience-field-generator-system")
class MultipromptRadienceFieldGeneratorSystem(StableDreamer):
    @dataclass
    class Config(StableDreamer.Config):
        rgb_as_latents: bool = False
        initialize_shape: bool = True
        train_guidance: bool = False

    cfg: Config

    def __init__(self, cfg, guidance_backend=None, prompt_processor=None) -> None:
        super().__init__(cfg, guidance_backend=guidance_backend, prompt_utils=None)
        if self.cfg.initialize_shape and hasattr(self.geometry, "initialize_shape"):
            self.geometry.initialize_shape()
        self.prompt_processor = prompt_processor

    def forward(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        self.prompt_utils = self.prompt_processor(prompt=batch["prompt"])
        if "prompt_target" in batch:  # interpolation between two prompts (test time)
            target = self.prompt_processor(prompt=batch["prompt_target"])
            r = batch["ratio"]
            batch["text_embed"] = r * self.prompt_utils.get_global_text_embeddings() + (1 - r) * target.get_global_text_embeddings()
        else:
            batch["text_embed"] = self.prompt_utils.get_global_text_embeddings()
        if self.cfg.stage == "geometry":
            return {**self.renderer(**batch, render_rgb=False)}
        return {**self.renderer(**batch)}

    GEOMETRY_PASS_WEIGHT = 0.2       # multiprompt_radience_field_generator.py:203

    def _rgb_as_latents(self) -> bool:
        return self.cfg.rgb_as_latents
