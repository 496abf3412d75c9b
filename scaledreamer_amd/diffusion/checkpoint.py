"""Loading the frozen prior's weights from disk into the name-keyed LDM layout of `weights.py`.

The reference reads two formats:
  * SD guidance (`stable_diffusion_asd_guidance.py:61-71`): a diffusers pipeline directory —
    `<path>/unet/diffusion_pytorch_model.{safetensors,bin}` and `<path>/vae/diffusion_pytorch_model.{safetensors,bin}` with
    diffusers parameter names (`down_blocks.0.resnets.0.norm1.weight` ...);
  * MVDream guidance (`mvdream_asd_guidance.py:67`, `extern/mvdream/model_zoo.py`): one LDM checkpoint whose UNet lives under
    `model.diffusion_model.` and whose VAE lives under `first_stage_model.` (names = `weights.unet_layout` / `vae_encoder_layout`).
Both end up as `{ldm_name: tensor}` dictionaries checked against the layout's shapes (missing / unexpected / mis-shaped keys
raise).  Random weights are never substituted silently: `resolve_params` raises unless the caller passed
`allow_random_weights=True` (bench, smoke and the parity tests do; no pretrained weights exist offline).
"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional, Tuple

import torch

from . import weights as W

P = Dict[str, torch.Tensor]


class MissingWeightsError(FileNotFoundError):
    pass


def _read_state_dict(path: str) -> P:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    return obj


def _first_existing(*paths: str) -> Optional[str]:
    for p in paths:
        if os.path.isfile(p):
            return p
    return None


# ---- diffusers -> LDM parameter names -----------------------------------------------------------------------------------------
_RES = {"norm1": "in_layers.0", "conv1": "in_layers.2", "time_emb_proj": "emb_layers.1", "norm2": "out_layers.0",
        "conv2": "out_layers.3", "conv_shortcut": "skip_connection"}


def diffusers_unet_key_to_ldm(key: str, cfg: W.UNetConfig) -> str:
    """UNet2DConditionModel name -> UNetModel name.  Block numbering follows the construction loop of `weights.unet_layout`:
    input block index = 1 + level * (num_res_blocks + 1) + j, output block index = level' * (num_res_blocks + 1) + j."""
    nrb = cfg.num_res_blocks
    m = re.match(r"time_embedding\.linear_(\d)\.(.+)", key)
    if m:
        return f"time_embed.{(int(m.group(1)) - 1) * 2}.{m.group(2)}"
    if key.startswith("conv_in."):
        return "input_blocks.0.0." + key[len("conv_in."):]
    if key.startswith("conv_norm_out."):
        return "out.0." + key[len("conv_norm_out."):]
    if key.startswith("conv_out."):
        return "out.2." + key[len("conv_out."):]

    def res(rest: str) -> str:
        head, tail = rest.split(".", 1)
        if head not in _RES:
            raise KeyError(key)
        return f"{_RES[head]}.{tail}"

    m = re.match(r"down_blocks\.(\d+)\.(resnets|attentions|downsamplers)\.(\d+)\.(.+)", key)
    if m:
        lvl, kind, j, rest = int(m.group(1)), m.group(2), int(m.group(3)), m.group(4)
        if kind == "downsamplers":
            return f"input_blocks.{(lvl + 1) * (nrb + 1)}.0.op.{rest[len('conv.'):]}"
        i = 1 + lvl * (nrb + 1) + j
        return f"input_blocks.{i}.0.{res(rest)}" if kind == "resnets" else f"input_blocks.{i}.1.{rest}"
    m = re.match(r"mid_block\.(resnets|attentions)\.(\d+)\.(.+)", key)
    if m:
        kind, j, rest = m.group(1), int(m.group(2)), m.group(3)
        return f"middle_block.{2 * j}.{res(rest)}" if kind == "resnets" else f"middle_block.1.{rest}"
    m = re.match(r"up_blocks\.(\d+)\.(resnets|attentions|upsamplers)\.(\d+)\.(.+)", key)
    if m:
        lvl, kind, j, rest = int(m.group(1)), m.group(2), int(m.group(3)), m.group(4)
        if kind == "upsamplers":
            i = lvl * (nrb + 1) + nrb
            # the upsampler follows the ResBlock and, where the resolution has attention, the transformer
            ds_here = 2 ** (len(cfg.channel_mult) - 1 - lvl)
            n = 2 if ds_here in cfg.attention_resolutions else 1
            return f"output_blocks.{i}.{n}.conv.{rest[len('conv.'):]}"
        i = lvl * (nrb + 1) + j
        return f"output_blocks.{i}.0.{res(rest)}" if kind == "resnets" else f"output_blocks.{i}.1.{rest}"
    raise KeyError(key)


_VAE_ATTN = {"group_norm": "norm", "query": "q", "to_q": "q", "key": "k", "to_k": "k", "value": "v", "to_v": "v",
             "proj_attn": "proj_out", "to_out.0": "proj_out"}


def diffusers_vae_key_to_ldm(key: str) -> Optional[str]:
    """AutoencoderKL name -> LDM first-stage name (encoder + quant_conv only; decoder keys return None)."""
    if key.startswith("quant_conv."):
        return key
    if not key.startswith("encoder."):
        return None
    k = key[len("encoder."):]
    for a, b in (("conv_in.", "conv_in."), ("conv_out.", "conv_out."), ("conv_norm_out.", "norm_out.")):
        if k.startswith(a):
            return "encoder." + b + k[len(a):]
    m = re.match(r"down_blocks\.(\d+)\.resnets\.(\d+)\.(.+)", k)
    if m:
        return f"encoder.down.{m.group(1)}.block.{m.group(2)}.{m.group(3).replace('conv_shortcut', 'nin_shortcut')}"
    m = re.match(r"down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.+)", k)
    if m:
        return f"encoder.down.{m.group(1)}.downsample.conv.{m.group(2)}"
    m = re.match(r"mid_block\.resnets\.(\d)\.(.+)", k)
    if m:
        return f"encoder.mid.block_{int(m.group(1)) + 1}.{m.group(2).replace('conv_shortcut', 'nin_shortcut')}"
    m = re.match(r"mid_block\.attentions\.0\.(.+)\.(weight|bias)", k)
    if m and m.group(1) in _VAE_ATTN:
        return f"encoder.mid.attn_1.{_VAE_ATTN[m.group(1)]}.{m.group(2)}"
    raise KeyError(key)


def _check(params: P, shapes: W.Shapes, what: str) -> P:
    missing = [k for k in shapes if k not in params]
    extra = [k for k in params if k not in shapes]
    if missing or extra:
        raise KeyError(f"{what}: {len(missing)} missing / {len(extra)} unexpected parameters (e.g. missing {missing[:3]}, unexpected {extra[:3]})")
    out = {}
    for k, shp in shapes.items():
        t = params[k]
        if tuple(t.shape) != tuple(shp):
            if len(shp) == 4 and t.ndim == 2 and tuple(t.shape) == tuple(shp[:2]):     # Linear stored for a 1x1 convolution
                t = t.reshape(shp)
            else:
                raise ValueError(f"{what}: {k} has shape {tuple(t.shape)}, the architecture needs {tuple(shp)}")
        out[k] = t
    return out


def load_diffusers_pipeline(path: str, unet_cfg: W.UNetConfig, vae_cfg: W.VAEConfig) -> Tuple[P, P]:
    """<path>/unet + <path>/vae of a diffusers StableDiffusionPipeline -> (unet params, vae-encoder params) in LDM names."""
    uf = _first_existing(*(os.path.join(path, "unet", f) for f in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin")))
    vf = _first_existing(*(os.path.join(path, "vae", f) for f in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin")))
    if uf is None or vf is None:
        raise MissingWeightsError(f"{path}: no unet/ or vae/ diffusion_pytorch_model.(safetensors|bin)")
    unet = {diffusers_unet_key_to_ldm(k, unet_cfg): v for k, v in _read_state_dict(uf).items()}
    vae = {}
    for k, v in _read_state_dict(vf).items():
        nk = diffusers_vae_key_to_ldm(k)
        if nk is not None:
            vae[nk] = v
    return (_check(unet, W.unet_layout(unet_cfg)[0], "diffusers UNet"), _check(vae, W.vae_encoder_layout(vae_cfg)[0], "diffusers VAE encoder"))


def load_ldm_checkpoint(path: str, unet_cfg: W.UNetConfig, vae_cfg: W.VAEConfig) -> Tuple[P, P]:
    """one LDM / MVDream checkpoint: `model.diffusion_model.*` and `first_stage_model.{encoder,quant_conv}.*`."""
    sd = _read_state_dict(path)
    up, vp = "model.diffusion_model.", "first_stage_model."
    unet = {k[len(up):]: v for k, v in sd.items() if k.startswith(up)}
    vshapes = W.vae_encoder_layout(vae_cfg)[0]
    vae = {k[len(vp):]: v for k, v in sd.items() if k.startswith(vp) and k[len(vp):] in vshapes}
    return _check(unet, W.unet_layout(unet_cfg)[0], "LDM UNet"), _check(vae, vshapes, "LDM first stage")


def hub_cache_snapshot(repo_id: str) -> Optional[str]:
    """A hub id ("org/name", the reference's default `pretrained_model_name_or_path`) resolved OFFLINE against the local Hugging Face
    cache layout: $HF_HUB_CACHE | $HF_HOME/hub | ~/.cache/huggingface/hub / models--org--name / snapshots / <revision> — the newest
    snapshot that holds a unet/ directory.  None when the model was never downloaded to this machine."""
    if not repo_id or repo_id.count("/") != 1 or repo_id.startswith(("/", ".", "~")):
        return None
    roots = [os.environ.get("HF_HUB_CACHE"), os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
             os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]
    for root in roots:
        snaps = os.path.join(root, "models--" + repo_id.replace("/", "--"), "snapshots") if root else None
        if snaps and os.path.isdir(snaps):
            cands = [os.path.join(snaps, d) for d in os.listdir(snaps) if os.path.isdir(os.path.join(snaps, d, "unet"))]
            if cands:
                return max(cands, key=os.path.getmtime)
    return None


def resolve_params(cfg, unet_cfg: W.UNetConfig, vae_cfg: W.VAEConfig):
    """(unet_params, vae_params, description) for a guidance config: `ckpt_path` (MVDream) or `pretrained_model_name_or_path`
    (SD, a local diffusers directory) when they exist on disk; seeded random weights only behind `allow_random_weights`."""
    ckpt = getattr(cfg, "ckpt_path", None)
    name = getattr(cfg, "pretrained_model_name_or_path", None)
    if ckpt and os.path.isfile(ckpt):
        return (*load_ldm_checkpoint(ckpt, unet_cfg, vae_cfg), f"LDM checkpoint {ckpt}")
    if name and os.path.isdir(name):
        return (*load_diffusers_pipeline(name, unet_cfg, vae_cfg), f"diffusers pipeline {name}")
    if name and os.path.isfile(name):
        return (*load_ldm_checkpoint(name, unet_cfg, vae_cfg), f"LDM checkpoint {name}")
    snap = hub_cache_snapshot(name) if name else None
    if snap:
        return (*load_diffusers_pipeline(snap, unet_cfg, vae_cfg), f"diffusers pipeline {name} (local hub cache: {snap})")
    wanted = ckpt or name or getattr(cfg, "model_name", None)
    if not getattr(cfg, "allow_random_weights", False):
        raise MissingWeightsError(
            f"diffusion weights {wanted!r} are not on disk (no hub download: this box has no network).  Point "
            "`pretrained_model_name_or_path` at a local diffusers directory / `ckpt_path` at an LDM checkpoint, or set "
            "`allow_random_weights: true` to run on seeded random weights (benchmarks and tests only — it distills nothing).")
    seed = getattr(cfg, "weights_seed", 1)
    from ..registry import warn

    warn(f"diffusion prior: {wanted!r} not found, using SEEDED RANDOM weights (seed {seed}) because allow_random_weights is set")
    return None, None, f"seeded random init (seed {seed})"
