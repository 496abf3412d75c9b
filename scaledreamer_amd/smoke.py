"""One tiny end-to-end ASD step on cuda:0 (renderer + guidance + backward + optimizer) with reduced-width
diffusion weights, called by __graft_entry__.smoke()."""
from __future__ import annotations

import random

import torch


def build_smoke_system(seed: int = 0, n_batches: int = 2):
    """the asd_sd_nerf preset with a reduced-width UNet (same topology, 1/6 width: seconds, not minutes) and the full-size VAE
    encoder (34 M parameters), plus `n_batches` seeded camera batches on the device"""
    from . import presets
    from .data import RandomCameraIterableDataset
    from .diffusion import weights as W
    from .diffusion.engine import HipBackend
    from .guidance import PromptUtils
    from .registry import find
    from . import plugins  # noqa: F401

    torch.manual_seed(seed)
    random.seed(seed)
    dev = torch.device("cuda", 0)
    with presets.random_weights_allowed():   # smoke test: seeded random prior (no checkpoint offline); scoped, not a process-wide switch
        cfg = presets.asd_sd_nerf()
    backend = HipBackend(dev, unet_cfg=W.UNetConfig(model_channels=128, context_dim=128), vae_cfg=W.VAEConfig(), seed=3)
    g = torch.Generator().manual_seed(1)
    pu = PromptUtils(torch.randn(4, 77, 128, generator=g).to(dev), torch.randn(1, 77, 128, generator=g).expand(4, -1, -1).contiguous().to(dev),
                     front_threshold=30.0, back_threshold=30.0)
    system = find(cfg["system_type"])(cfg["system"], guidance_backend=backend, prompt_utils=pu)
    system.train()
    data = RandomCameraIterableDataset(cfg["data"])
    batches = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.collate().items()} for _ in range(n_batches)]
    return system, batches


def run_smoke_step() -> float:
    system, batches = build_smoke_system(0, 2)
    loss = None
    for batch in batches:
        loss = system.train_one_step(batch)
    torch.cuda.synchronize()
    val = float(loss.item())
    assert val == val and abs(val) < 1e12, f"non-finite smoke loss {val}"
    g_enc = system.geometry.encoding.encoding.encoding.params.grad
    assert g_enc is not None and float(g_enc.abs().sum()) > 0, "no gradient reached the hash grid"
    return val
