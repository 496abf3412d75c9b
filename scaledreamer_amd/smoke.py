"""One tiny end-to-end ASD step on cuda:0 (renderer + guidance + backward + optimizer) with reduced-width
diffusion weights, called by __graft_entry__.smoke()."""
from __future__ import annotations

import random

import torch


def run_smoke_step() -> float:
    from . import presets
    from .data import RandomCameraIterableDataset
    from .diffusion import weights as W
    from .diffusion.engine import HipBackend
    from .guidance import PromptUtils
    from .registry import find
    from . import plugins  # noqa: F401

    torch.manual_seed(0)
    random.seed(0)
    dev = torch.device("cuda", 0)
    cfg = presets.asd_sd_nerf()
    ucfg = W.UNetConfig(model_channels=128, context_dim=128)      # same topology, 1/6 width: seconds, not minutes
    vcfg = W.VAEConfig()                                           # full-size VAE encoder (34 M parameters)
    backend = HipBackend(dev, unet_cfg=ucfg, vae_cfg=vcfg, seed=3)
    g = torch.Generator().manual_seed(1)
    pu = PromptUtils(torch.randn(4, 77, 128, generator=g).to(dev), torch.randn(1, 77, 128, generator=g).expand(4, -1, -1).contiguous().to(dev),
                     front_threshold=30.0, back_threshold=30.0)
    system = find(cfg["system_type"])(cfg["system"], guidance_backend=backend, prompt_utils=pu)
    system.train()
    data = RandomCameraIterableDataset(cfg["data"])
    loss = None
    for _ in range(2):
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.collate().items()}
        loss = system.train_one_step(batch)
    torch.cuda.synchronize()
    val = float(loss.item())
    assert val == val and abs(val) < 1e12, f"non-finite smoke loss {val}"
    g_enc = system.geometry.encoding.encoding.encoding.params.grad
    assert g_enc is not None and float(g_enc.abs().sum()) > 0, "no gradient reached the hash grid"
    return val
