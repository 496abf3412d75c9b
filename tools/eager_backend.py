"""Library-op execution of the frozen SD-2.1 UNet / VAE encoder on the GPU (PyTorch-ROCm eager: MIOpen
convolutions, hipBLASLt GEMMs, SDPA attention), fp16 weights and activations like the reference's diffusers
pipeline (stable_diffusion_asd_guidance.py:38,57-59; channels_last :88-89).

A/B measurement tool only (what PyTorch-ROCm's libraries give on the same GPU): it lives in tools/, is not part of the
product package and is registered as guidance.backend = "eager" only when this module is imported
(`bench.py --backend eager` does).  The product's only diffusion path is scaledreamer_amd/diffusion/engine.py (HIP).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaledreamer_amd.guidance import DiffusionBackend, register_backend  # noqa: E402
from scaledreamer_amd.diffusion import weights as W  # noqa: E402

P = Dict[str, torch.Tensor]


def _gn(p, name, x, eps):
    return F.group_norm(x.float(), 32, p[name + ".weight"].float(), p[name + ".bias"].float(), eps).to(x.dtype)


def _conv(p, name, x, stride=1, padding=1):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def _res(p, name, x, emb):
    h = _conv(p, name + ".in_layers.2", F.silu(_gn(p, name + ".in_layers.0", x, 1e-5)))
    h = h + _lin(p, name + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = _conv(p, name + ".out_layers.3", F.silu(_gn(p, name + ".out_layers.0", h, 1e-5)))
    if name + ".skip_connection.weight" in p:
        x = _conv(p, name + ".skip_connection", x, padding=0)
    return x + h


def _attn(p, name, x, ctx, heads):
    b, n, c = x.shape
    d = c // heads
    q = _lin(p, name + ".to_q", x).view(b, n, heads, d).transpose(1, 2)
    k = _lin(p, name + ".to_k", ctx).view(b, ctx.shape[1], heads, d).transpose(1, 2)
    v = _lin(p, name + ".to_v", ctx).view(b, ctx.shape[1], heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    return _lin(p, name + ".to_out.0", o.transpose(1, 2).reshape(b, n, c))


def _transformer(p, name, x, ctx, head_dim, depth):
    b, c, h, w = x.shape
    x_in = x
    x = _gn(p, name + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = _lin(p, name + ".proj_in", x)
    heads = c // head_dim
    for d in range(depth):
        blk = f"{name}.transformer_blocks.{d}"
        ln = lambda t, n: F.layer_norm(t, (c,), p[f"{blk}.{n}.weight"], p[f"{blk}.{n}.bias"])
        y = ln(x, "norm1")
        x = _attn(p, blk + ".attn1", y, y, heads) + x
        x = _attn(p, blk + ".attn2", ln(x, "norm2"), ctx, heads) + x
        a, gate = _lin(p, blk + ".ff.net.0.proj", ln(x, "norm3")).chunk(2, dim=-1)
        x = _lin(p, blk + ".ff.net.2", a * F.gelu(gate)) + x
    x = _lin(p, name + ".proj_out", x)
    return x.reshape(b, h, w, c).permute(0, 3, 1, 2) + x_in


def _apply(p, layers, h, emb, ctx, cfg):
    for kind, name, cin, cout in layers:
        if kind == "conv":
            h = _conv(p, name, h)
        elif kind == "res":
            h = _res(p, name, h, emb)
        elif kind == "attn":
            h = _transformer(p, name, h, ctx, cfg.num_head_channels, cfg.transformer_depth)
        elif kind == "down":
            h = _conv(p, name, h, stride=2)
        elif kind == "up":
            h = _conv(p, name, F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def unet_forward(p: P, layout, cfg: W.UNetConfig, x, t, ctx):
    _, inputs, middle, outputs = layout
    dt = p["time_embed.0.weight"].dtype
    emb = _lin(p, "time_embed.2", F.silu(_lin(p, "time_embed.0", timestep_embedding(t, cfg.model_channels).to(dt))))
    hs, h = [], x.to(dt).contiguous(memory_format=torch.channels_last)
    ctx = ctx.to(dt)
    for blk in inputs:
        h = _apply(p, blk.layers, h, emb, ctx, cfg)
        hs.append(h)
    h = _apply(p, middle.layers, h, emb, ctx, cfg)
    for blk in outputs:
        h = _apply(p, blk.layers, torch.cat([h, hs.pop()], dim=1), emb, ctx, cfg)
    return _conv(p, "out.2", F.silu(_gn(p, "out.0", h, 1e-5)))


def _vae_res(p, name, x):
    h = _conv(p, name + ".conv1", F.silu(_gn(p, name + ".norm1", x, 1e-6)))
    h = _conv(p, name + ".conv2", F.silu(_gn(p, name + ".norm2", h, 1e-6)))
    if name + ".nin_shortcut.weight" in p:
        x = _conv(p, name + ".nin_shortcut", x, padding=0)
    return x + h


def _vae_attn(p, name, x):
    h = _gn(p, name + ".norm", x, 1e-6)
    q, k, v = (_conv(p, f"{name}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    f = lambda t: t.reshape(b, 1, c, hh * ww).transpose(2, 3)
    o = F.scaled_dot_product_attention(f(q), f(k), f(v)).transpose(2, 3).reshape(b, c, hh, ww)
    return x + _conv(p, name + ".proj_out", o, padding=0)


def vae_encode_moments(p: P, plan, x):
    h = x.to(p["quant_conv.weight"].dtype).contiguous(memory_format=torch.channels_last)
    for kind, name, cin, cout in plan:
        if kind == "conv":
            h = _conv(p, name, h)
        elif kind == "res":
            h = _vae_res(p, name, h)
        elif kind == "down":
            h = _conv(p, name, F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
        elif kind == "attn":
            h = _vae_attn(p, name, h)
        elif kind == "out":
            h = _conv(p, name + ".conv_out", F.silu(_gn(p, name + ".norm_out", h, 1e-6)))
        elif kind == "quant":
            h = _conv(p, name, h, padding=0)
    return h


class EagerBackend(DiffusionBackend):
    def __init__(self, device, dtype=torch.float16, seed: int = 1, unet_cfg: Optional[W.UNetConfig] = None,
                 vae_cfg: Optional[W.VAEConfig] = None, unet_params: Optional[P] = None, vae_params: Optional[P] = None):
        self.unet_cfg = unet_cfg or W.UNetConfig()
        self.vae_cfg = vae_cfg or W.VAEConfig()
        self.scaling_factor = self.vae_cfg.scale_factor
        self.unet_layout = W.unet_layout(self.unet_cfg)
        self.vae_shapes, self.vae_plan = W.vae_encoder_layout(self.vae_cfg)
        up = unet_params if unet_params is not None else W.gen_params(self.unet_layout[0], seed)
        vp = vae_params if vae_params is not None else W.gen_params(self.vae_shapes, seed + 1)
        cl = lambda t: t.contiguous(memory_format=torch.channels_last) if t.ndim == 4 else t
        self.up = {k: cl(v.to(device=device, dtype=dtype)) for k, v in up.items()}
        self.vp = {k: cl(v.to(device=device, dtype=dtype)) for k, v in vp.items()}
        self.dtype = dtype

    @torch.no_grad()
    def unet(self, latents, t, context, camera=None, num_frames: int = 1):
        if camera is not None:
            raise NotImplementedError("the library-op backend covers the SD-2.1 UNet only")
        return unet_forward(self.up, self.unet_layout, self.unet_cfg, latents, t, context)

    def encode(self, images):
        return vae_encode_moments(self.vp, self.vae_plan, images)


@register_backend("eager")
def _make_eager(cfg, device, dtype):
    return EagerBackend(device, dtype, seed=getattr(cfg, "weights_seed", 1))
