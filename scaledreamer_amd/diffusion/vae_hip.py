"""HIP execution of the frozen SD VAE encoder, forward AND input-gradient backward.

The reference keeps the VAE encoder inside the autograd graph of the ASD loss (the latents are a function of
the rendered image: stable_diffusion_asd_guidance.py:171-178,204-208,225; parameters frozen :101-102), so
every step pays a forward and an input-gradient ("dgrad") pass at 512x512.  Architecture: Encoder
(extern/mvdream/ldm/modules/diffusionmodules/model.py:452-543), ResnetBlock :88-146, Downsample :66-85,
AttnBlock :152-203, quant_conv (models/autoencoder.py:32,81-85).

The product path is `HipVAEEncoder`: ONE autograd node around the C-ABI network asd_vae_enc_fwd / asd_vae_enc_bwd
(csrc/net.hip), which enqueues every layer itself — 3x3 convolutions forward and dgrad are the same implicit-GEMM MFMA kernel
(the dgrad uses weights transposed/flipped once at load; the stride-2 dgrad is the kernel's "transposed" gather mode),
GroupNorm+SiLU forward/backward are the NHWC kernels of nn_ops.hip, 1x1 convolutions and the single-head mid-block attention are
GEMMs around row-softmax kernels; conv_out and quant_conv are both linear and are folded into one 3x3 convolution at pack time.
The per-layer autograd Functions below (_Conv3x3Fn, _GroupNormFn) drive the same leaf kernels one at a time; they are kept as
the unit-test harness of those kernels' gradients (tests/test_gpu_vae_hip.py), not used by the encoder.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

import ctypes as C

from .._lib import AsdError, VaeDesc, check, i32, lib, ptr, stream
from . import hip_ops as H
from . import weights as W

P = Dict[str, torch.Tensor]


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_fwd, bias, w_bwd, stride, pad, residual=None):
        B, Hh, Ww, _ = x.shape
        if stride == 2:  # asymmetric (0,1,0,1) zero padding, model.py:80-85
            out_hw = ((Hh + 1 - 3) // 2 + 1, (Ww + 1 - 3) // 2 + 1)
        else:
            out_hw = (Hh, Ww)
        ctx.w_bwd, ctx.stride, ctx.in_hw = w_bwd, stride, (Hh, Ww)
        ctx.has_res, ctx.n_out = residual is not None, w_fwd.shape[0]
        # the shortcut of a ResnetBlock (model.py:141-148: x + h) is added in the conv epilogue: no separate add pass
        return H.conv3x3(x, w_fwd, bias=bias, stride=stride, pad=pad, out_hw=out_hw, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if dy.shape[-1] % 32:  # e.g. the 8 moment channels: pad the contraction dim to the kernel's K granularity
            dy = F.pad(dy, (0, 32 - dy.shape[-1] % 32))
        if ctx.stride == 1:
            dx = H.conv3x3(dy, ctx.w_bwd, stride=1, pad=1)
        else:
            dx = H.conv3x3(dy, ctx.w_bwd, stride=1, pad=0, upsample=2, out_hw=ctx.in_hw)
        return dx, None, None, None, None, None, (dy[..., :ctx.n_out].reshape(-1, ctx.n_out) if ctx.has_res else None)


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, silu):
        y, stats = H.groupnorm(x, gamma, beta, eps, silu, return_stats=True)
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.eps, ctx.silu = eps, silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta = ctx.saved_tensors
        return H.groupnorm_bwd(x, dy.contiguous(), gamma, beta, ctx.eps, ctx.silu, stats), None, None, None, None


def vae_desc(cfg: W.VAEConfig) -> VaeDesc:
    d = VaeDesc()
    d.in_channels, d.ch, d.n_levels, d.num_res_blocks = cfg.in_channels, cfg.ch, len(cfg.ch_mult), cfg.num_res_blocks
    for i, m in enumerate(cfg.ch_mult):
        d.ch_mult[i] = m
    d.z_channels, d.embed_dim = cfg.z_channels, cfg.embed_dim
    return d


class _EncodeFn(torch.autograd.Function):
    """moments = Encoder(images) as ONE autograd node (tensor-level seam: NCHW images in, NCHW moments out)."""

    @staticmethod
    def forward(ctx, images, enc):
        B, Cin, Hh, Ww = images.shape
        x = torch.zeros((B, Hh, Ww, 32), device=images.device, dtype=torch.float16)     # NHWC, channels padded to the K granularity
        x[..., :Cin] = images.permute(0, 2, 3, 1)
        m, saved = enc.forward_nhwc(x)
        ctx.enc, ctx.saved, ctx.cin, ctx.in_dtype = enc, saved, Cin, images.dtype
        return m.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, d_m):
        dx = ctx.enc.backward_nhwc(ctx.saved, d_m.permute(0, 2, 3, 1).contiguous().float())
        return dx[..., :ctx.cin].permute(0, 3, 1, 2).to(ctx.in_dtype), None


class HipVAEEncoder:
    """images [B,3,H,W] in [-1,1] -> posterior moments [B, 2*embed_dim, H/8, W/8] fp32, differentiable w.r.t. the images, through
    the C-ABI network asd_vae_enc_* (csrc/net.hip)."""

    def __init__(self, params: P, cfg: Optional[W.VAEConfig] = None, device="cuda", use_graph: bool = False):
        from .engine import CNet

        self.cfg = cfg or W.VAEConfig()
        self.device = torch.device(device)
        self.net = CNet("vae_enc", vae_desc(self.cfg), W.pack_vae_encoder(params, self.cfg), self.device)
        self.tuned = set()       # input shapes whose GEMM shapes (forward and backward) have been through the autotuner

    def needs_tune(self, shape) -> bool:
        return H.AUTOTUNE and shape not in self.tuned and not torch.cuda.is_current_stream_capturing()

    def workspace_bytes(self, B, Hh, Ww, tune) -> int:
        nb = lib().asd_vae_enc_workspace_bytes(self.net.handle, i32(B), i32(Hh), i32(Ww), i32(int(tune)))
        if nb < 0:
            raise AsdError(lib().asd_last_error().decode())
        return nb

    def forward_nhwc(self, x: torch.Tensor):
        """x fp16 [B,H,W,32] -> (moments fp32 [B,H/8,W/8,2*embed_dim], saved).  asd_vae_enc_fwd leaves the activations its input
        gradient needs in the workspace; `saved` (workspace + shape) is what a backward pass needs."""
        B, Hh, Ww, _ = x.shape
        tune = self.needs_tune((B, Hh, Ww))
        ws = torch.empty(self.workspace_bytes(B, Hh, Ww, tune), dtype=torch.uint8, device=x.device)
        m = torch.empty((B, Hh // 8, Ww // 8, 2 * self.cfg.embed_dim), device=x.device, dtype=torch.float32)
        check(lib().asd_vae_enc_fwd(self.net.handle, ptr(x), i32(B), i32(Hh), i32(Ww), ptr(ws), C.c_int64(ws.numel()), ptr(m), i32(int(tune)), stream()))
        return m, (ws, (B, Hh, Ww), tune)

    def backward_nhwc(self, saved, d_moments: torch.Tensor) -> torch.Tensor:
        """d_moments fp32 [B,H/8,W/8,2*embed_dim] -> image gradient fp16 [B,H,W,32] (frozen weights: dgrad only)"""
        ws, (B, Hh, Ww), tune = saved
        dx = torch.empty((B, Hh, Ww, 32), device=d_moments.device, dtype=torch.float16)
        check(lib().asd_vae_enc_bwd(self.net.handle, ptr(d_moments), i32(B), i32(Hh), i32(Ww), ptr(ws), C.c_int64(ws.numel()), ptr(dx),
                                    i32(int(tune)), stream()))
        if tune:
            self.tuned.add((B, Hh, Ww))
        return dx

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        if not images.is_cuda:
            raise AsdError("the HIP VAE encoder needs device tensors (there is no CPU fallback)")
        return _EncodeFn.apply(images, self)
