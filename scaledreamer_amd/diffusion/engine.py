"""Host side of the frozen SD-2.1 / MVDream prior on the HIP path: a thin caller of the C-ABI networks of libasd_hip.so.

The layer schedule of UNetModel.forward (extern/mvdream/ldm/modules/diffusionmodules/openaimodel.py:771-808; ResBlock :252-275;
SpatialTransformer attention.py:320-340; BasicTransformerBlock :270-275) is enqueued by the library itself (csrc/net.hip:
asd_unet_fwd, asd_vae_enc_fwd / _bwd) on NHWC fp16 activations with fp32 accumulation — the reference runs this network in fp16
through diffusers (stable_diffusion_asd_guidance.py:38,57-59,319-331).  What stays here: packing a name-keyed state dict into the
weight table the library publishes (weights.pack_unet: conv -> [Cout][ky][kx][Cin], q|k fused, all 22 time-embedding projections
fused into one GEMM, GEGLU rows interleaved), owning the workspace (activations of one forward, ~1 GB at batch 5, stay resident
in HBM) and capturing the ~450 launches of a forward into one HIP graph per input shape.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Tuple

import torch

import ctypes as C

from .._lib import AsdError, UNetDesc, WeightInfo, check, i32, lib, ptr, stream
from ..guidance import DiffusionBackend, register_backend
from . import hip_ops as H
from . import weights as W

P = Dict[str, torch.Tensor]


class CNet:
    """A frozen network held by the library (csrc/net.hip): created from its descriptor, bound to the packed weights its table asks
    for.  `kind` = "unet" | "vae_enc"."""

    def __init__(self, kind: str, desc, packed: P, device, dtype=torch.float16):
        self.kind, self.device = kind, torch.device(device)
        self._l = lib()
        h = C.c_void_p()
        check(getattr(self._l, f"asd_{kind}_create")(C.byref(desc), C.byref(h)))
        self.handle = h
        n = getattr(self._l, f"asd_{kind}_num_weights")(h)
        info = WeightInfo()
        self.tensors = []
        ptrs = (C.c_void_p * n)()
        for i in range(n):
            check(getattr(self._l, f"asd_{kind}_weight_info")(h, i32(i), C.byref(info)))
            name = info.name.decode()
            if name not in packed:
                raise KeyError(f"{kind}: the weight table asks for {name!r}, which the packed state dict does not hold")
            t = packed[name].to(device=self.device, dtype=dtype).contiguous()
            if t.numel() != info.rows * info.cols:
                raise ValueError(f"{kind}: {name} has {tuple(t.shape)}, the table needs [{info.rows}, {info.cols}]")
            self.tensors.append(t)
            ptrs[i] = t.data_ptr()
        check(getattr(self._l, f"asd_{kind}_bind_weights")(h, ptrs, i32(n)))

    def __del__(self):
        try:
            getattr(self._l, f"asd_{self.kind}_destroy")(self.handle)
        except Exception:
            pass


def unet_desc(cfg: W.UNetConfig) -> UNetDesc:
    d = UNetDesc()
    d.in_channels, d.out_channels, d.model_channels, d.num_res_blocks = cfg.in_channels, cfg.out_channels, cfg.model_channels, cfg.num_res_blocks
    d.n_levels = len(cfg.channel_mult)
    for i, m in enumerate(cfg.channel_mult):
        d.channel_mult[i] = m
    mask = 0
    for ds in cfg.attention_resolutions:
        mask |= 1 << (int(ds).bit_length() - 1)
    d.attention_ds_mask = mask
    d.num_head_channels, d.transformer_depth, d.context_dim = cfg.num_head_channels, cfg.transformer_depth, cfg.context_dim
    d.camera_dim = cfg.camera_dim or 0
    return d


class HipUNet:
    """eps = UNet(x, t, context[, camera]) through asd_unet_fwd.  Python only stages the inputs and owns the workspace; the ~450
    launches of a forward are enqueued by the library and captured once per input shape into a HIP graph."""

    def __init__(self, params: P, cfg: Optional[W.UNetConfig] = None, device="cuda", use_graph: bool = True):
        self.cfg = cfg or W.UNetConfig()
        if self.cfg.num_head_channels != 64:
            raise NotImplementedError("the attention kernel is built for head_dim 64")
        self.device = torch.device(device)
        self.use_graph = use_graph and os.environ.get("ASD_UNET_GRAPH", "1") != "0"     # 0: eager launches (tools/gemm_shapes.py traces them)
        self._graphs: Dict[Tuple, Tuple] = {}
        self.net = CNet("unet", unet_desc(self.cfg), W.pack_unet(params, self.cfg), self.device)

    # ---- one forward on staged inputs ---------------------------------------------------------------------------------------
    def _workspace_bytes(self, N, Hh, Ww, n_ctx, frames, tune, n_uniq: int = 0) -> int:
        nb = lib().asd_unet_workspace_bytes_shared(self.net.handle, i32(N), i32(Hh), i32(Ww), i32(n_ctx), i32(frames), i32(int(tune)), i32(n_uniq))
        if nb < 0:
            raise AsdError(lib().asd_last_error().decode())
        return nb

    def _run(self, st, tune: bool):
        xin, tin, cin, cam, out, ws, (N, Hh, Ww, n_ctx, frames, reps), share = st
        uniq, expand = share if share is not None else (None, None)
        n_uniq = 0 if uniq is None else uniq.numel()
        if tune:    # time GEMM shapes that have no plan yet (never under capture); needs the larger tuning workspace
            ws = torch.empty(self._workspace_bytes(N, Hh, Ww, n_ctx, frames, True, n_uniq), dtype=torch.uint8, device=self.device)
        check(lib().asd_unet_fwd_shared(self.net.handle, ptr(xin), ptr(tin), ptr(cin), ptr(cam), i32(N), i32(Hh), i32(Ww), i32(n_ctx), i32(frames),
                                        ptr(uniq), ptr(expand), i32(n_uniq), ptr(ws), C.c_int64(ws.numel()), ptr(out), i32(int(tune)), stream()))
        return out

    def staging(self, N: int, Hh: int, Ww: int, n_ctx: int, frames: int = 1, shared_reps: int = 0):
        """persistent input / output buffers of one input shape: x [N,H,W,32] fp16, t [N] fp32, context [N*ctx_stride, ctx_dim]
        fp16 (padding rows stay zero), camera [N,16] fp16, eps [N,H,W,out] fp32.  Callers may write them in place (the fused ASD
        kernels do) and call replay().
        shared_reps = r >= 2 declares the batch layout of the ASD step, [r repetitions of G entries | G entries] with N = (r + 1) G,
        where the repetitions carry the SAME (x, t, camera) under different text contexts (asd_latents_fwd writes it so): the
        network computes everything in front of its first cross-attention once per distinct input (asd_unet_fwd_shared)."""
        key = (N, Hh, Ww, n_ctx, frames, int(shared_reps) if shared_reps >= 2 and N % (shared_reps + 1) == 0 else 0)
        if key not in self._graphs:
            dev, ctx_stride = self.device, (n_ctx + 7) // 8 * 8
            xin = torch.zeros((N, Hh, Ww, 32), device=dev, dtype=torch.float16)
            tin = torch.zeros(N, device=dev, dtype=torch.float32)
            cin = torch.zeros((N * ctx_stride, self.cfg.context_dim), device=dev, dtype=torch.float16)
            cam = torch.zeros((N, self.cfg.camera_dim), device=dev, dtype=torch.float16) if self.cfg.camera_dim else None
            out = torch.empty((N, Hh, Ww, self.cfg.out_channels), device=dev, dtype=torch.float32)
            share = None
            if key[5]:
                G = N // (key[5] + 1)
                uniq = list(range(G)) + [key[5] * G + j for j in range(G)]
                expand = [i % G for i in range(key[5] * G)] + [G + j for j in range(G)]
                if (2 * G) % frames == 0:
                    share = (torch.tensor(uniq, device=dev, dtype=torch.int32), torch.tensor(expand, device=dev, dtype=torch.int32))
            st = [xin, tin, cin, cam, out, None, key, share]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):            # warm-up outside the capture: lazy module load + tuning of new GEMM shapes
                self._run(st, tune=H.AUTOTUNE)
            torch.cuda.current_stream().wait_stream(side)
            st[5] = torch.empty(self._workspace_bytes(N, Hh, Ww, n_ctx, frames, False, 0 if share is None else share[0].numel()),
                                dtype=torch.uint8, device=dev)
            g = None
            if self.use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._run(st, tune=False)
            self._graphs[key] = (g, st)
        return self._graphs[key]

    def replay(self, key) -> torch.Tensor:
        g, st = self._graphs[key]
        if g is not None:
            g.replay()
        else:
            self._run(st, tune=False)
        return st[4]

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor, camera: Optional[torch.Tensor] = None,
                 num_frames: int = 1) -> torch.Tensor:
        """x [N,4,H,W], t [N], context [N,n_ctx,ctx_dim] (+ camera [N,16], num_frames for MVDream) -> eps [N,4,H,W]."""
        N, Cin, Hh, Ww = x.shape
        assert N % num_frames == 0, "[UNet] input batch size must be dividable by num_frames!"
        assert (camera is not None) == (self.cfg.camera_dim is not None), "camera is given iff the UNet is camera-conditioned"
        frames = num_frames if self.cfg.camera_dim is not None else 1
        n_ctx = context.shape[1]
        key = (N, Hh, Ww, n_ctx, frames)
        g_st = self.staging(*key)
        key = g_st[1][6]
        xin, tin, cin, cam, out = g_st[1][:5]
        xin[..., :Cin].copy_(x.permute(0, 2, 3, 1))
        tin.copy_(t)
        cin.view(N, -1, cin.shape[-1])[:, :n_ctx].copy_(context)
        if cam is not None:
            cam.copy_(camera)
        return self.replay(key).permute(0, 3, 1, 2).to(x.dtype)


class HipBackend(DiffusionBackend):
    """UNet eps-prediction and VAE encoder (forward + input gradient) as C-ABI networks on the hand-written HIP kernels."""

    def __init__(self, device, dtype=torch.float16, seed: int = 1, unet_cfg: Optional[W.UNetConfig] = None,
                 vae_cfg: Optional[W.VAEConfig] = None, unet_params: Optional[P] = None, vae_params: Optional[P] = None,
                 use_graph: bool = True):
        from .vae_hip import HipVAEEncoder

        self.device = torch.device(device)
        self.unet_cfg = unet_cfg or W.UNetConfig()
        self.vae_cfg = vae_cfg or W.VAEConfig()
        self.scaling_factor = self.vae_cfg.scale_factor
        self.context_dim, self.camera_dim = self.unet_cfg.context_dim, self.unet_cfg.camera_dim or 0
        layout = W.unet_layout(self.unet_cfg)
        up = unet_params if unet_params is not None else W.gen_params(layout[0], seed)
        self.hip_unet = HipUNet(up, self.unet_cfg, device, use_graph=use_graph)
        del up
        vshapes, _ = W.vae_encoder_layout(self.vae_cfg)
        vp = vae_params if vae_params is not None else W.gen_params(vshapes, seed + 1)
        self.hip_vae = HipVAEEncoder(vp, self.vae_cfg, device)

    # tensor-level seams (tests, tools)
    @torch.no_grad()
    def unet(self, latents, t, context, camera=None, num_frames: int = 1):
        return self.hip_unet(latents, t, context, camera=camera, num_frames=num_frames)

    def encode(self, images):
        return self.hip_vae(images)

    # buffer protocol of the fused guidance: no re-layout between the ASD kernels and the networks
    def vae_forward(self, x_nhwc32):
        return self.hip_vae.forward_nhwc(x_nhwc32)

    def vae_backward(self, saved, d_moments_nhwc):
        return self.hip_vae.backward_nhwc(saved, d_moments_nhwc)

    def unet_buffers(self, N, hl, wl, n_ctx, frames=1, shared_reps: int = 0):
        from ..guidance import UNetIO

        _, st = self.hip_unet.staging(N, hl, wl, n_ctx, frames if self.camera_dim else 1, shared_reps)
        xin, tin, cin, cam, out = st[:5]
        return UNetIO(st[6], xin, tin, cin, cam, out)

    def unet_run(self, io):
        return self.hip_unet.replay(io.key)


def _backend_from_cfg(cfg, device, dtype, unet_cfg: W.UNetConfig) -> HipBackend:
    """weights come from `cfg.ckpt_path` / `cfg.pretrained_model_name_or_path` (checkpoint.resolve_params); seeded random
    weights only when the config says `allow_random_weights` — never silently."""
    from . import checkpoint

    vae_cfg = W.VAEConfig()
    up, vp, what = checkpoint.resolve_params(cfg, unet_cfg, vae_cfg)
    backend = HipBackend(device, dtype, seed=getattr(cfg, "weights_seed", 1), unet_cfg=unet_cfg, vae_cfg=vae_cfg,
                         unet_params=up, vae_params=vp)
    backend.weights_source = what
    return backend


@register_backend("hip")
def _make_hip(cfg, device, dtype):
    return _backend_from_cfg(cfg, device, dtype, W.UNetConfig())


@register_backend("hip-mvdream")
def _make_hip_mvdream(cfg, device, dtype):
    """MVDream: the SD-2.1 UNet with camera_dim = 16 (extern/mvdream/configs/sd-v2-base.yaml:13-27)."""
    return _backend_from_cfg(cfg, device, dtype, W.UNetConfig(camera_dim=16))
