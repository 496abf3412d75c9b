"""HIP execution engine of the frozen SD-2.1 UNet (eps-prediction) for the ASD step.

Drives the hand-written gfx950 kernels of libasd_hip.so (asd_gemm_f16 / asd_groupnorm_f16 / asd_layernorm_f16 /
asd_geglu_f16 / asd_attention_f16 ...) layer by layer in the order of UNetModel.forward
(extern/mvdream/ldm/modules/diffusionmodules/openaimodel.py:771-808; ResBlock :252-275; SpatialTransformer
attention.py:320-340; BasicTransformerBlock :270-275), on NHWC fp16 activations with fp32 accumulation — the
reference runs this network in fp16 through diffusers (stable_diffusion_asd_guidance.py:38,57-59,319-331).

MI355X-first choices: weights are packed once (conv -> [Cout][ky][kx][Cin], q|k fused, all 22 time-embedding
projections fused into one GEMM, V projections emitted transposed for the attention kernel); the ~450 kernel
launches of one forward are captured into a HIP graph per input shape and replayed, so the host never sits
between two kernels; activations of one forward (~1 GB at batch 5) simply stay resident in HBM.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from ..guidance import DiffusionBackend, register_backend
from . import hip_ops as H
from . import weights as W

P = Dict[str, torch.Tensor]


class HipUNet:
    def __init__(self, params: P, cfg: Optional[W.UNetConfig] = None, device="cuda", use_graph: bool = True):
        self.cfg = cfg or W.UNetConfig()
        if self.cfg.num_head_channels != 64:
            raise NotImplementedError("the attention kernel is built for head_dim 64")
        self.device = torch.device(device)
        self.shapes, self.inputs, self.middle, self.outputs = W.unet_layout(self.cfg)
        self.use_graph = use_graph
        self._graphs: Dict[Tuple, Tuple] = {}
        self._num_frames = 1
        self._pack(params)

    # ---- weight packing ---------------------------------------------------------------------------
    def _pack(self, p: P):
        dev = self.device
        f16 = lambda t: t.to(device=dev, dtype=torch.float16).contiguous()
        w: Dict[str, torch.Tensor] = {}
        emb_w, emb_b, self.emb_slices = [], [], {}
        ck_w, cv_w, self.ctx_slices = [], [], {}
        off = coff = 0
        for name, t in p.items():
            if name.endswith(".weight") and t.ndim == 4:
                if t.shape[-1] == 3:
                    w[name] = H.pack_conv3x3_weight(f16(t))
                else:
                    w[name] = f16(t.reshape(t.shape[0], t.shape[1]))  # 1x1 conv == Linear on NHWC
            elif ".emb_layers.1." in name:
                continue
            elif ".attn1.to_q." in name or ".attn1.to_k." in name or ".attn2.to_k." in name or ".attn2.to_v." in name:
                continue
            else:
                w[name] = f16(t)
        for blk in list(self.inputs) + [self.middle] + list(self.outputs):
            for kind, name, cin, cout in blk.layers:
                if kind == "res":
                    emb_w.append(p[name + ".emb_layers.1.weight"])
                    emb_b.append(p[name + ".emb_layers.1.bias"])
                    self.emb_slices[name] = (off, cout)
                    off += cout
                elif kind == "attn":
                    for d in range(self.cfg.transformer_depth):
                        b = f"{name}.transformer_blocks.{d}.attn1"
                        w[b + ".to_qk.weight"] = f16(torch.cat([p[b + ".to_q.weight"], p[b + ".to_k.weight"]], 0))
                        b2 = f"{name}.transformer_blocks.{d}.attn2"
                        ck_w.append(p[b2 + ".to_k.weight"])
                        cv_w.append(p[b2 + ".to_v.weight"])
                        self.ctx_slices[b2] = (coff, cout)
                        coff += cout
        for k in [k for k in w if k.endswith(".ff.net.0.proj.weight")]:
            kb = k[:-len("weight")] + "bias"
            w[k], w[kb] = H.pack_geglu_weight(w[k], w[kb])
        w["emb_all.weight"] = f16(torch.cat(emb_w, 0))
        w["emb_all.bias"] = f16(torch.cat(emb_b, 0))
        # the context is the same for every cross-attention: all 16 K and V^T projections are two GEMMs per forward
        w["ctx_k_all.weight"] = f16(torch.cat(ck_w, 0))
        w["ctx_v_all.weight"] = f16(torch.cat(cv_w, 0))
        self.w = w

    # ---- layers -----------------------------------------------------------------------------------------
    def _res(self, name, x, emb_all, B, hw):
        """x: [B*hw, Cin] (NHWC flattened). Returns [B*hw, Cout]."""
        w = self.w
        Hh, Ww = hw
        cin = x.shape[-1]
        off, cout = self.emb_slices[name]
        h = H.groupnorm(x.view(B, Hh * Ww, cin), w[name + ".in_layers.0.weight"], w[name + ".in_layers.0.bias"], 1e-5, True)
        h = H.conv3x3(h.view(B, Hh, Ww, cin), w[name + ".in_layers.2.weight"], bias=w[name + ".in_layers.2.bias"],
                      row_bias=emb_all[:, off:off + cout], rows_per_group=Hh * Ww)
        h = H.groupnorm(h.view(B, Hh * Ww, cout), w[name + ".out_layers.0.weight"], w[name + ".out_layers.0.bias"], 1e-5, True)
        if name + ".skip_connection.weight" in w:
            skip = H.gemm(x, w[name + ".skip_connection.weight"], bias=w[name + ".skip_connection.bias"])
        else:
            skip = x
        h = H.conv3x3(h.view(B, Hh, Ww, cout), w[name + ".out_layers.3.weight"], bias=w[name + ".out_layers.3.bias"], residual=skip)
        return h.view(B * Hh * Ww, cout)

    def _transformer(self, name, x, ctx_pad, B, hw, n_ctx, ctx_stride):
        w = self.w
        L = hw[0] * hw[1]
        F_ = self._num_frames  # MVDream: self-attention spans the F views of a group (attention.py:348-354)
        C = x.shape[-1]
        heads = C // 64
        h = H.groupnorm(x.view(B, L, C), w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-6, False).view(B * L, C)
        h = H.gemm(h, w[name + ".proj_in.weight"], bias=w[name + ".proj_in.bias"])
        for d in range(self.cfg.transformer_depth):
            b = f"{name}.transformer_blocks.{d}"
            y = H.layernorm(h, w[b + ".norm1.weight"], w[b + ".norm1.bias"])
            qk = H.gemm(y, w[b + ".attn1.to_qk.weight"])                       # [M, 2C]
            vT = H.gemm(w[b + ".attn1.to_v.weight"], y)                         # [C, M] = V^T (operands swapped)
            o = H.attention(qk[:, :C], qk[:, C:], vT, B // F_, heads, F_ * L, F_ * L)
            h = H.gemm(o, w[b + ".attn1.to_out.0.weight"], bias=w[b + ".attn1.to_out.0.bias"], residual=h)
            y = H.layernorm(h, w[b + ".norm2.weight"], w[b + ".norm2.bias"])
            q = H.gemm(y, w[b + ".attn2.to_q.weight"])
            coff, _ = self.ctx_slices[b + ".attn2"]
            kc_all, vT_all = self._ctx_kv
            o = H.attention(q, kc_all[:, coff:coff + C], vT_all[coff:coff + C], B, heads, L, n_ctx, ctx_stride)
            h = H.gemm(o, w[b + ".attn2.to_out.0.weight"], bias=w[b + ".attn2.to_out.0.bias"], residual=h)
            y = H.layernorm(h, w[b + ".norm3.weight"], w[b + ".norm3.bias"])
            g = H.gemm(y, w[b + ".ff.net.0.proj.weight"], bias=w[b + ".ff.net.0.proj.bias"], act=2)   # GEGLU fused in the epilogue
            h = H.gemm(g, w[b + ".ff.net.2.weight"], bias=w[b + ".ff.net.2.bias"], residual=h)
        return H.gemm(h, w[name + ".proj_out.weight"], bias=w[name + ".proj_out.bias"], residual=x)

    def _apply(self, layers, h, emb_all, ctx_pad, B, hw, n_ctx, ctx_stride):
        w = self.w
        for kind, name, cin, cout in layers:
            if kind == "conv":
                h = H.conv3x3(h.view(B, hw[0], hw[1], h.shape[-1]), w[name + ".weight"], bias=w[name + ".bias"]).view(-1, cout)
            elif kind == "res":
                h = self._res(name, h, emb_all, B, hw)
            elif kind == "attn":
                h = self._transformer(name, h, ctx_pad, B, hw, n_ctx, ctx_stride)
            elif kind == "down":
                h = H.conv3x3(h.view(B, hw[0], hw[1], cin), w[name + ".weight"], bias=w[name + ".bias"], stride=2)
                hw = (h.shape[1], h.shape[2])
                h = h.view(-1, cout)
            elif kind == "up":
                h = H.conv3x3(h.view(B, hw[0], hw[1], cin), w[name + ".weight"], bias=w[name + ".bias"], upsample=True)
                hw = (h.shape[1], h.shape[2])
                h = h.view(-1, cout)
        return h, hw

    # ---- forward ------------------------------------------------------------------------------------------
    def _forward_impl(self, x_nhwc32: torch.Tensor, t: torch.Tensor, ctx_pad: torch.Tensor, n_ctx: int, ctx_stride: int,
                      camera: Optional[torch.Tensor] = None):
        cfg, w = self.cfg, self.w
        B, Hh, Ww, _ = x_nhwc32.shape
        t_emb = H.timestep_embedding(t, cfg.model_channels)
        e = H.gemm(t_emb, w["time_embed.0.weight"], bias=w["time_embed.0.bias"], act=1)
        e = H.gemm(e, w["time_embed.2.weight"], bias=w["time_embed.2.bias"])
        if camera is not None:  # MultiViewUNetModel: emb += camera_embed(camera)  (openaimodel.py:1197-1200)
            c = H.gemm(camera, w["camera_embed.0.weight"], bias=w["camera_embed.0.bias"], act=1)
            e = H.gemm(c, w["camera_embed.2.weight"], bias=w["camera_embed.2.bias"], residual=e)
        emb_all = H.gemm(H.silu(e), w["emb_all.weight"], bias=w["emb_all.bias"])     # every ResBlock's emb_layers at once
        self._ctx_kv = (H.gemm(ctx_pad, w["ctx_k_all.weight"]),       # [B*ctx_stride, sum C]
                        H.gemm(w["ctx_v_all.weight"], ctx_pad))       # [sum C, B*ctx_stride] = every V^T
        hs: List[Tuple[torch.Tensor, Tuple[int, int]]] = []
        h, hw = x_nhwc32.reshape(B * Hh * Ww, 32), (Hh, Ww)
        for blk in self.inputs:
            h, hw = self._apply(blk.layers, h, emb_all, ctx_pad, B, hw, n_ctx, ctx_stride)
            hs.append((h, hw))
        h, hw = self._apply(self.middle.layers, h, emb_all, ctx_pad, B, hw, n_ctx, ctx_stride)
        for blk in self.outputs:
            skip, _ = hs.pop()
            h = H.concat(h, skip)
            h, hw = self._apply(blk.layers, h, emb_all, ctx_pad, B, hw, n_ctx, ctx_stride)
        C = h.shape[-1]
        h = H.groupnorm(h.view(B, hw[0] * hw[1], C), w["out.0.weight"], w["out.0.bias"], 1e-5, True)
        out = H.conv3x3(h.view(B, hw[0], hw[1], C), w["out.2.weight"], bias=w["out.2.bias"], out_f32=True)
        return out  # [B, H, W, out_channels] fp32

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor, camera: Optional[torch.Tensor] = None,
                 num_frames: int = 1) -> torch.Tensor:
        """x [N,4,H,W], t [N], context [N,n_ctx,ctx_dim] (+ camera [N,16], num_frames for MVDream) -> eps [N,4,H,W]."""
        N, Cin, Hh, Ww = x.shape
        assert N % num_frames == 0, "[UNet] input batch size must be dividable by num_frames!"
        assert (camera is not None) == (self.cfg.camera_dim is not None), "camera is given iff the UNet is camera-conditioned"
        self._num_frames = num_frames if self.cfg.camera_dim is not None else 1
        n_ctx = context.shape[1]
        ctx_stride = (n_ctx + 7) // 8 * 8
        key = (N, Hh, Ww, n_ctx, self._num_frames)
        if not self.use_graph:
            xin, tin, cin, cam = self._stage_inputs(x, t, context, ctx_stride, camera)
            return self._forward_impl(xin, tin, cin, n_ctx, ctx_stride, cam).permute(0, 3, 1, 2).to(x.dtype)
        if key not in self._graphs:
            xin, tin, cin, cam = self._stage_inputs(x, t, context, ctx_stride, camera)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):            # warm-up outside the capture (allocator, lazy module load)
                self._forward_impl(xin, tin, cin, n_ctx, ctx_stride, cam)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._forward_impl(xin, tin, cin, n_ctx, ctx_stride, cam)
            self._graphs[key] = (g, xin, tin, cin, cam, out)
        g, xin, tin, cin, cam, out = self._graphs[key]
        self._write_inputs(xin, tin, cin, x, t, context, ctx_stride)
        if cam is not None:
            cam.copy_(camera)
        g.replay()
        return out.permute(0, 3, 1, 2).to(x.dtype)

    def _write_inputs(self, xin, tin, cin, x, t, context, ctx_stride):
        N, C = x.shape[:2]
        xin[..., :C].copy_(x.permute(0, 2, 3, 1))
        tin.copy_(t)
        cin.view(N, ctx_stride, -1)[:, :context.shape[1]].copy_(context)

    def _stage_inputs(self, x, t, context, ctx_stride, camera=None):
        N, _, Hh, Ww = x.shape
        xin = torch.zeros((N, Hh, Ww, 32), device=self.device, dtype=torch.float16)
        tin = torch.zeros(N, device=self.device, dtype=torch.float32)
        cin = torch.zeros((N * ctx_stride, context.shape[2]), device=self.device, dtype=torch.float16)
        self._write_inputs(xin, tin, cin, x, t, context, ctx_stride)
        cam = None if camera is None else camera.to(device=self.device, dtype=torch.float16).contiguous().clone()
        return xin, tin, cin, cam


class HipBackend(DiffusionBackend):
    """UNet eps-prediction and VAE encoder (forward + input gradient) on the hand-written HIP kernels."""

    def __init__(self, device, dtype=torch.float16, seed: int = 1, unet_cfg: Optional[W.UNetConfig] = None,
                 vae_cfg: Optional[W.VAEConfig] = None, unet_params: Optional[P] = None, vae_params: Optional[P] = None,
                 use_graph: bool = True):
        from .vae_hip import HipVAEEncoder

        self.unet_cfg = unet_cfg or W.UNetConfig()
        self.vae_cfg = vae_cfg or W.VAEConfig()
        self.scaling_factor = self.vae_cfg.scale_factor
        layout = W.unet_layout(self.unet_cfg)
        up = unet_params if unet_params is not None else W.gen_params(layout[0], seed)
        self.hip_unet = HipUNet(up, self.unet_cfg, device, use_graph=use_graph)
        del up
        vshapes, _ = W.vae_encoder_layout(self.vae_cfg)
        vp = vae_params if vae_params is not None else W.gen_params(vshapes, seed + 1)
        self.hip_vae = HipVAEEncoder(vp, self.vae_cfg, device)

    @torch.no_grad()
    def unet(self, latents, t, context, camera=None, num_frames: int = 1):
        return self.hip_unet(latents, t, context, camera=camera, num_frames=num_frames)

    def encode(self, images):
        return self.hip_vae(images)


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
