"""Generator-backed SDF geometries of the multi-prompt configs on the HIP samplers:
  `3DConv-net`               custom/amortized/models/geometry/stylegan_3dconv_net.py:21-422  (feature volume [B,32,128^3] from the
                             StyleGAN-3D generator, trilinear lookups)
  `Triplane-transformer-sdf` custom/amortized/models/geometry/triplane_transformer.py:20-315 (three [32,64,64] planes from the
                             transformer, 3 bilinear lookups concatenated)
Both: contract to [-1,1] -> sample features -> VanillaMLP sdf / feature heads -> sdf + sphere bias -> finite-difference
sdf_grad / normal from 3 offset lookups.  Shipped configurations run lookup + heads + bias + finite differences as ONE fused HIP kernel each
way (asd_voxfield_* / asd_trifield_*, `_VoxFieldFn` / `_TriFieldFn` below); the composed form (HIP samplers of samplers.py + torch MLP heads) is
the fallback for head shapes the fused kernels are not instantiated for and the A/B partner (ASD_VOXFIELD=0 / ASD_TRIFIELD=0).  The generators
live in generators.py (Generator3D on csrc/conv3d.hip, TriplaneTransformer on csrc/transformer.hip).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Union

import torch
import torch.nn.functional as F

from . import _lib, ops
from .config import C
from .generators import Generator3D, TriplaneTransformer
from .geometry import BaseImplicitGeometry
from .networks import get_activation, get_mlp
from .registry import register
from .samplers import channels_last, contract_to_unisphere_custom, get_trilinear_feature, planes_channels_last, sample_from_planes

_MLP1 = {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64, "n_hidden_layers": 1}
_MLP2 = {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64, "n_hidden_layers": 2}


class _SampledSdfGeometry(BaseImplicitGeometry):
    """shared forward of the two classes (their bodies are the same code in the reference, stylegan_3dconv_net.py:259-346 and
    triplane_transformer.py:156-240)"""

    def _heads(self, input_dim: int) -> None:
        self.sdf_network = get_mlp(input_dim, 1, self.cfg.mlp_network_config)
        if self.cfg.n_feature_dims > 0:
            self.feature_network = get_mlp(input_dim, self.cfg.n_feature_dims, self.cfg.mlp_network_config)
        if self.cfg.normal_type == "pred":
            raise NotImplementedError("normal_type == pred is not implemented yet.")
        if self.cfg.isosurface_deformable_grid:
            assert self.cfg.isosurface_method == "mt", "isosurface_deformable_grid only works with mt"
            self.deformation_network = get_mlp(input_dim, 3, self.cfg.mlp_network_config)
        self.finite_difference_normal_eps: Optional[float] = None

    def interpolate_encodings(self, points: torch.Tensor, space_cache: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def get_shifted_sdf(self, points, sdf):
        c = self.cfg
        if c.sdf_bias == "ellipsoid":
            assert len(c.sdf_bias_params) == 3
            size = torch.as_tensor(c.sdf_bias_params).to(points)
            bias = ((points / size) ** 2).sum(dim=-1, keepdim=True).sqrt() - 1.0
        elif c.sdf_bias == "sphere":
            assert isinstance(c.sdf_bias_params, float)
            bias = (points ** 2).sum(dim=-1, keepdim=True).sqrt() - c.sdf_bias_params
        elif isinstance(c.sdf_bias, float):
            bias = c.sdf_bias
        else:
            raise ValueError(f"Unknown sdf bias {c.sdf_bias}")
        return sdf + bias

    # COMPOSED path only (the fused kernels keep nothing per evaluation but the points, so they never chunk): above this many points per
    # call the torch MLP heads run chunk by chunk under activation checkpointing — they keep ~1.2 KB of autograd state per point and SDF
    # evaluation (4 evaluations per point with the finite-difference normal), i.e. 245 GB for the 50.6 M samples of a 256 x 256 x 4-view
    # step.  Chunks are recomputed in the backward pass (one extra forward); values and gradients are those of the un-chunked graph.
    CHECKPOINT_ABOVE = 8 * 1024 * 1024
    CHECKPOINT_CHUNK = 2 * 1024 * 1024

    def _use_fused(self, points) -> bool:
        return False

    def forward(self, points: torch.Tensor, space_cache: Any, output_normal: bool = False) -> Dict[str, torch.Tensor]:
        batch_size, n_points, _ = points.shape
        if torch.is_grad_enabled() and batch_size * n_points > self.CHECKPOINT_ABOVE and not self._use_fused(points):
            from torch.utils.checkpoint import checkpoint

            per = max(1, self.CHECKPOINT_CHUNK // batch_size)
            parts = [checkpoint(self._forward_points, points[:, i:i + per], space_cache, output_normal, use_reentrant=False)
                     for i in range(0, n_points, per)]
            return {k: torch.cat([p[k] for p in parts], dim=1).reshape(batch_size * n_points, -1) for k in parts[0]}
        return {k: v.reshape(batch_size * n_points, -1) for k, v in self._forward_points(points, space_cache, output_normal).items()}

    def _forward_points(self, points: torch.Tensor, space_cache: Any, output_normal: bool) -> Dict[str, torch.Tensor]:
        """SDF / features (/ finite-difference normal) of [batch, points, 3] -> outputs as [batch, points, k]; the public forward flattens
        them batch-major (triplane_transformer.py:153-200, stylegan_3dconv_net.py forward).  With a normal the centre and its three
        +eps probes go through the feature sampler and the SDF head as ONE stencil of 4 points per sample — one sampler launch and one
        head pass over 4 N points instead of one over N plus one over 3 N; per point the arithmetic is that of forward_sdf."""
        if output_normal and self.cfg.normal_type != "finite_difference":
            raise NotImplementedError(f"normal_type {self.cfg.normal_type!r}: only the finite-difference normal of the shipped configs is implemented")
        batch, n = points.shape[:2]
        query = points
        if output_normal:
            eps = self.finite_difference_normal_eps
            assert eps is not None, "update_step() sets finite_difference_normal_eps before the first forward"
            probes = (points[:, :, None, :] + eps * torch.eye(3, dtype=points.dtype, device=points.device)).clamp(-self.cfg.radius, self.cfg.radius)
            query = torch.cat([points[:, :, None, :], probes], dim=2).reshape(batch, 4 * n, 3)         # [centre, +x, +y, +z] per sample
        enc = self.interpolate_encodings(contract_to_unisphere_custom(query, self.bbox, self.unbounded), space_cache)
        sdf_all = self.get_shifted_sdf(query, self.sdf_network(enc).view(batch, -1, 1))
        if not output_normal:
            out = {"sdf": sdf_all}
            if self.cfg.n_feature_dims > 0:
                out["features"] = self.feature_network(enc).view(batch, n, self.cfg.n_feature_dims)
            return out
        stencil = sdf_all.view(batch, n, 4)
        out = {"sdf": stencil[..., :1]}
        if self.cfg.n_feature_dims > 0:
            centre_enc = enc.view(batch, n, 4, -1)[:, :, 0]
            out["features"] = self.feature_network(centre_enc).view(batch, n, self.cfg.n_feature_dims)
        sdf_grad = (stencil[..., 1:] - stencil[..., :1]) / eps
        normal = F.normalize(sdf_grad, dim=-1)
        out.update(normal=normal, shading_normal=normal, sdf_grad=sdf_grad)
        return out

    @staticmethod
    def _gate(space_cache, cache_cl, b):
        """(cache as the fused field node reads it, entry, gradient slot) — see _GradSlot"""
        if not cache_cl.requires_grad:
            return cache_cl, b, None
        gated, slot = _gate_of(space_cache, cache_cl)
        return gated, b, slot

    def forward_sdf(self, points: torch.Tensor, space_cache: Any) -> torch.Tensor:
        batch_size = points.shape[0]
        pts = contract_to_unisphere_custom(points, self.bbox, self.unbounded)
        enc = self.interpolate_encodings(pts.reshape(batch_size, -1, 3), space_cache).reshape(*pts.shape[:-1], -1)
        return self.get_shifted_sdf(points, self.sdf_network(enc).reshape(*pts.shape[:-1], 1))

    def forward_field(self, points, space_cache):
        pts = contract_to_unisphere_custom(points, self.bbox, self.unbounded)
        enc = self.interpolate_encodings(pts, space_cache)
        sdf = self.get_shifted_sdf(points, self.sdf_network(enc).reshape(*pts.shape[:-1], 1))
        deformation = self.deformation_network(enc).reshape(*pts.shape[:-1], 3) if self.cfg.isosurface_deformable_grid else None
        return sdf, deformation

    def forward_level(self, field, threshold):
        return field - threshold

    def export(self, points, space_cache, **kwargs) -> Dict[str, Any]:
        if self.cfg.n_feature_dims == 0:
            return {}
        pts = contract_to_unisphere_custom(points, self.bbox, self.unbounded)
        enc = self.interpolate_encodings(pts, space_cache)
        return {"features": self.feature_network(enc).view(*pts.shape[:-1], self.cfg.n_feature_dims)}

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        if self.cfg.normal_type != "finite_difference":
            raise NotImplementedError(f"normal_type == {self.cfg.normal_type} is not implemented yet.")
        if isinstance(self.cfg.finite_difference_normal_eps, float):
            self.finite_difference_normal_eps = self.cfg.finite_difference_normal_eps


class _GradSlot:
    """ONE gradient buffer per feature volume and backward pass.  The VolSDF renderer evaluates the field chunk by chunk and pass by pass
    (proposal sdf, main samples): every evaluation is its own autograd node, and a node that returned a fresh zeros_like(volume) — 268 MB at
    128^3 x 32 — had autograd memset, fill and sum one volume per chunk.  The scatter kernels accumulate, so the first node of a backward
    pass allocates the buffer and hands it to autograd, the later ones add into it in place and return None.

    The invariant that makes this sound is structural since round 6 (`_CacheGate` below): the field nodes never see the cache itself but the
    output of ONE identity node per cache whose only consumers they are.  The gate's gradient edge therefore receives exactly one defined
    gradient — the buffer — and the gate runs after every field node has run (autograd's dependency count), checks that what arrives IS the
    buffer, hands it on and drops the slot's reference.  Other consumers of the cache (a regulariser on the planes, a second relayout) meet the
    gate's result at the cache's own edge, after the last scatter, through autograd's normal accumulation."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def acquire(self, like: torch.Tensor):
        """(buffer, first): `first` tells the caller to return the buffer as its gradient"""
        if self.buf is None or self.buf.shape != like.shape:
            self.buf = torch.zeros_like(like)
            return self.buf, True
        return self.buf, False

    def release(self):
        self.buf = None


class _CacheGate(torch.autograd.Function):
    """identity on the channel-last cache; its output is private to the fused field nodes of this cache (see _GradSlot)"""

    @staticmethod
    def forward(ctx, cache_cl, slot):
        ctx.slot = slot
        return cache_cl.view_as(cache_cl)

    @staticmethod
    def backward(ctx, grad):
        buf = ctx.slot.buf
        ctx.slot.release()                               # the next backward pass over this graph (retain_graph) starts from a fresh buffer
        if grad is not None and buf is not None and grad.data_ptr() != buf.data_ptr():
            raise RuntimeError("fused field: the gated cache alias received a gradient that is not the shared buffer — the alias is private "
                               "to the field nodes, read the cache itself instead")
        return grad, None


def _gate_of(owner: torch.Tensor, cache_cl: torch.Tensor):
    """(gated alias of the channel-last cache, its slot): one gate per cache tensor and autograd graph, kept on the cache object itself"""
    st = owner.__dict__.get("_asd_gate")
    if st is None or st[2] != owner._version or st[0].shape != cache_cl.shape:
        slot = _GradSlot()
        st = owner.__dict__["_asd_gate"] = (_CacheGate.apply(cache_cl, slot), slot, owner._version)
    return st[0], st[1]


class _VoxFieldFn(torch.autograd.Function):
    """(sdf, features, normal, sdf_grad) of ONE batch entry from its points, its channel-last feature volume and the two MLP heads: trilinear
    lookup, heads, bias and finite differences in one kernel each way (include/asd_hip.h: asd_voxfield_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, points, vol_cl, b, slot, w1s, w2s, w1f, w2f, fcfg, want_normal):
        """vol_cl: the WHOLE channel-last cache [B, D, H, W, C] (contiguous), b: the entry these points sample"""
        sdf, feats, normal, fdg, enc = ops.voxfield_fwd(vol_cl[b], fcfg, w1s, w2s, w1f, w2f, points, want_normal)
        if not want_normal:
            normal, fdg = sdf.new_zeros(0), sdf.new_zeros(0)
            ctx.mark_non_differentiable(normal, fdg)
        ctx.save_for_backward(points, vol_cl, w1s, w2s, w1f, w2f, enc, sdf)
        ctx.fcfg, ctx.want_normal, ctx.slot, ctx.b = fcfg, want_normal, slot, b
        ctx.set_materialize_grads(False)
        return sdf, feats, normal, fdg

    @staticmethod
    def backward(ctx, d_sdf, d_feats, d_normal, d_fdg):
        points, vol_cl, w1s, w2s, w1f, w2f, enc, sdf = ctx.saved_tensors
        if d_sdf is None and d_feats is None and d_normal is None and d_fdg is None:
            return (None,) * 10
        d_vol, first = (ctx.slot or _GradSlot()).acquire(vol_cl)
        c = lambda t: None if t is None else t.contiguous()
        dw = ops.voxfield_bwd(vol_cl[ctx.b], ctx.fcfg, w1s, w2s, w1f, w2f, points, enc, sdf, c(d_sdf), c(d_feats),
                              c(d_normal) if ctx.want_normal else None, c(d_fdg) if ctx.want_normal else None, d_vol[ctx.b])
        return None, (d_vol if first else None), None, None, dw[0], dw[1], dw[2], dw[3], None, None


@register("3DConv-net")
class Voxel_3d_Sdf(_SampledSdfGeometry):
    @dataclass
    class Config(BaseImplicitGeometry.Config):
        n_input_dims: int = 3
        n_feature_dims: int = 3
        space_generator_config: dict = field(default_factory=lambda: {
            "z_dim": 512, "w_dim": 512, "num_layers": 2, "img_resolution": 128, "img_channels": 32, "channel_multiplier": 1})
        mlp_network_config: dict = field(default_factory=lambda: dict(_MLP1))
        pos_encoding_config: dict = field(default_factory=lambda: {
            "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
            "per_level_scale": 1.447269237440378})
        backbone: str = "3dconv_net"
        truncation_psi: Any = 1.0
        activation: str = "none"
        normal_type: Optional[str] = "finite_difference"
        finite_difference_normal_eps: Union[float, str] = 0.01
        shape_init: Optional[str] = None
        shape_init_params: Optional[Any] = None
        shape_init_mesh_up: str = "+z"
        shape_init_mesh_front: str = "+x"
        force_shape_init: bool = False
        sdf_bias: Union[float, str] = 0.0
        sdf_bias_params: Optional[Any] = None
        isosurface_remove_outliers: bool = False

    cfg: Config

    def configure(self) -> None:
        super().configure()
        if self.cfg.backbone != "3dconv_net":
            raise ValueError(f"Unknown backbone {self.cfg.backbone}")
        self.space_generator = Generator3D(**self.cfg.space_generator_config)
        self._heads(self.cfg.space_generator_config["img_channels"])
        self.noise_dim = self.cfg.space_generator_config["z_dim"]
        self.truncation_psi = 1.0

    def initialize_shape(self) -> None:
        if self.cfg.shape_init is None and not self.cfg.force_shape_init:
            return
        if self.cfg.weights is not None and not self.cfg.force_shape_init:
            return
        raise NotImplementedError("SDF pre-fitting (stylegan_3dconv_net.py:139-223) is a one-off initialisation outside the step path")

    def generate_space_cache(self, styles: torch.Tensor, text_embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = self.space_generator(z=styles, c=text_embed, truncation_psi=self.truncation_psi)["image"]
        return get_activation(self.cfg.activation)(out)

    def interpolate_encodings(self, points, space_cache):
        return get_trilinear_feature(points=points, voxel=space_cache).reshape(*points.shape[:-1], -1)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        super().update_step(epoch, global_step, on_load_weights)
        self.truncation_psi = C(self.cfg.truncation_psi, epoch, global_step)
        self._fcfg = self._field_cfg()

    # ---- fused path: lookup + heads + bias + finite differences as one kernel (the shipped configuration) -------------------------------
    def _field_cfg(self) -> Optional[_lib.FieldCfg]:
        c = self.cfg
        if c.sdf_bias == "sphere" and isinstance(c.sdf_bias_params, float):
            bias, value = _lib.ASD_BIAS_SPHERE, float(c.sdf_bias_params)
        elif isinstance(c.sdf_bias, float):
            bias, value = _lib.ASD_BIAS_CONST, float(c.sdf_bias)
        else:
            return None
        m = c.mlp_network_config
        ok = (c.space_generator_config["img_channels"] == 32 and c.n_feature_dims == 3 and m.get("otype") == "VanillaMLP"
              and m.get("n_neurons") == 64 and m.get("n_hidden_layers") == 1 and m.get("activation") == "ReLU"
              and m.get("output_activation", "none") in (None, "none") and c.normal_type == "finite_difference"
              and self.finite_difference_normal_eps is not None and not self.unbounded and not c.isosurface_deformable_grid)
        if not ok:
            return None
        f = _lib.FieldCfg()
        for d in range(3):
            f.bbox_min[d], f.bbox_max[d] = -c.radius, c.radius
        f.radius, f.bias_mode, f.bias_value = c.radius, bias, value
        f.blob_scale, f.blob_std, f.activation = 0.0, 1.0, _lib.ASD_ACT_NONE
        f.fd_eps, f.n_hidden, f.n_feature_dims, f.field_mode = float(self.finite_difference_normal_eps), 64, 3, _lib.ASD_FIELD_SDF
        return f

    def _heads_weights(self):
        return (self.sdf_network.layers[0].weight, self.sdf_network.layers[2].weight,
                self.feature_network.layers[0].weight, self.feature_network.layers[2].weight)

    def _use_fused(self, points) -> bool:
        return getattr(self, "_fcfg", None) is not None and points.is_cuda and os.environ.get("ASD_VOXFIELD", "1") != "0"

    def _forward_points(self, points, space_cache, output_normal):
        if not self._use_fused(points):
            return super()._forward_points(points, space_cache, output_normal)
        vol = channels_last(space_cache)                                  # [B, D, H, W, 32] (the HIP generator's own layout: a view)
        w = self._heads_weights()
        need_grad = torch.is_grad_enabled() and (vol.requires_grad or w[0].requires_grad)
        outs = []
        for b in range(points.shape[0]):
            pts = points[b].reshape(-1, 3).contiguous().float()
            if need_grad:
                outs.append(_VoxFieldFn.apply(pts, *self._gate(space_cache, vol, b), *w, self._fcfg, bool(output_normal)))
            else:
                with torch.no_grad():
                    outs.append(ops.voxfield_fwd(vol[b], self._fcfg, *w, pts, bool(output_normal), save_enc=False)[:4])
        st = lambda i: torch.stack([o[i] for o in outs], 0)
        out = {"sdf": st(0)[..., None], "features": st(1)}
        if output_normal:
            normal, sdf_grad = st(2), st(3)
            out.update(normal=normal, shading_normal=normal, sdf_grad=sdf_grad)
        return out

    def forward_sdf(self, points, space_cache):
        if not self._use_fused(points):
            return super().forward_sdf(points, space_cache)
        vol, w = channels_last(space_cache), self._heads_weights()
        B = points.shape[0]
        with torch.set_grad_enabled(torch.is_grad_enabled()):
            pts = points.reshape(B, -1, 3)
            need_grad = torch.is_grad_enabled() and (vol.requires_grad or w[0].requires_grad)
            outs = []
            for b in range(B):
                p = pts[b].contiguous().float()
                if need_grad:
                    outs.append(_VoxFieldFn.apply(p, *self._gate(space_cache, vol, b), *w, self._fcfg, False)[0])
                else:
                    outs.append(ops.voxfield_fwd(vol[b], self._fcfg, *w, p, False, want_features=False, save_enc=False)[0])
        return torch.stack(outs, 0).reshape(*points.shape[:-1], 1)


class _TriFieldFn(torch.autograd.Function):
    """(sdf, features, normal, sdf_grad) of ONE batch entry from its points, its channel-last planes [3, H, W, 32] and the two 96 -> 64 -> 64 -> 1 | 3
    heads (include/asd_hip.h: asd_trifield_fwd / _bwd); nothing but the points and the sdf is kept for the backward pass"""

    @staticmethod
    def forward(ctx, points, planes_cl, b, slot, s1, s2, s3, f1, f2, f3, fcfg, want_normal):
        """planes_cl: the WHOLE channel-last cache [B, 3, H, W, 32] (contiguous), b: the entry these points sample"""
        w6 = (s1.t().contiguous(), s2.contiguous(), s3.contiguous(), f1.t().contiguous(), f2.contiguous(), f3.contiguous())
        sdf, feats, normal, fdg = ops.trifield_fwd(planes_cl[b], fcfg, w6, points, want_normal)
        if not want_normal:
            normal, fdg = sdf.new_zeros(0), sdf.new_zeros(0)
            ctx.mark_non_differentiable(normal, fdg)
        ctx.save_for_backward(points, planes_cl, sdf, *w6)
        ctx.fcfg, ctx.want_normal, ctx.slot, ctx.b = fcfg, want_normal, slot, b
        ctx.set_materialize_grads(False)
        return sdf, feats, normal, fdg

    @staticmethod
    def backward(ctx, d_sdf, d_feats, d_normal, d_fdg):
        points, planes_cl, sdf, *w6 = ctx.saved_tensors
        if d_sdf is None and d_feats is None and d_normal is None and d_fdg is None:
            return (None,) * 12
        d_pl, first = (ctx.slot or _GradSlot()).acquire(planes_cl)
        c = lambda t: None if t is None else t.contiguous()
        dws = ops.trifield_bwd(planes_cl[ctx.b], ctx.fcfg, w6, points, sdf, c(d_sdf), c(d_feats), c(d_normal) if ctx.want_normal else None,
                               c(d_fdg) if ctx.want_normal else None, d_pl[ctx.b])
        return (None, (d_pl if first else None), None, None, *dws, None, None)


@register("Triplane-transformer-sdf")
class TriplaneTransformerSDF(_SampledSdfGeometry):
    @dataclass
    class Config(BaseImplicitGeometry.Config):
        n_feature_dims: int = 3
        space_generator_config: dict = field(default_factory=lambda: {
            "inner_dim": 768, "condition_dim": 1024, "triplane_low_res": 32, "triplane_high_res": 64, "triplane_dim": 32,
            "num_layers": 12, "num_heads": 16, "flash_attention": False, "local_text": False})
        mlp_network_config: dict = field(default_factory=lambda: dict(_MLP2))
        backbone: str = "triplane_transformer"
        normal_type: Optional[str] = "finite_difference"
        finite_difference_normal_eps: Union[float, str] = 0.01
        sdf_bias: Union[float, str] = 0.0
        sdf_bias_params: Optional[Any] = None
        isosurface_remove_outliers: bool = False

    cfg: Config

    def configure(self) -> None:
        super().configure()
        if self.cfg.backbone != "triplane_transformer":
            raise ValueError(f"Unknown backbone {self.cfg.backbone}")
        self.space_generator = TriplaneTransformer(**self.cfg.space_generator_config)
        self._heads(self.cfg.space_generator_config["triplane_dim"] * 3)
        self.noise_dim = None

    def initialize_shape(self) -> None:
        pass

    def generate_space_cache(self, styles: Optional[torch.Tensor], text_embed: torch.Tensor) -> torch.Tensor:
        return self.space_generator(text_embed=text_embed)

    def interpolate_encodings(self, points, space_cache):
        return sample_from_planes(plane_features=space_cache, coordinates=points).view(*points.shape[:-1], -1)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        super().update_step(epoch, global_step, on_load_weights)
        self._fcfg = self._field_cfg()

    # ---- fused path: three lookups + both heads + bias + finite differences as one kernel each way (the shipped configuration) -----------
    def _field_cfg(self) -> Optional[_lib.FieldCfg]:
        c = self.cfg
        if c.sdf_bias == "sphere" and isinstance(c.sdf_bias_params, float):
            bias, value = _lib.ASD_BIAS_SPHERE, float(c.sdf_bias_params)
        elif isinstance(c.sdf_bias, float):
            bias, value = _lib.ASD_BIAS_CONST, float(c.sdf_bias)
        else:
            return None
        m = c.mlp_network_config
        ok = (c.space_generator_config["triplane_dim"] == 32 and c.n_feature_dims == 3 and m.get("otype") == "VanillaMLP"
              and m.get("n_neurons") == 64 and m.get("n_hidden_layers") == 2 and m.get("activation") == "ReLU"
              and m.get("output_activation", "none") in (None, "none") and c.normal_type == "finite_difference"
              and self.finite_difference_normal_eps is not None and not self.unbounded and not c.isosurface_deformable_grid)
        if not ok:
            return None
        f = _lib.FieldCfg()
        for d in range(3):
            f.bbox_min[d], f.bbox_max[d] = -c.radius, c.radius
        f.radius, f.bias_mode, f.bias_value = c.radius, bias, value
        f.blob_scale, f.blob_std, f.activation = 0.0, 1.0, _lib.ASD_ACT_NONE
        f.fd_eps, f.n_hidden, f.n_feature_dims, f.field_mode = float(self.finite_difference_normal_eps), 64, 3, _lib.ASD_FIELD_SDF
        return f

    def _heads_weights(self):
        s, f = self.sdf_network.layers, self.feature_network.layers
        return (s[0].weight, s[2].weight, s[4].weight, f[0].weight, f[2].weight, f[4].weight)

    def _use_fused(self, points) -> bool:
        return getattr(self, "_fcfg", None) is not None and points.is_cuda and os.environ.get("ASD_TRIFIELD", "1") != "0"

    def _forward_points(self, points, space_cache, output_normal):
        if not self._use_fused(points):
            return super()._forward_points(points, space_cache, output_normal)
        planes = planes_channels_last(space_cache)                        # [B, 3, H, W, 32]
        w = self._heads_weights()
        need_grad = torch.is_grad_enabled() and (planes.requires_grad or w[0].requires_grad)
        outs = []
        for b in range(points.shape[0]):
            pts = points[b].reshape(-1, 3).contiguous().float()
            if need_grad:
                outs.append(_TriFieldFn.apply(pts, *self._gate(space_cache, planes, b), *w, self._fcfg, bool(output_normal)))
            else:
                with torch.no_grad():
                    w6 = (w[0].t().contiguous(), w[1], w[2], w[3].t().contiguous(), w[4], w[5])
                    outs.append(ops.trifield_fwd(planes[b], self._fcfg, w6, pts, bool(output_normal)))
        st = lambda i: torch.stack([o[i] for o in outs], 0)
        out = {"sdf": st(0)[..., None], "features": st(1)}
        if output_normal:
            normal, sdf_grad = st(2), st(3)
            out.update(normal=normal, shading_normal=normal, sdf_grad=sdf_grad)
        return out

    def forward_sdf(self, points, space_cache):
        if not self._use_fused(points):
            return super().forward_sdf(points, space_cache)
        planes, w = planes_channels_last(space_cache), self._heads_weights()
        B = points.shape[0]
        pts = points.reshape(B, -1, 3)
        need_grad = torch.is_grad_enabled() and (planes.requires_grad or w[0].requires_grad)
        outs = []
        for b in range(B):
            p = pts[b].contiguous().float()
            if need_grad:
                outs.append(_TriFieldFn.apply(p, *self._gate(space_cache, planes, b), *w, self._fcfg, False)[0])
            else:
                w6 = (w[0].t().contiguous(), w[1], w[2], w[3].t().contiguous(), w[4], w[5])
                outs.append(ops.trifield_fwd(planes[b], self._fcfg, w6, p, False, want_features=False)[0])
        return torch.stack(outs, 0).reshape(*points.shape[:-1], 1)
