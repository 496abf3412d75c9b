"""`implicit-volume` geometry (threestudio/models/geometry/implicit_volume.py:19-207) on the HIP path.

Module / parameter layout is the reference's (encoding / density_network / feature_network as nn.Modules
with fp32 nn.Parameters — the optimizer addresses them by dotted name, systems/utils.py:19-39); the
arithmetic of forward()/forward_density() runs in the fused HIP field kernels when the configuration is
the one those kernels are built for, otherwise in the composed path (HIP hash grid + torch VanillaMLP).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Union

import torch
import torch.nn.functional as F

from . import _lib, ops
from .base import BaseModule
from .networks import VanillaMLP, get_activation, get_encoding, get_mlp
from .registry import register, warn


def scale_tensor(dat, inp_scale, tgt_scale):
    """threestudio/utils/ops.py:27-38"""
    if inp_scale is None:
        inp_scale = (0, 1)
    if tgt_scale is None:
        tgt_scale = (0, 1)
    dat = (dat - inp_scale[0]) / (inp_scale[1] - inp_scale[0])
    return dat * (tgt_scale[1] - tgt_scale[0]) + tgt_scale[0]


def contract_to_unisphere(x, bbox, unbounded: bool = False):
    """threestudio/models/geometry/base.py:20-32"""
    if unbounded:
        x = scale_tensor(x, bbox, (0, 1))
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        mask = mag.squeeze(-1) > 1
        x[mask] = (2 - 1 / mag[mask]) * (x[mask] / mag[mask])
        return x / 4 + 0.5
    return scale_tensor(x, bbox, (0, 1))


class BaseImplicitGeometry(BaseModule):
    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0
        isosurface: bool = True
        isosurface_method: str = "mt"
        isosurface_resolution: int = 128
        isosurface_threshold: Union[float, str] = 0.0
        isosurface_chunk: int = 0
        isosurface_coarse_to_fine: bool = True
        isosurface_deformable_grid: bool = False
        isosurface_remove_outliers: bool = True
        isosurface_outlier_n_faces_threshold: Union[int, float] = 0.01

    cfg: Config

    def configure(self) -> None:
        r = self.cfg.radius
        self.register_buffer("bbox", torch.as_tensor([[-r, -r, -r], [r, r, r]], dtype=torch.float32))
        self.unbounded = False

    def isosurface(self):
        raise NotImplementedError("mesh extraction is post-training asset export (SURVEY.md §2.1 #17): out of scope")


class _FieldFn(torch.autograd.Function):
    """sigma, features, normal = field(points); backward scatters into the table and reduces MLP grads."""

    @staticmethod
    def forward(ctx, points, grid, w1d, w2d, w1f, w2f, geom, want_normal, n_dev=None):
        sigma, feats, normal, enc = ops.field_fwd(geom._meta, geom._fcfg, grid, w1d, w2d, w1f, w2f, points, want_normal, n_dev=n_dev)
        ctx.save_for_backward(points, grid, w1d, w2d, w1f, w2f, enc, sigma)
        ctx.geom, ctx.n_dev = geom, n_dev
        ctx.set_materialize_grads(False)
        if normal is None:
            normal = sigma.new_zeros(0)
        ctx.mark_non_differentiable(*([normal] if not want_normal else []))
        return sigma, feats, normal

    @staticmethod
    def backward(ctx, d_sigma, d_feats, d_normal):
        points, grid, w1d, w2d, w1f, w2f, enc, sigma = ctx.saved_tensors
        g = ctx.geom
        d_grid = torch.zeros_like(grid)
        if d_sigma is None and d_feats is None and d_normal is None:
            return None, d_grid, torch.zeros_like(w1d), torch.zeros_like(w2d), torch.zeros_like(w1f), torch.zeros_like(w2f), None, None, None
        dw = ops.field_bwd(g._meta, g._fcfg, grid, w1d, w2d, w1f, w2f, points, enc, sigma,
                           None if d_sigma is None else d_sigma.contiguous(),
                           None if d_feats is None else d_feats.contiguous(),
                           None if d_normal is None else d_normal.contiguous(), d_grid, n_dev=ctx.n_dev)
        return None, d_grid, dw[0], dw[1], dw[2], dw[3], None, None, None


@register("implicit-volume")
class ImplicitVolume(BaseImplicitGeometry):
    @dataclass
    class Config(BaseImplicitGeometry.Config):
        n_input_dims: int = 3
        n_feature_dims: int = 3
        density_activation: Optional[str] = "softplus"
        density_bias: Union[float, str] = "blob_magic3d"
        density_blob_scale: float = 10.0
        density_blob_std: float = 0.5
        pos_encoding_config: dict = field(
            default_factory=lambda: {
                "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                "base_resolution": 16, "per_level_scale": 1.447269237440378,
            }
        )
        mlp_network_config: dict = field(
            default_factory=lambda: {
                "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
                "n_hidden_layers": 1,
            }
        )
        normal_type: Optional[str] = "finite_difference"
        finite_difference_normal_eps: float = 0.01
        isosurface_threshold: Union[float, str] = 25.0
        anneal_density_blob_std_config: Optional[dict] = None

    cfg: Config

    def configure(self) -> None:
        super().configure()
        self.encoding = get_encoding(self.cfg.n_input_dims, self.cfg.pos_encoding_config)
        self.density_network = get_mlp(self.encoding.n_output_dims, 1, self.cfg.mlp_network_config)
        if self.cfg.n_feature_dims > 0:
            self.feature_network = get_mlp(self.encoding.n_output_dims, self.cfg.n_feature_dims, self.cfg.mlp_network_config)
        if self.cfg.normal_type == "pred":
            self.normal_network = get_mlp(self.encoding.n_output_dims, 3, self.cfg.mlp_network_config)
        self._meta = self.encoding.encoding.encoding.meta
        self._fcfg = self._make_field_cfg()

    # ---- fused-kernel eligibility ---------------------------------------------------------------
    def _make_field_cfg(self) -> Optional[_lib.FieldCfg]:
        c = self.cfg
        act = {"softplus": _lib.ASD_ACT_SOFTPLUS, "exp": _lib.ASD_ACT_EXP, "trunc_exp": _lib.ASD_ACT_TRUNC_EXP,
               None: _lib.ASD_ACT_NONE, "none": _lib.ASD_ACT_NONE}.get(c.density_activation, -1)
        if isinstance(c.density_bias, str):
            bias = {"blob_magic3d": _lib.ASD_BIAS_BLOB_MAGIC3D, "blob_dreamfusion": _lib.ASD_BIAS_BLOB_DREAMFUSION}.get(c.density_bias, -1)
            bias_value = 0.0
        else:
            bias, bias_value = _lib.ASD_BIAS_CONST, float(c.density_bias)
        mlp = c.mlp_network_config
        ok = (
            act >= 0 and bias >= 0 and self._meta.n_levels == 16 and self.encoding.n_output_dims == 32
            and not self.encoding.include_xyz and isinstance(self.density_network, VanillaMLP)
            and mlp.get("n_neurons") == 64 and mlp.get("n_hidden_layers") == 1
            and mlp.get("output_activation", "none") in (None, "none") and c.n_feature_dims in (0, 3)
            and c.normal_type in (None, "finite_difference") and c.n_input_dims == 3
        )
        if not ok:
            return None
        f = _lib.FieldCfg()
        for d in range(3):
            f.bbox_min[d], f.bbox_max[d] = -c.radius, c.radius
        f.radius, f.bias_mode, f.bias_value = c.radius, bias, bias_value
        f.blob_scale, f.blob_std, f.activation = c.density_blob_scale, c.density_blob_std, act
        f.fd_eps, f.n_hidden, f.n_feature_dims = c.finite_difference_normal_eps, 64, c.n_feature_dims
        return f

    @property
    def fused(self) -> bool:
        return self._fcfg is not None

    def _weights(self):
        w1d, w2d = self.density_network.layers[0].weight, self.density_network.layers[2].weight
        if self.cfg.n_feature_dims > 0:
            return w1d, w2d, self.feature_network.layers[0].weight, self.feature_network.layers[2].weight
        return w1d, w2d, w1d, w2d  # dummies, never dereferenced when n_feature_dims == 0

    # ---- reference surface ----------------------------------------------------------------------
    def get_activated_density(self, points, density):
        c = self.cfg
        if c.density_bias == "blob_dreamfusion":
            bias = c.density_blob_scale * torch.exp(-0.5 * (points**2).sum(dim=-1) / c.density_blob_std**2)[..., None]
        elif c.density_bias == "blob_magic3d":
            bias = c.density_blob_scale * (1 - torch.sqrt((points**2).sum(dim=-1)) / c.density_blob_std)[..., None]
        elif isinstance(c.density_bias, float):
            bias = c.density_bias
        else:
            raise ValueError(f"Unknown density bias {c.density_bias}")
        raw = density + bias
        return raw, get_activation(c.density_activation)(raw)

    def forward(self, points: torch.Tensor, output_normal: bool = False, n_dev: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """n_dev (extension of the reference signature, fused path only): int32 device scalar — only the first n_dev points are valid,
        the rows behind them are neither read nor written (the renderer's capacity-sized sample buffers, no host read of the count)"""
        if self.fused and points.is_cuda:
            return self._forward_fused(points, output_normal, n_dev)
        if n_dev is not None:
            raise ValueError("n_dev needs the fused field kernels")
        return self._forward_composed(points, output_normal)

    def _forward_fused(self, points, output_normal, n_dev=None):
        if output_normal and self.cfg.normal_type is None:
            raise AttributeError(f"Unknown normal type {self.cfg.normal_type}")
        shape = points.shape[:-1]
        flat = points.reshape(-1, 3).contiguous().float()
        w1d, w2d, w1f, w2f = self._weights()
        grid = self.encoding.encoding.encoding.params
        if torch.is_grad_enabled() and grid.requires_grad:
            sigma, feats, normal = _FieldFn.apply(flat, grid, w1d, w2d, w1f, w2f, self, bool(output_normal), n_dev)
        else:
            sigma, feats, normal, _ = ops.field_fwd(self._meta, self._fcfg, grid.detach(), w1d.detach(), w2d.detach(),
                                                    w1f.detach(), w2f.detach(), flat, bool(output_normal), n_dev=n_dev)
        out = {"density": sigma.view(*shape, 1)}
        if self.cfg.n_feature_dims > 0:
            out["features"] = feats.view(*shape, self.cfg.n_feature_dims)
        if output_normal:
            n = normal.view(*shape, 3)
            out.update({"normal": n, "shading_normal": n})
        return out

    def _forward_composed(self, points, output_normal):
        c = self.cfg
        points_unscaled = points
        pts = contract_to_unisphere(points, self.bbox, self.unbounded)
        enc = self.encoding(pts.view(-1, c.n_input_dims))
        density = self.density_network(enc).view(*pts.shape[:-1], 1)
        _, density = self.get_activated_density(points_unscaled, density)
        out = {"density": density}
        if c.n_feature_dims > 0:
            out["features"] = self.feature_network(enc).view(*pts.shape[:-1], c.n_feature_dims)
        if output_normal:
            if c.normal_type in ("finite_difference", "finite_difference_laplacian"):
                eps = c.finite_difference_normal_eps
                if c.normal_type == "finite_difference_laplacian":
                    offs = torch.as_tensor([[eps, 0, 0], [-eps, 0, 0], [0, eps, 0], [0, -eps, 0], [0, 0, eps], [0, 0, -eps]]).to(points_unscaled)
                    po = (points_unscaled[..., None, :] + offs).clamp(-c.radius, c.radius)
                    do = self.forward_density(po)
                    normal = -0.5 * (do[..., 0::2, 0] - do[..., 1::2, 0]) / eps
                else:
                    offs = torch.as_tensor([[eps, 0.0, 0.0], [0.0, eps, 0.0], [0.0, 0.0, eps]]).to(points_unscaled)
                    po = (points_unscaled[..., None, :] + offs).clamp(-c.radius, c.radius)
                    do = self.forward_density(po)
                    normal = -(do[..., 0::1, 0] - density) / eps
                normal = F.normalize(normal, dim=-1)
            elif c.normal_type == "pred":
                normal = F.normalize(self.normal_network(enc).view(*pts.shape[:-1], 3), dim=-1)
            else:
                raise AttributeError(f"Unknown normal type {c.normal_type}")
            out.update({"normal": normal, "shading_normal": normal})
        return out

    def forward_density(self, points: torch.Tensor, n_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """n_dev (extension of the reference signature): int32 device scalar, the number of leading points that are valid — the
        renderer passes capacity-sized candidate buffers and the marcher's device-side count instead of reading the count back."""
        if self.fused and points.is_cuda and not (torch.is_grad_enabled() and self.encoding.encoding.encoding.params.requires_grad):
            w1d, w2d, _, _ = self._weights()
            grid = self.encoding.encoding.encoding.params
            flat = points.reshape(-1, 3).contiguous().float()
            s = ops.field_density(self._meta, self._fcfg, grid.detach(), w1d.detach(), w2d.detach(), flat, n_dev=n_dev)
            return s.view(*points.shape[:-1], 1)
        if self.fused and points.is_cuda:
            return self._forward_fused(points, False)["density"]
        pts = contract_to_unisphere(points, self.bbox, self.unbounded)
        density = self.density_network(self.encoding(pts.reshape(-1, self.cfg.n_input_dims))).reshape(*pts.shape[:-1], 1)
        return self.get_activated_density(points, density)[1]

    def forward_field(self, points):
        if self.cfg.isosurface_deformable_grid:
            warn(f"{self.__class__.__name__} does not support isosurface_deformable_grid. Ignoring.")
        return self.forward_density(points), None

    def forward_level(self, field, threshold):
        return -(field - threshold)

    def export(self, points, **kwargs) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        if self.cfg.n_feature_dims == 0:
            return out
        out["features"] = self.forward(points)["features"]
        return out
