"""Hypernetwork-conditioned field and background of the multi-prompt configs on the HIP path:
  `Hyper-iNGP`  (custom/amortized/models/geometry/hyper_iNGP.py:113-424  Hypernet_Sdf, :18-111 LinearHyperNetwork)
  `multiprompt-neural-hashgrid-environment-map-background`
                (custom/amortized/models/background/multiprompt_neural_environment_hashgrid_map_background.py:19-116)

A text embedding [B, c_dim] is mapped by a small Linear-LayerNorm-SiLU-Linear network to per-prompt MLP weights
W1 [B, 32, 64], W2 [B, 64, 1|3]; the field of prompt b is  bmm(relu(bmm(enc, W1_b)), W2_b)  on the shared hash grid.
The fused SDF kernels (asd_field_fwd / asd_field_bwd with field_mode = ASD_FIELD_SDF) evaluate encode + both MLPs + sphere
bias + the finite-difference sdf_grad / normal in one pass per prompt, taking that prompt's weights as plain pointers;
their weight gradients flow back into the hypernetwork through autograd.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .background import BaseBackground
from .geometry import BaseImplicitGeometry, contract_to_unisphere
from .networks import get_activation, get_encoding
from .registry import register


class LinearHyperNetwork(nn.Module):
    def __init__(self, n_input_dims: int, config: dict):
        super().__init__()
        self.n_input_dims = n_input_dims
        self.config = config
        self.c_dim = config["c_dim"]
        out_dims = dict(config.get("out_dims", {"sdf_weights": [64, 1], "feature_weights": [64, 3]}))
        self.out_dims = {k: [n_input_dims] + (list(v) if isinstance(v, (list, tuple)) else [v]) for k, v in out_dims.items()}
        self.spectral_norm = config.get("spectral_norm", False)
        self.n_output_dims = sum(a * b for ch in self.out_dims.values() for a, b in zip(ch[:-1], ch[1:]))
        self.n_neurons, self.n_hidden_layers = config["n_neurons"], config["n_hidden_layers"]
        layers: List[nn.Module] = [self.make_linear(self.c_dim, self.n_neurons, has_bias=False), nn.LayerNorm(self.n_neurons), nn.SiLU(inplace=True)]
        for _ in range(self.n_hidden_layers - 1):
            layers += [self.make_linear(self.n_neurons, self.n_neurons, has_bias=True), nn.LayerNorm(self.n_neurons), nn.SiLU(inplace=True)]
        layers += [self.make_linear(self.n_neurons, self.n_output_dims, has_bias=True)]
        self.layers = nn.Sequential(*layers)
        self.output_activation = get_activation(config.get("output_activation", None))

    def make_linear(self, dim_in, dim_out, has_bias):
        layer = nn.Linear(dim_in, dim_out, bias=has_bias)
        if self.spectral_norm:
            layer = nn.utils.spectral_norm(layer)
        if has_bias:
            nn.init.zeros_(layer.bias)
        nn.init.xavier_normal_(layer.weight, gain=1.0)
        return layer

    def forward(self, x: torch.Tensor) -> Dict[str, List[torch.Tensor]]:
        with torch.autocast(device_type=x.device.type, enabled=False):
            out = self.layers(x)
            if self.output_activation is not None:
                out = self.output_activation(out)
        res, start = {}, 0
        for item, ch in self.out_dims.items():
            ws = []
            for cin, cout in zip(ch[:-1], ch[1:]):
                ws.append(out[:, start:start + cin * cout].reshape(*x.shape[:-1], cin, cout))
                start += cin * cout
            res[item] = ws
        return res


def hypernet_forward(enc: torch.Tensor, params, activation=torch.relu, output_activation=None) -> torch.Tensor:
    """hyper_iNGP.py:236-261: chained bmm without bias."""
    if torch.is_tensor(params):
        params = [params]
    for idx, p in enumerate(params):
        assert enc.shape[0] == p.shape[0] and enc.shape[-1] == p.shape[1]
        enc = torch.bmm(enc, p)
        if activation is not None and idx < len(params) - 1:
            enc = activation(enc)
        elif output_activation is not None and idx == len(params) - 1:
            enc = output_activation(enc)
    return enc


class _SdfFieldFn(torch.autograd.Function):
    """(sdf, features, normal, sdf_grad) of ONE prompt from its points and hypernetwork weights (fused HIP kernels)."""

    @staticmethod
    def forward(ctx, points, grid, w1d, w2d, w1f, w2f, meta, fcfg, want_normal):
        if want_normal:
            sdf, feats, normal, fdg, enc = ops.field_fwd(meta, fcfg, grid, w1d, w2d, w1f, w2f, points, True, want_fd_grad=True)
        else:
            sdf, feats, normal, enc = ops.field_fwd(meta, fcfg, grid, w1d, w2d, w1f, w2f, points, False)
            normal, fdg = sdf.new_zeros(0), sdf.new_zeros(0)
            ctx.mark_non_differentiable(normal, fdg)
        ctx.save_for_backward(points, grid, w1d, w2d, w1f, w2f, enc, sdf)
        ctx.meta, ctx.fcfg, ctx.want_normal = meta, fcfg, want_normal
        ctx.set_materialize_grads(False)
        return sdf, feats, normal, fdg

    @staticmethod
    def backward(ctx, d_sdf, d_feats, d_normal, d_fdg):
        points, grid, w1d, w2d, w1f, w2f, enc, sdf = ctx.saved_tensors
        d_grid = torch.zeros_like(grid)
        if d_sdf is None and d_feats is None and d_normal is None and d_fdg is None:
            return (None, d_grid, *(torch.zeros_like(w) for w in (w1d, w2d, w1f, w2f)), None, None, None)
        c = lambda t: None if t is None else t.contiguous()
        dw = ops.field_bwd(ctx.meta, ctx.fcfg, grid, w1d, w2d, w1f, w2f, points, enc, sdf, c(d_sdf), c(d_feats),
                           c(d_normal) if ctx.want_normal else None, d_grid, d_fd_grad=c(d_fdg) if ctx.want_normal else None)
        return None, d_grid, dw[0], dw[1], dw[2], dw[3], None, None, None


@register("Hyper-iNGP")
class Hypernet_Sdf(BaseImplicitGeometry):
    @dataclass
    class Config(BaseImplicitGeometry.Config):
        n_input_dims: int = 3
        n_feature_dims: int = 3
        hypernet_config: dict = field(default_factory=lambda: {
            "c_dim": 768, "out_dims": {"sdf_weights": [64, 1], "feature_weights": [64, 3]}, "spectral_norm": False,
            "n_neurons": 64, "n_hidden_layers": 1, "output_activation": None})
        pos_encoding_config: dict = field(default_factory=lambda: {
            "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
            "per_level_scale": 1.447269237440378})
        backbone: str = "linear_hypernetwork"
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
        self.encoding = get_encoding(self.cfg.n_input_dims, self.cfg.pos_encoding_config)
        if self.cfg.backbone != "linear_hypernetwork":
            raise NotImplementedError
        self.hypernet = LinearHyperNetwork(self.encoding.n_output_dims, self.cfg.hypernet_config)
        if self.cfg.normal_type == "pred":
            raise NotImplementedError("normal_type == pred is not implemented yet.")
        self.finite_difference_normal_eps: Optional[float] = None
        self._meta = self.encoding.encoding.encoding.meta
        self._fcfg: Optional[_lib.FieldCfg] = None

    def initialize_shape(self) -> None:
        if self.cfg.shape_init is None and not self.cfg.force_shape_init:
            return
        if self.cfg.weights is not None and not self.cfg.force_shape_init:
            return
        raise NotImplementedError

    # ---- fused-kernel configuration --------------------------------------------------------------------------------------
    def _field_cfg(self) -> Optional[_lib.FieldCfg]:
        c = self.cfg
        od = self.hypernet.out_dims
        if c.sdf_bias == "sphere" and isinstance(c.sdf_bias_params, float):
            bias, value = _lib.ASD_BIAS_SPHERE, float(c.sdf_bias_params)
        elif isinstance(c.sdf_bias, float):
            bias, value = _lib.ASD_BIAS_CONST, float(c.sdf_bias)
        else:
            return None
        ok = (self._meta.n_levels == 16 and self.encoding.n_output_dims == 32 and not self.encoding.include_xyz
              and od.get("sdf_weights") == [32, 64, 1] and od.get("feature_weights") == [32, 64, 3] and c.n_feature_dims == 3
              and c.normal_type == "finite_difference" and self.finite_difference_normal_eps is not None and c.n_input_dims == 3)
        if not ok:
            return None
        f = _lib.FieldCfg()
        for d in range(3):
            f.bbox_min[d], f.bbox_max[d] = -c.radius, c.radius
        f.radius, f.bias_mode, f.bias_value = c.radius, bias, value
        f.blob_scale, f.blob_std, f.activation = 0.0, 1.0, _lib.ASD_ACT_NONE
        f.fd_eps, f.n_hidden, f.n_feature_dims, f.field_mode = float(self.finite_difference_normal_eps), 64, 3, _lib.ASD_FIELD_SDF
        return f

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        if self.cfg.normal_type == "finite_difference":
            if isinstance(self.cfg.finite_difference_normal_eps, float):
                self.finite_difference_normal_eps = self.cfg.finite_difference_normal_eps
        else:
            raise NotImplementedError(f"normal_type == {self.cfg.normal_type} is not implemented yet.")
        self._fcfg = self._field_cfg()

    # ---- reference surface ---------------------------------------------------------------------------------------------
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

    def generate_space_cache(self, styles, text_embed: Optional[torch.Tensor] = None) -> Any:
        return self.hypernet(text_embed)  # the noise `styles` is not used by the hypernetwork

    hypernet_forward = staticmethod(hypernet_forward)

    def _fused(self, points: torch.Tensor, space_cache: Dict, output_normal: bool):
        """points [B, Np, 3] -> per-prompt fused kernels; returns (sdf [B*Np,1], features, normal, sdf_grad)."""
        grid = self.encoding.encoding.encoding.params
        need_grad = torch.is_grad_enabled() and (grid.requires_grad or space_cache["sdf_weights"][0].requires_grad)
        outs = []
        for b in range(points.shape[0]):
            w1d, w2d = (w[b].t().contiguous() for w in space_cache["sdf_weights"])           # [in, out] -> [out, in]
            w1f, w2f = (w[b].t().contiguous() for w in space_cache["feature_weights"])
            pts = points[b].reshape(-1, 3).contiguous().float()
            if need_grad:
                outs.append(_SdfFieldFn.apply(pts, grid, w1d, w2d, w1f, w2f, self._meta, self._fcfg, bool(output_normal)))
            else:
                with torch.no_grad():
                    if output_normal:
                        s, f, n, g, _ = ops.field_fwd(self._meta, self._fcfg, grid, w1d, w2d, w1f, w2f, pts, True, want_fd_grad=True)
                    else:
                        (s, f, n, _), g = ops.field_fwd(self._meta, self._fcfg, grid, w1d, w2d, w1f, w2f, pts, False), None
                outs.append((s, f, n, g))
        cat = lambda i: torch.cat([o[i] for o in outs], 0) if len(outs) > 1 else outs[0][i]
        return cat(0).view(-1, 1), cat(1), (cat(2), cat(3)) if output_normal else (None, None)

    def _require_fused(self, points: torch.Tensor) -> None:
        if self._fcfg is None:
            raise NotImplementedError(
                "Hyper-iNGP runs as the fused SDF field kernels only: 16-level x 2-feature hash grid without xyz passthrough, "
                "hypernetwork MLPs [32, 64, 1] / [32, 64, 3], finite-difference normals, sphere or constant SDF bias "
                "(the configuration of asd_sd_hyper_iNGP_50k.yaml); this configuration is outside that")
        if not points.is_cuda:
            raise _lib.AsdError("the HIP path needs device tensors (there is no CPU fallback)")

    def forward(self, points: torch.Tensor, space_cache: Dict, output_normal: bool = False) -> Dict[str, torch.Tensor]:
        """points [B, Np, 3] + per-prompt MLP weights from the hypernetwork -> sdf [B*Np, 1], features [B*Np, 3] and, with
        output_normal, the finite-difference sdf_grad and its normalisation (hyper_iNGP.py:229-330), one fused kernel per prompt."""
        if output_normal and self.cfg.normal_type == "analytic":
            raise NotImplementedError("analytic normal is not implemented yet.")
        self._require_fused(points)
        sdf, feats, (normal, sdf_grad) = self._fused(points, space_cache, output_normal)
        out = {"sdf": sdf, "features": feats}
        if output_normal:
            out.update(normal=normal, shading_normal=normal, sdf_grad=sdf_grad)
        return out

    def forward_sdf(self, points: torch.Tensor, space_cache: Dict) -> torch.Tensor:
        """sdf only, [..., 1] in the shape of points[..., :1] (hyper_iNGP.py:332-349); used for evaluation / iso-surfaces"""
        self._require_fused(points)
        with torch.no_grad():
            sdf, _, _ = self._fused(points.reshape(points.shape[0], -1, 3), space_cache, False)
        return sdf.view(*points.shape[:-1], 1)

    def forward_field(self, points, space_cache):
        return self.forward_sdf(points, space_cache), None

    def forward_level(self, field, threshold):
        return field - threshold

    def export(self, points, space_cache, **kwargs) -> Dict[str, Any]:
        if self.cfg.n_feature_dims == 0:
            return {}
        return {"features": self.forward(points, space_cache)["features"].view(*points.shape[:-1], self.cfg.n_feature_dims)}


@register("multiprompt-neural-hashgrid-environment-map-background")
class MultipromptNeuralHashgridEnvironmentMapBackground(BaseBackground):
    @dataclass
    class Config(BaseBackground.Config):
        n_output_dims: int = 3
        color_activation: str = "sigmoid"
        pos_encoding_config: dict = field(default_factory=lambda: {
            "otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 4,
            "per_level_scale": 1.8114473285278132})
        hypernet_config: dict = field(default_factory=lambda: {
            "c_dim": 1024, "out_dims": {"bg_weights": [64, 3]}, "spectral_norm": False, "n_neurons": 64, "n_hidden_layers": 1,
            "output_activation": None})
        random_aug: bool = False
        random_aug_prob: float = 0.5
        eval_color: Optional[Tuple[float, float, float]] = None

    cfg: Config

    def configure(self) -> None:
        self.encoding = get_encoding(3, self.cfg.pos_encoding_config)
        self.hypernet = LinearHyperNetwork(self.encoding.n_output_dims, self.cfg.hypernet_config)
        self.enabling_hypernet = True
        self.rand_fn = lambda b, c, like: (lambda r: r.pin_memory().to(like.device, non_blocking=True) if like.is_cuda else r)(torch.rand(b, 1, 1, c)).to(like)   # injectable for tests; pinned + asynchronous: no host stall

    hypernet_forward = staticmethod(hypernet_forward)

    def forward(self, dirs: torch.Tensor, text_embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B, H, W, 3] unit directions -> [B, H, W, n_output_dims] colours (multiprompt_neural_environment_hashgrid_map_background.py:82-116).
        Three cases, decided BEFORE any field evaluation:
          * eval with a fixed colour: a constant image;
          * training with random_aug, with probability random_aug_prob: one uniform random colour per batch element.  The reference
            evaluates the whole field and multiplies it by zero, whose only lasting effect is that every parameter receives an all-zero
            gradient (so AdamW still decays its moments and applies weight decay this step); the same gradient is produced here by a
            zero-weighted sum over the parameters — no hash-grid gather, no hypernetwork pass, same draws from `random` / torch in the
            same order;
          * otherwise: hash-grid encoding of the directions through the per-prompt MLP the hypernetwork emits, then the colour activation."""
        lead, n_out = dirs.shape[:-1], self.cfg.n_output_dims
        if not self.training and self.cfg.eval_color is not None:
            return torch.as_tensor(self.cfg.eval_color).to(dirs).expand(*lead, n_out).contiguous()
        if self.training and self.cfg.random_aug and random.random() < self.cfg.random_aug_prob:
            zero_grad_anchor = sum(p.sum() for p in self.parameters() if p.requires_grad) * 0.0
            return self.rand_fn(dirs.shape[0], n_out, dirs).expand(*lead, n_out) + zero_grad_anchor
        mlp = self.hypernet(text_embed)["bg_weights"]
        feats = self.encoding(((dirs + 1.0) * 0.5).reshape(-1, 3)).view(dirs.shape[0], -1, self.encoding.n_output_dims)
        return get_activation(self.cfg.color_activation)(hypernet_forward(feats, mlp).view(*lead, n_out))
