"""Encodings and MLPs with the reference's module layout (threestudio/models/networks.py).

`Encoding` is the stand-in for `tcnn.Encoding(n_in, {"otype": "HashGrid", ...}, dtype=float32)`
(networks.py:55-64): an nn.Module with a flat fp32 `.params`, `.n_output_dims`, and a forward that runs the
HIP hash-grid kernels through the C ABI.  The nesting CompositeEncoding -> TCNNEncoding -> Encoding keeps the
reference's state-dict key `geometry.encoding.encoding.encoding.params`.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops
from .base import Updateable
from .config import config_to_primitive


def get_activation(name):
    """threestudio/utils/ops.py:78-113 (the subset reachable from the shipped configs + generic F.*)."""
    if name is None:
        return lambda x: x
    name = name.lower()
    if name == "none":
        return lambda x: x
    if name == "exp":
        return torch.exp
    if name == "shifted_exp":
        return lambda x: torch.exp(x - 1.0)
    if name == "sigmoid":
        return torch.sigmoid
    if name == "tanh":
        return torch.tanh
    if name == "shifted_softplus":
        return lambda x: F.softplus(x - 1.0)
    if name == "scale_-11_01":
        return lambda x: x * 0.5 + 0.5
    if name == "sigmoid-mipnerf":
        return lambda x: torch.sigmoid(x) * (1 + 2 * 0.001) - 0.001
    if name == "lin2srgb":
        return lambda x: torch.where(
            x > 0.0031308, torch.pow(torch.clamp(x, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * x
        ).clamp(0.0, 1.0)
    try:
        return getattr(F, name)
    except AttributeError:
        raise ValueError(f"Unknown activation function: {name}")


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, meta):
        out = ops.hashgrid_fwd(meta, params, x)
        ctx.save_for_backward(x)
        ctx.meta = meta
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return None, ops.hashgrid_bwd(ctx.meta, x, dout.contiguous()), None


class Encoding(nn.Module):
    """tcnn.Encoding replacement for otype HashGrid / Grid(Hash) with Linear interpolation, F=2."""

    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=torch.float32, seed: int = 1337):
        super().__init__()
        cfg = config_to_primitive(encoding_config)
        otype = cfg.get("otype", "HashGrid")
        if otype not in ("HashGrid", "Grid") or (otype == "Grid" and cfg.get("type", "Hash") != "Hash"):
            raise NotImplementedError(f"encoding otype {otype!r} is not implemented by the HIP path")
        if n_input_dims != 3:
            raise NotImplementedError("the HIP hash grid is 3-D")
        if dtype != torch.float32:
            raise NotImplementedError("the reference instantiates tcnn.Encoding with dtype=float32 (networks.py:56)")
        if cfg.get("interpolation", "Linear") != "Linear":
            raise NotImplementedError("only Linear interpolation")
        self.n_input_dims = n_input_dims
        self.encoding_config = cfg
        self.meta = _lib.make_grid_meta(
            int(cfg.get("n_levels", 16)), int(cfg.get("n_features_per_level", 2)), int(cfg.get("log2_hashmap_size", 19)),
            int(cfg.get("base_resolution", 16)), float(cfg.get("per_level_scale", 2.0)),
        )
        self.n_output_dims = int(self.meta.n_levels * self.meta.n_features)
        # tcnn initialises grid parameters U(-1e-4, 1e-4)
        g = torch.Generator().manual_seed(seed)
        self.params = nn.Parameter((torch.rand(self.meta.n_params, generator=g) * 2 - 1) * 1e-4)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _HashGridFn.apply(x.contiguous().float(), self.params, self.meta)


class SphericalHarmonics(nn.Module):
    """tcnn `SphericalHarmonics` encoding (the default `dir_encoding_config` of neural-environment-map-background,
    neural_environment_map_background.py:21-23, reached by the 3DConv-net / triplane configs): the input in [0,1]^3 is mapped to
    [-1,1]^3 and expanded in the real SH basis of Instant-NGP up to `degree` (degree^2 outputs, no parameters).  Un-vendored
    third-party arithmetic restated from its published form; a few thousand ray directions per step: plain tensor ops."""

    def __init__(self, n_input_dims: int, encoding_config: dict, dtype=torch.float32):
        super().__init__()
        self.degree = int(encoding_config.get("degree", 4))
        if n_input_dims != 3 or not 1 <= self.degree <= 4:
            raise NotImplementedError("SphericalHarmonics: 3-D input, degree 1..4")
        self.n_input_dims, self.n_output_dims = 3, self.degree ** 2
        self.register_parameter("params", nn.Parameter(torch.zeros(0)))   # tcnn modules expose an (empty) params tensor

    def forward(self, v: torch.Tensor) -> torch.Tensor:
        x, y, z = (v.float() * 2.0 - 1.0).unbind(-1)
        xy, yz, xz, x2, y2, z2 = x * y, y * z, x * z, x * x, y * y, z * z
        out = [torch.full_like(x, 0.28209479177387814)]
        if self.degree > 1:
            out += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
        if self.degree > 2:
            out += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
                    -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2]
        if self.degree > 3:
            out += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z, 0.45704579946446572 * y * (1.0 - 5.0 * z2),
                    0.3731763325901154 * z * (5.0 * z2 - 3.0), 0.45704579946446572 * x * (1.0 - 5.0 * z2),
                    1.4453057213202769 * z * (x2 - y2), 0.59004358992664352 * x * (-x2 + 3.0 * y2)]
        return torch.stack(out, dim=-1)


class TCNNEncoding(nn.Module):
    def __init__(self, in_channels, config, dtype=torch.float32) -> None:
        super().__init__()
        self.n_input_dims = in_channels
        self.encoding = (SphericalHarmonics if config.get("otype") == "SphericalHarmonics" else Encoding)(in_channels, config, dtype=dtype)
        self.n_output_dims = self.encoding.n_output_dims

    def forward(self, x):
        return self.encoding(x)


class CompositeEncoding(nn.Module, Updateable):
    def __init__(self, encoding, include_xyz=False, xyz_scale=2.0, xyz_offset=-1.0):
        super().__init__()
        self.encoding = encoding
        self.include_xyz, self.xyz_scale, self.xyz_offset = include_xyz, xyz_scale, xyz_offset
        self.n_output_dims = int(self.include_xyz) * self.encoding.n_input_dims + self.encoding.n_output_dims

    def forward(self, x, *args):
        if not self.include_xyz:
            return self.encoding(x, *args)
        return torch.cat([x * self.xyz_scale + self.xyz_offset, self.encoding(x, *args)], dim=-1)


def get_encoding(n_input_dims: int, config) -> nn.Module:
    otype = config.get("otype", "HashGrid")
    if otype in ("ProgressiveBandFrequency", "ProgressiveBandHashGrid", "HashGridSpatialTime"):
        raise NotImplementedError(f"{otype} is not used by any shipped ScaleDreamer config (out of scope)")
    enc = TCNNEncoding(n_input_dims, config_to_primitive(config))
    return CompositeEncoding(enc, include_xyz=config.get("include_xyz", False), xyz_scale=2.0, xyz_offset=-1.0)


class VanillaMLP(nn.Module):
    """networks.py:214-251: Linear(no bias) / ReLU stack, autocast disabled, optional output activation."""

    def __init__(self, dim_in: int, dim_out: int, config: dict):
        super().__init__()
        self.n_neurons, self.n_hidden_layers = config["n_neurons"], config["n_hidden_layers"]
        layers = [nn.Linear(dim_in, self.n_neurons, bias=False), nn.ReLU(inplace=True)]
        for _ in range(self.n_hidden_layers - 1):
            layers += [nn.Linear(self.n_neurons, self.n_neurons, bias=False), nn.ReLU(inplace=True)]
        layers += [nn.Linear(self.n_neurons, dim_out, bias=False)]
        self.layers = nn.Sequential(*layers)
        self.output_activation = get_activation(config.get("output_activation", None))

    def forward(self, x):
        with torch.autocast(device_type=x.device.type, enabled=False):
            return self.output_activation(self.layers(x))


def get_mlp(n_input_dims, n_output_dims, config) -> nn.Module:
    if config["otype"] == "VanillaMLP":
        return VanillaMLP(n_input_dims, n_output_dims, config_to_primitive(config))
    raise NotImplementedError(
        f"MLP otype {config['otype']!r}: every shipped config uses VanillaMLP (SURVEY.md §2.2 N3); "
        "tcnn.Network (FullyFusedMLP/CutlassMLP) is out of scope"
    )


def tcnn_grid_param_count(config: dict) -> int:
    c = config_to_primitive(config)
    m = _lib.make_grid_meta(int(c["n_levels"]), int(c.get("n_features_per_level", 2)), int(c["log2_hashmap_size"]),
                            int(c["base_resolution"]), float(c["per_level_scale"]))
    return int(m.n_params)


__all__ = ["Encoding", "TCNNEncoding", "CompositeEncoding", "get_encoding", "VanillaMLP", "get_mlp", "get_activation",
           "tcnn_grid_param_count", "math"]
