"""Feature samplers of the generator-backed geometries, drop-in for custom/amortized/models/geometry/utils.py:
`contract_to_unisphere_custom` (:15-27), `sample_from_planes` (:81-93), `get_trilinear_feature` (:95-110) — same arguments
(channel-FIRST feature tensors as the generators emit them), executed by the channel-last HIP gather / scatter kernels
(include/asd_hip.h: asd_voxel_sample_*, asd_triplane_sample_*, asd_relayout_f32).  The relayout of a feature volume is
cached per tensor version, because one step samples the same volume several times (proposal pass, main pass, 3 FD offsets).
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from .geometry import scale_tensor


def contract_to_unisphere_custom(x: torch.Tensor, bbox: torch.Tensor, unbounded: bool = False) -> torch.Tensor:
    if unbounded:
        x = scale_tensor(x, bbox, (-1, 1))
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        mask = mag.squeeze(-1) > 1
        x[mask] = (2 - 1 / mag[mask]) * (x[mask] / mag[mask])
        return x / 4 + 0.5
    return scale_tensor(x, bbox, (-1, 1))


class _ChannelsLast(torch.autograd.Function):
    """[B, C, *S] -> [B, *S, C] (and back in backward) through the tiled transpose kernel."""

    @staticmethod
    def forward(ctx, x):
        B, Cc = x.shape[:2]
        ctx.shape = x.shape
        return ops.relayout(x.reshape(B, Cc, -1)).view(B, *x.shape[2:], Cc)

    @staticmethod
    def backward(ctx, dy):
        B, Cc = ctx.shape[:2]
        return ops.relayout(dy.reshape(B, -1, Cc)).view(ctx.shape)


_cl_cache = {}   # id(tensor) -> (weakref to it, its version, channel-last copy); tensors compare element-wise, so no tensor keys


def _cached(x: torch.Tensor, make):
    hit = _cl_cache.get(id(x))
    need_graph = torch.is_grad_enabled() and x.requires_grad
    if hit is not None and hit[0]() is x and hit[1] == x._version and (hit[2].requires_grad or not need_graph):
        return hit[2]      # (a copy made under no_grad — the proposal pass — must not serve the differentiable pass)
    y = make(x)
    for k in [k for k, v in _cl_cache.items() if v[0]() is None]:   # drop entries of freed tensors
        del _cl_cache[k]
    _cl_cache[id(x)] = (weakref.ref(x), x._version, y)
    return y


def channels_last(x: torch.Tensor) -> torch.Tensor:
    """[B, C, *S] -> [B, *S, C], cached per tensor version (one step samples the same volume several times).  A tensor that already
    is channel-last in memory (the HIP generator's output: a permuted view) is handed on as that view — no kernel, no copy."""
    cl = x.movedim(1, -1)
    if cl.is_contiguous():
        return cl
    return _cached(x, _ChannelsLast.apply)


def planes_channels_last(planes: torch.Tensor) -> torch.Tensor:
    """[N, 3, C, H, W] -> [N, 3, H, W, C], cached per tensor version.  Planes that already are channel-last in memory (the HIP transformer's
    output: a permuted view) are handed on as that view — no kernel, no copy."""
    cl = planes.permute(0, 1, 3, 4, 2)
    if cl.is_contiguous():
        return cl

    def make(p):
        N, _, Cc, H, W = p.shape
        return _ChannelsLast.apply(p.reshape(N * 3, Cc, H, W)).view(N, 3, H, W, Cc)
    return _cached(planes, make)


class _VoxelSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxel_cl, points):
        ctx.save_for_backward(points)
        ctx.shape = tuple(voxel_cl.shape)
        return ops.voxel_sample_fwd(voxel_cl.contiguous(), points)

    @staticmethod
    def backward(ctx, d_out):
        (points,) = ctx.saved_tensors
        return ops.voxel_sample_bwd(d_out.contiguous(), points, ctx.shape), None


class _TriplaneSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes_cl, points, coord_scale):
        ctx.save_for_backward(points)
        ctx.shape, ctx.coord_scale = tuple(planes_cl.shape), coord_scale
        return ops.triplane_sample_fwd(planes_cl.contiguous(), points, coord_scale)

    @staticmethod
    def backward(ctx, d_out):
        (points,) = ctx.saved_tensors
        return ops.triplane_sample_bwd(d_out.contiguous(), points, ctx.shape, ctx.coord_scale), None, None


def get_trilinear_feature(points: torch.Tensor, voxel: torch.Tensor) -> torch.Tensor:
    """points [B, ..., 3] in [-1,1], voxel [B, Df, G1, G2, G3] -> [B, ..., Df].  (For B > 1 the reference's final
    reshape(df, -1).T interleaves batch and channel; this returns the per-batch features the B = 1 case defines.)"""
    B = voxel.shape[0]
    shape = points.shape[:-1]
    out = _VoxelSample.apply(channels_last(voxel), points.reshape(B, -1, 3).float())
    return out.reshape(*shape, voxel.shape[1])


def sample_from_planes(plane_features: torch.Tensor, coordinates: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
                       box_warp: float = 2) -> torch.Tensor:
    """plane_features [N, 3, C, H, W], coordinates [N, M, 3] -> [N, M, 3C]"""
    assert padding_mode == "zeros" and mode == "bilinear"
    N, n_planes, Cc, H, W = plane_features.shape
    assert n_planes == 3
    return _TriplaneSample.apply(planes_channels_last(plane_features), coordinates.float(), 2.0 / box_warp)
