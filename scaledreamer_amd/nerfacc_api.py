"""nerfacc-compatible surface (the part ScaleDreamer calls, SURVEY.md §8b "nerfacc surface") on the HIP path.

  OccGridEstimator(roi_aabb, resolution=32, levels=1)     nerf_volume_renderer.py:60-65
      .occs / .binaries / .aabbs / .resolution             (buffers: they travel in checkpoints)
      .sampling(...) -> (ray_indices, t_starts, t_ends)    nerf_volume_renderer.py:139-180
      .update_every_n_steps(step=, occ_eval_fn=)           nerf_volume_renderer.py:442-444
  render_weight_from_density / render_weight_from_alpha / accumulate_along_rays
                                                           nerf_volume_renderer.py:313-349

Sample placement follows the convention documented in include/asd_hip.h (nerfacc's own lattice phase is
unpinned).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops


def packed_info_from_ray_indices(ray_indices: torch.Tensor, n_rays: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(offset, count) int32 per ray from sorted ray_indices."""
    count = torch.bincount(ray_indices.long(), minlength=n_rays).to(torch.int32)
    offset, _ = ops.scan_i32(count)
    return offset, count


class _CompositeFn(torch.autograd.Function):
    """All per-ray accumulations of the renderer in one pass (+ one backward pass)."""

    @staticmethod
    def forward(ctx, sigma, rgb, bg, t0, t1, offset, count, mode):
        sigma, rgb, bg = sigma.contiguous(), rgb.contiguous(), bg.contiguous()
        out = ops.composite_fwd(sigma, t0, t1, rgb, offset, count, bg, mode)
        ctx.save_for_backward(sigma, rgb, bg, t0, t1, offset, count, out["weights"], out["opacity"], out["depth"])
        ctx.mode = mode
        ctx.set_materialize_grads(False)
        return out["weights"], out["opacity"], out["depth"], out["rgb_fg"], out["z_var"], out["comp_rgb"]

    @staticmethod
    def backward(ctx, d_w, d_op, d_dp, d_fg, d_zv, d_comp):
        sigma, rgb, bg, t0, t1, offset, count, w, op, dp = ctx.saved_tensors
        fwd = dict(weights=w, opacity=op, depth=dp)
        d_sigma, d_rgb, d_bg = ops.composite_bwd(sigma, t0, t1, rgb, offset, count, bg, fwd, d_comp_rgb=d_comp,
                                                 d_rgb_fg=d_fg, d_opacity=d_op, d_depth=d_dp, d_z_var=d_zv,
                                                 d_weights=d_w, mode=ctx.mode)
        return d_sigma, d_rgb, d_bg, None, None, None, None, None


def composite(sigma, rgb, bg, t0, t1, offset, count, mode: int = 0):
    return _CompositeFn.apply(sigma, rgb, bg, t0, t1, offset, count, mode)


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                               prefix_trans=None):
    if prefix_trans is not None:
        raise NotImplementedError("prefix_trans")
    if packed_info is not None:
        offset, count = packed_info[:, 0].to(torch.int32).contiguous(), packed_info[:, 1].to(torch.int32).contiguous()
        n_rays = offset.shape[0]
    else:
        offset, count = packed_info_from_ray_indices(ray_indices, n_rays)
    n = sigmas.shape[0]
    zeros3 = sigmas.new_zeros((n, 3))
    bg = sigmas.new_zeros((n_rays, 3))
    t0, t1 = t_starts.contiguous().float(), t_ends.contiguous().float()
    weights = _CompositeFn.apply(sigmas, zeros3, bg, t0, t1, offset, count, 0)[0]
    with torch.no_grad():
        alphas = 1.0 - torch.exp(-sigmas * (t1 - t0))
        trans = weights / alphas.clamp_min(1e-10)
    return weights, trans, alphas


def render_weight_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
    if prefix_trans is not None:
        raise NotImplementedError("prefix_trans")
    if packed_info is not None:
        offset, count = packed_info[:, 0].to(torch.int32).contiguous(), packed_info[:, 1].to(torch.int32).contiguous()
        n_rays = offset.shape[0]
    else:
        offset, count = packed_info_from_ray_indices(ray_indices, n_rays)
    n = alphas.shape[0]
    t = alphas.new_zeros(n)
    weights = _CompositeFn.apply(alphas, alphas.new_zeros((n, 3)), alphas.new_zeros((n_rays, 3)), t, t, offset, count, 1)[0]
    with torch.no_grad():
        trans = weights / alphas.clamp_min(1e-10)
    return weights, trans


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    """zeros[n_rays, C].index_add_(0, ray_indices, w[:, None] * values)   """
    src = weights[..., None] if values is None else weights[..., None] * values
    out = torch.zeros((n_rays, src.shape[-1]), device=src.device, dtype=src.dtype)
    return out.index_add_(0, ray_indices.long(), src)


class OccGridEstimator(nn.Module):
    def __init__(self, roi_aabb, resolution=32, levels: int = 1):
        super().__init__()
        if levels != 1:
            raise NotImplementedError("the reference uses levels=1 (nerf_volume_renderer.py:60-62)")
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        if len(set(resolution)) != 1:
            raise NotImplementedError("cubic occupancy grids only")
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(-1)
        self.levels = levels
        self.cells_per_lvl = int(resolution[0]) ** 3
        self.register_buffer("resolution", torch.tensor(resolution, dtype=torch.int32))
        self.register_buffer("aabbs", roi_aabb[None].clone())
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + list(resolution), dtype=torch.bool))
        self.register_buffer("_occ_bits", torch.zeros((self.cells_per_lvl + 31) // 32, dtype=torch.int32), persistent=False)
        self._bits_version = -1
        self._occ_mean = 0.0  # host mirror of occs.mean(), refreshed whenever the grid is rebuilt

    @property
    def device(self):
        return self.occs.device

    # -- occupancy bit field (what the marcher reads) ------------------------------------------
    def _bits(self) -> torch.Tensor:
        v = self.binaries._version
        if self._bits_version != v or self._occ_bits.device != self.binaries.device:
            self._occ_bits = ops.pack_bits(self.binaries)
            self._occ_mean = float(self.occs.mean().item())
            self._bits_version = self.binaries._version
        return self._occ_bits

    def march_cfg(self, near_plane, far_plane, render_step_size) -> _lib.MarchCfg:
        c = _lib.MarchCfg()
        aabb = self.aabbs[0].tolist() if not hasattr(self, "_aabb_host") else self._aabb_host
        self._aabb_host = aabb
        for i in range(6):
            c.aabb[i] = aabb[i]
        c.resolution = int(round(self.cells_per_lvl ** (1 / 3)))
        c.near_plane, c.far_plane, c.step = float(near_plane), float(far_plane), float(render_step_size)
        diag = sum((aabb[3 + i] - aabb[i]) ** 2 for i in range(3)) ** 0.5
        c.max_steps = int(diag / float(render_step_size)) + 3
        return c

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn: Optional[Callable] = None, alpha_fn: Optional[Callable] = None,
                 near_plane: float = 0.0, far_plane: float = 1e10, t_min=None, t_max=None,
                 render_step_size: float = 1e-3, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                 stratified: bool = False, cone_angle: float = 0.0, jitter: Optional[torch.Tensor] = None):
        if cone_angle != 0.0:
            raise NotImplementedError("cone_angle != 0 is not used by the reference renderer")
        if t_min is not None or t_max is not None:
            raise NotImplementedError("per-ray t_min/t_max")
        if alpha_fn is not None:
            raise NotImplementedError("alpha_fn (the reference passes sigma_fn)")
        rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
        n_rays = rays_o.shape[0]
        if stratified and jitter is None:
            jitter = torch.rand(n_rays, device=rays_o.device)
        cfg = self.march_cfg(near_plane, far_plane, render_step_size)
        bits = self._bits()
        count, offset, total, ray_idx, t0, t1, pts = ops.march(cfg, rays_o, rays_d, bits, jitter)
        if sigma_fn is not None and (early_stop_eps > 0 or alpha_thre > 0):
            alpha_thre = min(alpha_thre, self._occ_mean)
            n_cand = ray_idx.shape[0]
            if n_cand > 0:
                sigmas = sigma_fn(t0, t1, ray_idx.long()).reshape(-1).contiguous().float()
                assert sigmas.shape[0] == n_cand, f"sigmas must have shape of (N,)! Got {sigmas.shape}"
            else:
                sigmas = t0.new_zeros(0)
            keep, kept = ops.prune(sigmas, t0, t1, offset, count, early_stop_eps, alpha_thre)
            koff, ktot = ops.scan_i32(kept)
            n_out = int(ktot.item())
            ri, k0, k1, _, _ = ops.compact(rays_o, rays_d, offset, count, keep, t0, t1, koff, n_out)
            return ri, k0, k1
        return ray_idx.long(), t0, t1

    # -- maintenance -----------------------------------------------------------------------------
    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16, rng: Optional[torch.Generator] = None) -> None:
        if not self.training:
            raise RuntimeError("Please call `estimator.train()` before calling `update_every_n_steps()` during training.")
        if step % n == 0 and self.training:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps, rng)

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre, ema_decay, warmup_steps, rng=None):
        dev, n_cells = self.occs.device, self.cells_per_lvl
        res = int(round(n_cells ** (1 / 3)))
        if step < warmup_steps:
            idx = torch.arange(n_cells, device=dev)
        else:
            N = n_cells // 4
            uniform = torch.randint(n_cells, (N,), device=dev, generator=rng)
            occupied = torch.nonzero(self.binaries.reshape(-1))[:, 0]
            if occupied.numel() > N:
                occupied = occupied[torch.randint(occupied.numel(), (N,), device=dev, generator=rng)]
            idx = torch.unique(torch.cat([uniform, occupied]))
        coords = torch.stack([idx // (res * res), (idx // res) % res, idx % res], -1).float()
        x = (coords + torch.rand(coords.shape, device=dev, generator=rng)) / res
        aabb = self.aabbs[0]
        x = aabb[:3] + x * (aabb[3:] - aabb[:3])
        occ = occ_eval_fn(x).reshape(-1).contiguous().float()
        binaries_u8 = torch.empty(n_cells, device=dev, dtype=torch.uint8)
        bits = torch.empty((n_cells + 31) // 32, device=dev, dtype=torch.int32)
        ops.occgrid_update(self.occs, idx.to(torch.int32).contiguous(), occ, ema_decay, occ_thre, bits, binaries_u8)
        self.binaries.copy_(binaries_u8.view(self.binaries.shape).bool())
        self._occ_bits = bits
        self._occ_mean = float(self.occs.mean().item())
        self._bits_version = self.binaries._version
