"""`nerf-volume-renderer` (threestudio/models/renderers/nerf_volume_renderer.py:20-470) on the HIP path.

Same Config, forward signature and output dictionary as the reference class.  With the occupancy-grid
estimator (the shipped single-prompt configs) the sampling / pruning / compaction / compositing all run in
the HIP kernels of csrc/render.hip; geometry and background are called through their module interface, so
any registered geometry works, and the HIP `implicit-volume` / `neural-environment-map-background` plug in
with their fused kernels.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn.functional as F

import ctypes as C

from . import _lib, nerfacc_api, ops
from .base import BaseModule
from .registry import register, warn


class Renderer(BaseModule):
    """threestudio/models/renderers/base.py:15-72"""

    @dataclass
    class Config(BaseModule.Config):
        radius: float = 1.0

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        @dataclass
        class SubModules:
            geometry: Any
            material: Any
            background: Any

        self.sub_modules = SubModules(geometry, material, background)
        r = self.cfg.radius
        self.register_buffer("bbox", torch.as_tensor([[-r, -r, -r], [r, r, r]], dtype=torch.float32))

    @property
    def geometry(self):
        return self.sub_modules.geometry

    @property
    def material(self):
        return self.sub_modules.material

    @property
    def background(self):
        return self.sub_modules.background

    def set_geometry(self, geometry) -> None:
        self.sub_modules.geometry = geometry

    def set_material(self, material) -> None:
        self.sub_modules.material = material

    def set_background(self, background) -> None:
        self.sub_modules.background = background


class VolumeRenderer(Renderer):
    pass


def chunk_batch(func, chunk_size: int, *args, **kwargs):
    """threestudio/utils/ops.py:116-180 for tensor / dict-of-tensor outputs."""
    if chunk_size <= 0:
        return func(*args, **kwargs)
    B = next(a.shape[0] for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor))
    outs = []
    for i in range(0, max(1, B), chunk_size):
        sl = lambda a: a[i:i + chunk_size] if isinstance(a, torch.Tensor) else a
        outs.append(func(*[sl(a) for a in args], **{k: sl(v) for k, v in kwargs.items()}))
    if isinstance(outs[0], dict):
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    return torch.cat(outs, 0)


class LazyOutputs(dict):
    """Output dictionary whose expensive entries nobody may ever read are produced on first access.

    The reference returns `normal` / `shading_normal` (three extra hash encodes per kept sample: the finite-difference normal,
    implicit_volume.py:137-177) whenever `material.requires_normal` is set (nerf_volume_renderer.py:281-283), although the shipped
    asd_sd_nerf config neither shades with them (NoMaterial) nor weights a loss on them (lambda_orient = 0, asd_sd_nerf.yaml:106).
    Same surface here — the keys exist, `in`, indexing, `.get`, `.items()`, `**out` all work and hand out real, differentiable
    tensors — but an entry registered with `defer` costs nothing until somebody asks for it."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._deferred = {}

    def defer(self, keys, produce):
        """`produce()` -> dict holding every key of `keys`; called at most once, on the first access to any of them"""
        cell = {"fn": produce, "keys": tuple(keys)}
        for k in keys:
            self._deferred[k] = cell

    def _force(self, key):
        cell = self._deferred.get(key)
        if cell is not None:
            vals = cell["fn"]()
            for k in cell["keys"]:
                if self._deferred.get(k) is cell:          # (a key written or deleted by the user meanwhile stays as the user left it)
                    del self._deferred[k]
                    dict.__setitem__(self, k, vals[k])

    def __getitem__(self, key):
        self._force(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._deferred

    # writes replace a deferred entry instead of being overwritten by it on the next read
    def __setitem__(self, key, value):
        self._deferred.pop(key, None)
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        if self._deferred.pop(key, None) is None or dict.__contains__(self, key):
            dict.__delitem__(self, key)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    _MISSING = object()

    def pop(self, key, default=_MISSING):
        if key in self:
            v = self[key]
            dict.__delitem__(self, key)
            return v
        if default is LazyOutputs._MISSING:
            raise KeyError(key)
        return default

    def __iter__(self):                    # (also sends `{**out}` / dict(out) down the keys() + __getitem__ path); iterates over a
        yield from list(dict.__iter__(self)) + list(self._deferred)     # snapshot: forcing an entry inside the loop moves its key

    def __len__(self):
        return dict.__len__(self) + len(self._deferred)

    def keys(self):
        return list(iter(self))

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def pending(self):
        """keys not produced yet (tests / tools)"""
        return set(self._deferred)


class _RenderPass:
    """state of one fused training pass (include/asd_hip.h: asd_render_fwd / asd_render_bwd): the parameter block, the tensors its
    pointers refer to (kept alive until the backward pass) and typed views into the workspace"""

    _DT = {"count": torch.int32, "offset": torch.int32, "total": torch.int32, "kept": torch.int32, "koff": torch.int32, "n_kept": torch.int32,
           "ray_idx": torch.int64, "keep": torch.uint8, "c_ray_idx": torch.int32}

    def __init__(self, march_cfg, meta, fcfg, rays_o, rays_d, bits, jitter, early_stop_eps, alpha_thre, prune, color_act, capacity):
        self.march_cfg, self.meta, self.fcfg = march_cfg, meta, fcfg
        self.rays_o, self.rays_d, self.bits, self.jitter = rays_o, rays_d, bits, jitter
        self.n_rays, self.capacity = rays_o.shape[0], int(capacity)
        self.early_stop_eps, self.alpha_thre, self.prune, self.color_act = float(early_stop_eps), float(alpha_thre), int(prune), int(color_act)
        self.layout = _lib.RenderLayout()
        _lib.check(_lib.lib().asd_render_layout_init(_lib.i32(self.n_rays), _lib.i32(self.capacity), C.byref(self.layout)))
        self.ws = None

    def params(self, grid, w1d, w2d, w1f, w2f, bg) -> "_lib.RenderParams":
        p = _lib.RenderParams()
        p.march = self.march_cfg
        p.meta, p.field = C.addressof(self.meta), C.addressof(self.fcfg)
        p.rays_o, p.rays_d, p.n_rays = self.rays_o.data_ptr(), self.rays_d.data_ptr(), self.n_rays
        p.occ_bits, p.jitter = self.bits.data_ptr(), (None if self.jitter is None else self.jitter.data_ptr())
        p.grid, p.w1d, p.w2d, p.w1f, p.w2f, p.bg = (t.data_ptr() for t in (grid, w1d, w2d, w1f, w2f, bg))
        p.early_stop_eps, p.alpha_thre, p.prune, p.color_act, p.capacity = self.early_stop_eps, self.alpha_thre, self.prune, self.color_act, self.capacity
        return p

    def view(self, name: str, cols: int = 0, rows: Optional[int] = None) -> torch.Tensor:
        """typed view of a workspace buffer: per-ray buffers have n_rays rows, per-sample ones `capacity` (or the first `rows`)"""
        dt = self._DT.get(name, torch.float32)
        per_ray = name in ("count", "offset", "kept", "koff", "opacity", "depth", "z_var", "rgb_fg", "comp_rgb")
        n = 1 if name in ("total", "n_kept") else (self.n_rays if per_ray else self.capacity)
        if rows is not None:
            n = rows
        width = max(cols, 1)
        off = getattr(self.layout, name)
        nbytes = n * width * torch.empty(0, dtype=dt).element_size()
        t = self.ws[off:off + nbytes].view(dt)
        return t.view(n, cols) if cols else t


class _RenderFn(torch.autograd.Function):
    """the whole training pass of the renderer as one autograd node: (hash table, four MLP weights, per-ray background) -> (comp_rgb,
    comp_rgb_fg, opacity, depth, z_variance); nothing in between is a Python-issued launch"""

    @staticmethod
    def forward(ctx, grid, w1d, w2d, w1f, w2f, bg, st):
        st.ws = torch.empty(st.layout.total_bytes, dtype=torch.uint8, device=grid.device)
        p = st.params(grid, w1d, w2d, w1f, w2f, bg)
        _lib.check(_lib.lib().asd_render_fwd(C.byref(p), _lib.ptr(st.ws), _lib.stream()))
        ctx.st = st
        ctx.save_for_backward(grid, w1d, w2d, w1f, w2f, bg)
        ctx.set_materialize_grads(False)
        # asd_render_bwd re-reads opacity / depth / weights from st.ws, which autograd does not version-check: the caller gets COPIES of the
        # five per-ray outputs (they are the contiguous tail of the layout: one n_rays * ~36 B copy), so an in-place op on a returned tensor
        # cannot corrupt the gradients and a surviving output does not pin the n_rays * max_steps * ~217 B workspace
        L, n = st.layout, st.n_rays
        tail = st.ws[L.opacity:L.c_feats].clone()

        def out(off, cols):
            t = tail[off - L.opacity:off - L.opacity + n * max(cols, 1) * 4].view(torch.float32)
            return t.view(n, cols) if cols else t
        return out(L.comp_rgb, 3), out(L.rgb_fg, 3), out(L.opacity, 0), out(L.depth, 0), out(L.z_var, 0)

    @staticmethod
    def backward(ctx, d_comp, d_fg, d_op, d_dp, d_zv):
        grid, w1d, w2d, w1f, w2f, bg = ctx.saved_tensors
        st = ctx.st
        d_grid = torch.zeros_like(grid)
        ws_ = (w1d, w2d, w1f, w2f)           # the four small gradients: views of ONE zeroed buffer (16-byte aligned pieces)
        offs, total = [], 0
        for w in ws_:
            offs.append(total)
            total += (w.numel() + 3) // 4 * 4
        flat = torch.zeros(total, device=grid.device, dtype=torch.float32)
        dws = [flat[o:o + w.numel()].view_as(w) for o, w in zip(offs, ws_)]
        if all(g is None for g in (d_comp, d_fg, d_op, d_dp, d_zv)):
            return (d_grid, *dws, None, None)
        p = st.params(grid, w1d, w2d, w1f, w2f, bg)
        nf = C.c_int64(0)
        _lib.check(_lib.lib().asd_render_bwd_workspace(C.byref(p), C.byref(nf)))
        bws = torch.empty(nf.value, device=grid.device, dtype=torch.float32)
        d_bg = torch.empty_like(bg) if ctx.needs_input_grad[5] else None
        k = ops._Keep()
        _lib.check(_lib.lib().asd_render_bwd(C.byref(p), _lib.ptr(st.ws), k(d_comp), k(d_fg), k(d_op), k(d_dp), k(d_zv), _lib.ptr(d_grid),
                                             *(_lib.ptr(t) for t in dws), _lib.ptr(d_bg), _lib.ptr(bws), _lib.stream()))
        if d_bg is not None and d_comp is None:
            d_bg.zero_()
        return (d_grid, *dws, d_bg, None)


def validate_empty_rays(ray_indices, t_start, t_end):
    """threestudio/utils/ops.py:514-520: substitute one dummy sample when nothing was sampled."""
    if ray_indices.nelement() == 0:
        warn("Empty rays_indices!")
        ray_indices = torch.zeros(1, dtype=torch.long, device=ray_indices.device)
        t_start = torch.zeros(1, device=ray_indices.device)
        t_end = torch.zeros(1, device=ray_indices.device)
    return ray_indices, t_start, t_end


@register("nerf-volume-renderer")
class NeRFVolumeRenderer(VolumeRenderer):
    @dataclass
    class Config(VolumeRenderer.Config):
        num_samples_per_ray: int = 512
        eval_chunk_size: int = 160000
        randomized: bool = True
        near_plane: float = 0.0
        far_plane: float = 1e10
        return_comp_normal: bool = False
        return_normal_perturb: bool = False
        estimator: str = "occgrid"  # in ["occgrid", "proposal", "importance"]
        grid_prune: bool = True
        prune_alpha_threshold: bool = True
        proposal_network_config: Optional[dict] = None
        prop_optimizer_config: Optional[dict] = None
        prop_scheduler_config: Optional[dict] = None
        num_samples_per_ray_proposal: int = 64
        num_samples_per_ray_importance: int = 64

    cfg: Config
    # upper bound of the sync-free candidate buffers (~28 B per candidate across ray_idx / t0 / t1 / points / sigma / keep: 1.3 GB);
    # also keeps every 3-component index of those buffers far below 2^31
    MAX_CANDIDATE_CAPACITY = 48 << 20

    def configure(self, geometry, material, background) -> None:
        super().configure(geometry, material, background)
        if self.cfg.estimator == "occgrid":
            self.estimator = nerfacc_api.OccGridEstimator(roi_aabb=self.bbox.view(-1), resolution=32, levels=1)
            if not self.cfg.grid_prune:
                self.estimator.occs.fill_(True)
                self.estimator.binaries.fill_(True)
            self.render_step_size = 1.732 * 2 * self.cfg.radius / self.cfg.num_samples_per_ray
            self.randomized = self.cfg.randomized
        elif self.cfg.estimator in ("importance", "proposal"):
            raise NotImplementedError(
                f"estimator {self.cfg.estimator!r}: the single-prompt ASD configs use 'occgrid' "
                "(importance sampling belongs to the amortized renderer; proposal is unused — SURVEY.md §2.2 N9/N10)"
            )
        else:
            raise NotImplementedError("Unknown estimator, should be one of ['occgrid', 'proposal', 'importance'].")
        self.vars_in_forward: Dict[str, Any] = {}
        self.jitter_fn = lambda n, device: torch.rand(n, device=device)  # injectable

    # ------------------------------------------------------------------------------------------
    def _sample(self, rays_o_flatten, rays_d_flatten, sync_free: bool = False):
        """(ray_indices int64, t_starts, t_ends, points, dirs, offset int32, count int32, total) of the kept samples.  `total` is None
        and the sample tensors have exactly the kept length — or, with sync_free (and the fused candidate path), `total` is the kept
        count as an int32 DEVICE scalar and the sample tensors have the candidates' capacity: only their first `total` rows are
        written (ray_indices reads 0 behind them), and no device->host read happens in here at all."""
        n_rays = rays_o_flatten.shape[0]
        est = self.estimator
        jitter = self.jitter_fn(n_rays, rays_o_flatten.device) if self.randomized else None
        cfg = est.march_cfg(self.cfg.near_plane, self.cfg.far_plane, self.render_step_size)
        bits = est._bits()
        # Candidate buffers at their upper bound (every ray at most max_steps lattice points) with the count left on the device:
        # the pruning pass below only ever looks at [offset, offset + count) of each ray, and the density kernel takes the total
        # as a device scalar — no device->host read of the candidate count (one of the two syncs of this method; the kept count
        # below stays, it sizes every tensor of the output dictionary).
        prune = self.cfg.grid_prune and self.cfg.prune_alpha_threshold
        fused_density = self.training and prune and getattr(self.geometry, "fused", False) and rays_o_flatten.is_cuda
        n_cap = n_rays * int(cfg.max_steps) if fused_density else None
        if n_cap is not None and n_cap > self.MAX_CANDIDATE_CAPACITY:
            n_cap = None     # large batches (256^2 x 4 views: 1.3e8 lattice points): exact-size buffers after one count read-back
        count, offset, total, ray_idx, t0, t1, pts = ops.march(cfg, rays_o_flatten, rays_d_flatten, bits, jitter, n_max=n_cap)
        if self.cfg.grid_prune:
            early_stop_eps, alpha_thre = 1e-4, (0.01 if self.cfg.prune_alpha_threshold else 0.0)
        else:
            early_stop_eps, alpha_thre = 0.0, 0.0
        if prune and (early_stop_eps > 0 or alpha_thre > 0):
            alpha_thre = min(alpha_thre, est._occ_mean)
            if ray_idx.shape[0] > 0:
                # sigma at the candidate mid-points (the reference's sigma_fn, nerf_volume_renderer.py:153-167);
                # the marcher already produced the positions o + d*(t0+t1)/2
                if fused_density:
                    sigma = self.geometry.forward_density(pts, n_dev=total)[..., 0]
                elif self.training:
                    sigma = self.geometry.forward_density(pts)[..., 0]
                else:
                    sigma = chunk_batch(self.geometry.forward_density, self.cfg.eval_chunk_size, pts)[..., 0]
                sigma = sigma.contiguous().float()
            else:
                sigma = t0.new_zeros(0)
            keep, kept = ops.prune(sigma, t0, t1, offset, count, early_stop_eps, alpha_thre)
            koff, ktot = ops.scan_i32(kept)
            if sync_free and fused_density and n_cap is not None and ray_idx.shape[0] > 0:
                ri, k0, k1, kp, kd = ops.compact(rays_o_flatten, rays_d_flatten, offset, count, keep, t0, t1, koff, ray_idx.shape[0],
                                                 zero_ray_idx=True)
                return ri, k0, k1, kp, kd, koff, kept, ktot.reshape(1)
            n_out = int(ktot.item())
            ri, k0, k1, kp, kd = ops.compact(rays_o_flatten, rays_d_flatten, offset, count, keep, t0, t1, koff, n_out)
            return ri, k0, k1, kp, kd, koff, kept, None
        n_out = ray_idx.shape[0]
        ri, k0, k1, kp, kd = ops.compact(rays_o_flatten, rays_d_flatten, offset, count, None, t0, t1, offset, n_out)
        return ri, k0, k1, kp, kd, offset, count, None

    def forward(self, rays_o: torch.Tensor, rays_d: torch.Tensor, light_positions: torch.Tensor,
                bg_color: Optional[torch.Tensor] = None, **kwargs) -> Dict[str, torch.Tensor]:
        batch_size, height, width = rays_o.shape[:3]
        rays_o_flatten = rays_o.reshape(-1, 3).contiguous().float()
        rays_d_flatten = rays_d.reshape(-1, 3).contiguous().float()
        light_positions_flatten = light_positions.reshape(-1, 1, 1, 3).expand(-1, height, width, -1).reshape(-1, 3)
        n_rays = rays_o_flatten.shape[0]

        # Sync-free training path: the kept-sample count stays on the device (the field kernels take it as `n_dev`, compositing is per
        # ray), the per-sample tensors are capacity-sized, and the per-sample entries of the output dictionary are cut to their exact
        # length only if somebody reads them (LazyOutputs) — the host never waits for the GPU inside a step, so it runs ahead of it and
        # its launch latencies disappear from the step (the read-back cost 0.1 ms of GPU idle time plus a host-bound renderer forward).
        # Needs: fused field kernels, a material that is a row-wise function of the features, no consumer of the normal in here.
        sync_free = (self.training and os.environ.get("ASD_SYNC_FREE", "1") != "0" and getattr(self.geometry, "fused", False)
                     and rays_o_flatten.is_cuda and getattr(self.material, "elementwise", False) and not self.cfg.return_comp_normal
                     and not self.cfg.return_normal_perturb
                     and not (self.material.requires_normal and getattr(self.material, "reads_normal", True)))
        if sync_free and self._fused_pass_ok(n_rays):
            return self._forward_fused_pass(batch_size, height, width, rays_o_flatten, rays_d_flatten, rays_d, bg_color, kwargs)
        with torch.no_grad():
            ray_indices, t_starts_, t_ends_, positions, t_dirs, offset, count, n_dev = self._sample(rays_o_flatten, rays_d_flatten, sync_free)
        if n_dev is not None:
            return self._forward_sync_free(batch_size, height, width, rays_d, bg_color, ray_indices, t_starts_, t_ends_, positions, t_dirs,
                                           offset, count, n_dev, kwargs)
        if ray_indices.nelement() == 0:
            ray_indices, t_starts_, t_ends_ = validate_empty_rays(ray_indices, t_starts_, t_ends_)
            positions = rays_o_flatten[ray_indices] + rays_d_flatten[ray_indices] * 0.0
            t_dirs = rays_d_flatten[ray_indices]
            count = torch.zeros(n_rays, dtype=torch.int32, device=rays_o.device)
            count[0] = 1
            offset = torch.ones(n_rays, dtype=torch.int32, device=rays_o.device)
            offset[0] = 0
        self._last_n = int(ray_indices.shape[0])
        t_starts, t_ends = t_starts_[..., None], t_ends_[..., None]
        t_light_positions = light_positions_flatten[ray_indices]
        t_positions = (t_starts + t_ends) / 2.0
        t_intervals = t_ends - t_starts

        lazy_normal = None
        if self.training:
            # the normal is evaluated with the field only if something inside this call reads it (a shading material, comp_normal,
            # normal_perturb); otherwise it becomes a deferred entry of the output dictionary (LazyOutputs)
            want_normal = bool(self.material.requires_normal)
            now = want_normal and (getattr(self.material, "reads_normal", True) or self.cfg.return_comp_normal or self.cfg.return_normal_perturb)
            geo_out = self.geometry(positions, output_normal=now)
            if want_normal and not now:
                def lazy_normal(geometry=self.geometry, pts=positions):
                    g = geometry(pts, output_normal=True)
                    return {"normal": g["normal"], "shading_normal": g["shading_normal"]}
            rgb_fg_all = self.material(viewdirs=t_dirs, positions=positions, light_positions=t_light_positions,
                                       **geo_out, **kwargs)
            comp_rgb_bg = self.background(dirs=rays_d)
        else:
            geo_out = chunk_batch(self.geometry, self.cfg.eval_chunk_size, positions,
                                  output_normal=self.material.requires_normal)
            rgb_fg_all = chunk_batch(self.material, self.cfg.eval_chunk_size, viewdirs=t_dirs, positions=positions,
                                     light_positions=t_light_positions, **geo_out)
            comp_rgb_bg = chunk_batch(self.background, self.cfg.eval_chunk_size, dirs=rays_d)

        if bg_color is None:
            bg_color = comp_rgb_bg
        else:
            if bg_color.shape[:-1] == (batch_size,):
                bg_color = bg_color.unsqueeze(1).unsqueeze(1).expand(-1, height, width, -1)
        if bg_color.shape[:-1] == (batch_size, height, width):
            bg_color = bg_color.reshape(batch_size * height * width, -1)

        # T / alpha / weights + all per-ray accumulations (reference :312-364) in one fused pass
        weights_, opacity_, depth_, comp_rgb_fg, z_variance_, comp_rgb = nerfacc_api.composite(
            geo_out["density"][..., 0], rgb_fg_all, bg_color.float(), t_starts_, t_ends_, offset, count, 0
        )
        weights = weights_[..., None]
        opacity, depth, z_variance = opacity_[..., None], depth_[..., None], z_variance_[..., None]

        out = LazyOutputs({
            "comp_rgb": comp_rgb.view(batch_size, height, width, -1),
            "comp_rgb_fg": comp_rgb_fg.view(batch_size, height, width, -1),
            "comp_rgb_bg": comp_rgb_bg.view(batch_size, height, width, -1),
            "opacity": opacity.view(batch_size, height, width, 1),
            "depth": depth.view(batch_size, height, width, 1),
            "z_variance": z_variance.view(batch_size, height, width, 1),
        })

        def comp_normal_of(normal):
            cn = nerfacc_api.accumulate_along_rays(weights[..., 0], values=normal, ray_indices=ray_indices, n_rays=n_rays)
            cn = F.normalize(cn, dim=-1)
            return ((cn + 1.0) / 2.0 * opacity).view(batch_size, height, width, 3)

        if self.training:
            out.update({"weights": weights, "t_points": t_positions, "t_intervals": t_intervals, "t_dirs": t_dirs,
                        "ray_indices": ray_indices, "points": positions, **geo_out})
            if lazy_normal is not None:
                out.defer(("normal", "shading_normal"), lazy_normal)
            if "normal" in geo_out:
                if self.cfg.return_comp_normal:
                    out["comp_normal"] = comp_normal_of(geo_out["normal"])
                if self.cfg.return_normal_perturb:
                    out["normal_perturb"] = self.geometry(positions + torch.randn_like(positions) * 1e-2,
                                                          output_normal=self.material.requires_normal)["normal"]
        elif "normal" in geo_out:
            out["comp_normal"] = comp_normal_of(geo_out["normal"])
        return out

    # ---- the training pass as one C-ABI entry each way ---------------------------------------------------------------------------
    _COLOR_ACTS = {"sigmoid": 1, "none": 0, None: 0}

    def _fused_pass_ok(self, n_rays: int) -> bool:
        """asd_render_fwd covers: occupancy-grid sampling, a density field on the fused kernels with 3 feature dims, colour =
        sigmoid(features) | features, no normal consumer inside the call (the conditions of the sync-free path, narrowed)"""
        geo, mat = self.geometry, self.material
        fc = getattr(geo, "_fcfg", None)
        return (os.environ.get("ASD_RENDER_ENTRY", "1") != "0" and fc is not None and hasattr(geo, "_weights") and hasattr(geo, "_meta")
                and getattr(geo.cfg, "n_feature_dims", 0) == 3 and fc.field_mode == _lib.ASD_FIELD_DENSITY
                and getattr(mat.cfg, "color_activation", "?") in self._COLOR_ACTS and torch.is_grad_enabled()
                and geo.encoding.encoding.encoding.params.requires_grad
                and n_rays * int(self.estimator.march_cfg(self.cfg.near_plane, self.cfg.far_plane, self.render_step_size).max_steps) <= self.MAX_CANDIDATE_CAPACITY)

    def _forward_fused_pass(self, batch_size, height, width, rays_o_flatten, rays_d_flatten, rays_d, bg_color, kwargs):
        n_rays = rays_o_flatten.shape[0]
        est, geo = self.estimator, self.geometry
        jitter = self.jitter_fn(n_rays, rays_o_flatten.device) if self.randomized else None
        if jitter is not None:
            jitter = jitter.contiguous().float()
        mcfg = est.march_cfg(self.cfg.near_plane, self.cfg.far_plane, self.render_step_size)
        bits = est._bits()                       # (also refreshes the host mirror of the mean occupancy read below)
        prune = self.cfg.grid_prune and self.cfg.prune_alpha_threshold
        early_stop_eps, alpha_thre = (1e-4, min(0.01, est._occ_mean)) if prune else (0.0, 0.0)
        st = _RenderPass(mcfg, geo._meta, geo._fcfg, rays_o_flatten, rays_d_flatten, bits, jitter, early_stop_eps, alpha_thre, prune,
                         self._COLOR_ACTS[self.material.cfg.color_activation], n_rays * int(mcfg.max_steps))
        comp_rgb_bg = self.background(dirs=rays_d)
        if bg_color is None:
            bg_color = comp_rgb_bg
        elif bg_color.shape[:-1] == (batch_size,):
            bg_color = bg_color.unsqueeze(1).unsqueeze(1).expand(-1, height, width, -1)
        if bg_color.shape[:-1] == (batch_size, height, width):
            bg_color = bg_color.reshape(n_rays, -1)
        grid = geo.encoding.encoding.encoding.params
        comp_rgb, comp_rgb_fg, opacity, depth, z_var = _RenderFn.apply(grid, *geo._weights(), bg_color.float().contiguous(), st)
        n_dev = st.view("n_kept")
        self._last_n = n_dev
        out = LazyOutputs({
            "comp_rgb": comp_rgb.view(batch_size, height, width, -1),
            "comp_rgb_fg": comp_rgb_fg.view(batch_size, height, width, -1),
            "comp_rgb_bg": comp_rgb_bg.view(batch_size, height, width, -1),
            "opacity": opacity.view(batch_size, height, width, 1),
            "depth": depth.view(batch_size, height, width, 1),
            "z_variance": z_var.view(batch_size, height, width, 1),
        })
        material, bg_flat = self.material, bg_color

        def per_sample():
            """the per-sample entries, when somebody reads them: ONE read of the kept count, then the composed (differentiable) evaluation of
            the same samples — field, material, compositing weights — so that a loss on them reaches the parameters as in the reference"""
            n = int(n_dev.item())
            if n == 0:                          # the reference's dummy sample of an empty batch (validate_empty_rays)
                z = rays_o_flatten.new_zeros(1)
                pts0 = rays_o_flatten[:1].detach()
                g = geo(pts0, output_normal=False)
                o = {"weights": z[..., None], "t_points": z[..., None], "t_intervals": z[..., None], "t_dirs": rays_d_flatten[:1],
                     "ray_indices": torch.zeros(1, dtype=torch.long, device=z.device), "points": pts0}
                o.update({k: v * 0.0 for k, v in g.items()})
                return o
            t0, t1, pts = st.view("t0", rows=n), st.view("t1", rows=n), st.view("pts", 3, rows=n)
            g = geo(pts, output_normal=False)
            rgb = material(viewdirs=st.view("dirs", 3, rows=n), positions=pts, light_positions=None, **g, **kwargs)
            w = nerfacc_api.composite(g["density"][..., 0], rgb, bg_flat.float().detach(), t0, t1, st.view("koff"), st.view("kept"), 0)[0]
            o = {"weights": w[..., None], "t_points": ((t0 + t1) / 2.0)[..., None], "t_intervals": (t1 - t0)[..., None],
                 "t_dirs": st.view("dirs", 3, rows=n), "ray_indices": st.view("ray_idx", rows=n), "points": pts}
            o.update(g)
            return o

        geo_keys = ("density", "features")
        out.defer(("weights", "t_points", "t_intervals", "t_dirs", "ray_indices", "points") + geo_keys, per_sample)
        if self.material.requires_normal:
            def lazy_normal():
                n = int(n_dev.item())
                pts = st.view("pts", 3, rows=n) if n > 0 else rays_o_flatten[:1].detach()
                g = geo(pts, output_normal=True)
                return {"normal": g["normal"], "shading_normal": g["shading_normal"]}
            out.defer(("normal", "shading_normal"), lazy_normal)
        return out

    @property
    def last_n_samples(self) -> int:
        """kept samples of the last forward pass (reads the device count of a sync-free pass when asked)"""
        n = getattr(self, "_last_n", 0)
        return int(n.item()) if torch.is_tensor(n) else int(n)

    def _forward_sync_free(self, batch_size, height, width, rays_d, bg_color, ray_indices, t_starts_, t_ends_, positions, t_dirs,
                           offset, count, n_dev, kwargs):
        """forward() behind the sampler with the kept count on the device (see there): same arithmetic on the first n_dev rows"""
        n_rays = batch_size * height * width
        self._last_n = n_dev
        geo_out = self.geometry(positions, output_normal=False, n_dev=n_dev)
        rgb_fg_all = self.material(viewdirs=t_dirs, positions=positions, light_positions=None, **geo_out, **kwargs)
        comp_rgb_bg = self.background(dirs=rays_d)
        if bg_color is None:
            bg_color = comp_rgb_bg
        elif bg_color.shape[:-1] == (batch_size,):
            bg_color = bg_color.unsqueeze(1).unsqueeze(1).expand(-1, height, width, -1)
        if bg_color.shape[:-1] == (batch_size, height, width):
            bg_color = bg_color.reshape(n_rays, -1)
        weights_, opacity_, depth_, comp_rgb_fg, z_variance_, comp_rgb = nerfacc_api.composite(
            geo_out["density"][..., 0], rgb_fg_all, bg_color.float(), t_starts_, t_ends_, offset, count, 0
        )
        out = LazyOutputs({
            "comp_rgb": comp_rgb.view(batch_size, height, width, -1),
            "comp_rgb_fg": comp_rgb_fg.view(batch_size, height, width, -1),
            "comp_rgb_bg": comp_rgb_bg.view(batch_size, height, width, -1),
            "opacity": opacity_[..., None].view(batch_size, height, width, 1),
            "depth": depth_[..., None].view(batch_size, height, width, 1),
            "z_variance": z_variance_[..., None].view(batch_size, height, width, 1),
        })
        geo_keys = tuple(geo_out)

        def per_sample():                       # exact-length views, made (with ONE read of the count) when first asked for
            n = int(n_dev.item())
            if n == 0:                          # the reference's dummy sample of an empty batch (validate_empty_rays)
                z = t_starts_.new_zeros(1)
                o = {"weights": z[..., None], "t_points": z[..., None], "t_intervals": z[..., None], "t_dirs": rays_d.reshape(-1, 3)[:1],
                     "ray_indices": torch.zeros(1, dtype=torch.long, device=z.device), "points": positions[:1] * 0.0}
                o.update({k: geo_out[k][:1] * 0.0 for k in geo_keys})
                return o
            t0, t1 = t_starts_[:n, None], t_ends_[:n, None]
            o = {"weights": weights_[:n, None], "t_points": (t0 + t1) / 2.0, "t_intervals": t1 - t0, "t_dirs": t_dirs[:n],
                 "ray_indices": ray_indices[:n], "points": positions[:n]}
            o.update({k: geo_out[k][:n] for k in geo_keys})
            return o

        sample_keys = ("weights", "t_points", "t_intervals", "t_dirs", "ray_indices", "points") + geo_keys
        out.defer(sample_keys, per_sample)
        if self.material.requires_normal:
            def lazy_normal(geometry=self.geometry, pts=positions):
                n = int(n_dev.item())
                g = geometry(pts[:max(n, 1)], output_normal=True)
                return {"normal": g["normal"], "shading_normal": g["shading_normal"]}
            out.defer(("normal", "shading_normal"), lazy_normal)
        return out

    # ------------------------------------------------------------------------------------------
    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False) -> None:
        if self.cfg.estimator == "occgrid" and self.cfg.grid_prune:

            def occ_eval_fn(x):
                # approximates 1 - exp(-density * step) by its first-order term (reference :436-439)
                return self.geometry.forward_density(x) * self.render_step_size

            if self.training and not on_load_weights:
                self.estimator.update_every_n_steps(step=global_step, occ_eval_fn=occ_eval_fn)

    def update_step_end(self, epoch: int, global_step: int) -> None:
        pass

    def train(self, mode=True):
        self.randomized = mode and self.cfg.randomized
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()
