"""`generative-space-volsdf-volume-renderer`
(custom/amortized/models/renderers/generative_space_volsdf_volume_renderer.py:36-462, built on NeuSVolumeRenderer,
threestudio/models/renderers/neus_volume_renderer.py:19-96) on the HIP path: importance-sampled VolSDF rendering of a
generator-conditioned SDF field.  Per step and ray: 128 proposal intervals (no-grad SDF pass) -> transmittance cdf -> 64
resampled edges -> 193 merged intervals -> field with finite-difference normals -> alpha compositing.

Built as a pipeline of four stages on dense [n_rays, S] interval tensors (every ray holds exactly S samples, so the packed layout
is offset = ray * S, count = S — no bincount, no masks, no dummy samples):
    _cache_per_view   generator output per rendered view          (reference :89-130: cache creation / repeat per view)
    _intervals        importance-sampled (t_start, t_end)         (:205-278 prop_sigma_fn + estimator.sampling)
    _shade            field / material / background at the samples (:280-355)
    _composite        ONE fused alpha-compositing pass             (:357-430: render_weight_from_alpha + 4 accumulate_along_rays +
                      (asd_composite_*, mode 2) and a no-grad      comp_normal): weights, opacity, depth, foreground, z-variance
                      second pass for the normal image            and the background blend come out of a single kernel
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nerfacc_api
from .estimators import ImportanceEstimator
from .registry import register
from .renderer import VolumeRenderer


def volsdf_density(sdf: torch.Tensor, inv_std: torch.Tensor) -> torch.Tensor:
    """neus_volume_renderer.py:19-23"""
    inv_std = inv_std.clamp(0.0, 80.0)
    beta = 1 / inv_std
    return inv_std * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class LearnedVariance(nn.Module):
    def __init__(self, init_val, requires_grad=True):
        super().__init__()
        self.register_parameter("_inv_std", nn.Parameter(torch.tensor(init_val), requires_grad=requires_grad))

    @property
    def inv_std(self):
        return torch.exp(self._inv_std * 10.0)

    def forward(self, x):
        return torch.ones_like(x) * self.inv_std.clamp(1.0e-6, 1.0e6)


def _cache_batch(space_cache) -> int:
    """leading (prompt / view) dimension of a generator output: a tensor, or a dict of tensors / lists of tensors"""
    if torch.is_tensor(space_cache):
        return space_cache.shape[0]
    if isinstance(space_cache, dict):
        v = next(iter(space_cache.values()))
        return v.shape[0] if torch.is_tensor(v) else v[0].shape[0]
    raise ValueError("space_cache must be a tensor or a dict")


def _field_in_chunks(field, points: torch.Tensor, chunk: int, **kw) -> Dict[str, torch.Tensor]:
    """field(points[B, Np, 3]) evaluated in slices of the POINT axis (custom/amortized/models/renderers/utils.py: the batch axis
    belongs to the space cache); outputs are re-assembled in the field's own [B * Np, C] row order"""
    B, Np = points.shape[:2]
    if chunk <= 0 or Np <= chunk:
        return field(points, **kw)
    parts = [field(points[:, i:i + chunk], **kw) for i in range(0, Np, chunk)]
    return {k: torch.cat([p[k].view(B, -1, p[k].shape[-1]) for p in parts], dim=1).reshape(B * Np, -1) for k in parts[0]}


@register("generative-space-volsdf-volume-renderer")
class GenerativeSpaceVolSDFVolumeRenderer(VolumeRenderer):
    @dataclass
    class Config(VolumeRenderer.Config):
        num_samples_per_ray: int = 512
        randomized: bool = True
        eval_chunk_size: int = 320000
        learned_variance_init: float = 0.3
        cos_anneal_end_steps: int = 0
        use_volsdf: bool = False
        near_plane: float = 0.0
        far_plane: float = 1e10
        trainable_variance: bool = True
        estimator: str = "occgrid"
        grid_prune: bool = True
        prune_alpha_threshold: bool = True
        num_samples_per_ray_importance: int = 64
        train_chunk_size: int = 0

    cfg: Config

    def configure(self, geometry, material, background) -> None:
        super().configure(geometry, material, background)
        self.variance = LearnedVariance(self.cfg.learned_variance_init, requires_grad=self.cfg.trainable_variance)
        if self.cfg.estimator != "importance":     # the reference builds nothing else for this renderer (:76-78)
            raise NotImplementedError(f"Estimator {self.cfg.estimator} not implemented for generative-space-volsdf-volume-renderer")
        self.estimator = ImportanceEstimator()
        self.cos_anneal_ratio = 1.0
        self.randomized = self.cfg.randomized

    # ---- alpha models ---------------------------------------------------------------------------------------------------------------
    def get_alpha(self, sdf, normal, dirs, dists):
        """VolSDF: alpha = |dt| * sigma(sdf) (neus_volume_renderer.py:93-96); NeuS: the discrete opacity of the logistic cdf along
        the ray with the annealed cosine (:97-117)"""
        inv_std = self.variance(sdf)
        if self.cfg.use_volsdf:
            return dists.detach().abs() * volsdf_density(sdf, inv_std)
        cos = (dirs * normal).sum(-1, keepdim=True)
        k = self.cos_anneal_ratio
        slope = -(F.relu(0.5 - 0.5 * cos) * (1.0 - k) + F.relu(-cos) * k)
        half = slope * dists * 0.5
        cdf_in, cdf_out = torch.sigmoid((sdf - half) * inv_std), torch.sigmoid((sdf + half) * inv_std)
        return ((cdf_in - cdf_out + 1e-5) / (cdf_in + 1e-5)).clip(0.0, 1.0)

    def _chunk(self) -> int:
        """point-axis chunk of the field calls: none while training unless train_chunk_size asks for it"""
        return self.cfg.train_chunk_size if self.training else self.cfg.eval_chunk_size

    # ---- stage 1 ----------------------------------------------------------------------------------------------------------------------
    def _cache_per_view(self, space_cache, noise, text_embed, n_views: int):
        if space_cache is None:      # (generators that ignore the noise — the hypernetwork — are called with noise = None)
            space_cache = self.geometry.generate_space_cache(styles=noise, text_embed=text_embed)
        n_cache = _cache_batch(space_cache)
        if n_cache != n_views and self.training:       # several views of one prompt share its cache (4-view groups)
            if n_views < n_cache or n_views % n_cache or not torch.is_tensor(space_cache):
                raise NotImplementedError("per-view repetition is defined for tensor caches with n_views a multiple of the prompts")
            space_cache = space_cache.repeat_interleave(n_views // n_cache, dim=0)
        return space_cache

    # ---- stage 2 ----------------------------------------------------------------------------------------------------------------------
    def _intervals(self, ro, rd, space_cache) -> Tuple[torch.Tensor, torch.Tensor]:
        if not self.cfg.use_volsdf:
            raise ValueError("Currently only VolSDF supports importance sampling.")
        B = _cache_batch(space_cache)

        def proposal_density(t0, t1):      # [n_rays, S] interval edges -> sigma at the mid-points, no gradient
            mid = ro[:, None, :] + rd[:, None, :] * ((t0 + t1) * 0.5)[..., None]
            with torch.no_grad():
                fused_sdf = getattr(self.geometry, "_use_fused", None)
                if fused_sdf is not None and fused_sdf(mid):
                    # the reference evaluates the whole field here and keeps `sdf` (generative_space_volsdf_volume_renderer.py:233-252); the fused
                    # fields have an sdf-only entry (same value: the sdf head does not depend on the feature head)
                    sdf = self.geometry.forward_sdf(mid.reshape(B, -1, 3), space_cache).reshape(-1, 1)
                else:
                    sdf = _field_in_chunks(self.geometry, mid.reshape(B, -1, 3), self._chunk(), space_cache=space_cache, output_normal=False)["sdf"]
                return volsdf_density(sdf, self.variance(sdf)).reshape(t0.shape)

        return self.estimator.sampling(prop_sigma_fns=[proposal_density], prop_samples=[self.cfg.num_samples_per_ray_importance],
                                       num_samples=self.cfg.num_samples_per_ray, n_rays=ro.shape[0], near_plane=self.cfg.near_plane,
                                       far_plane=self.cfg.far_plane, sampling_type="uniform", stratified=self.randomized)

    # ---- stage 3 ----------------------------------------------------------------------------------------------------------------------
    def _shade(self, positions, t_dirs, t_light, rays_d, space_cache, text_embed, n_views: int, extra: Dict[str, Any]):
        geo = _field_in_chunks(self.geometry, positions.reshape(n_views, -1, 3), self._chunk(), space_cache=space_cache, output_normal=True)
        if self.training:
            rgb = self.material(viewdirs=t_dirs, positions=positions, light_positions=t_light, **geo, **extra)
        else:
            n, c = positions.shape[0], max(1, self.cfg.eval_chunk_size)
            rgb = torch.cat([self.material(viewdirs=t_dirs[i:i + c], positions=positions[i:i + c], light_positions=t_light[i:i + c],
                                           **{k: v[i:i + c] for k, v in geo.items()}) for i in range(0, n, c)], dim=0)
        hyper_bg = getattr(self.background, "enabling_hypernet", False)
        bg = self.background(dirs=rays_d, text_embed=text_embed) if hyper_bg else self.background(dirs=rays_d)
        return geo, rgb, bg

    # ---- stage 4 ----------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _composite(alpha, rgb, bg_flat, t0, t1, normal, n_rays: int, per_ray: int):
        dev = alpha.device
        offset = torch.arange(n_rays, device=dev, dtype=torch.int32) * per_ray
        count = torch.full((n_rays,), per_ray, device=dev, dtype=torch.int32)
        weights, opacity, depth, fg, z_var, comp = nerfacc_api.composite(alpha, rgb, bg_flat, t0, t1, offset, count, 2)
        comp_normal = None
        if normal is not None:       # the normal image takes no gradient through the normals (reference detaches them): a second,
            with torch.no_grad():    # forward-only pass of the same kernel accumulates sum_i w_i n_i
                acc_n = nerfacc_api.composite(alpha.detach(), normal.detach().contiguous(), torch.zeros_like(bg_flat), t0, t1, offset, count, 2)[3]
                unit = (F.normalize(acc_n, dim=-1) + 1.0) * 0.5
            comp_normal = unit * opacity[:, None]                  # lerp(0, (n + 1) / 2, opacity): differentiable in the opacity only
        return weights, opacity, depth, fg, z_var, comp, comp_normal

    # ---- one set of views with one cache entry per view -------------------------------------------------------------------------------
    def _render(self, rays_o, rays_d, light_positions, bg_color, space_cache, text_embed, extra) -> Dict[str, torch.Tensor]:
        V, H, W = rays_o.shape[:3]
        if torch.is_tensor(space_cache) and space_cache.shape[0] != V:
            raise AssertionError("space_cache must have the same batch size as rays_o")
        ro, rd = rays_o.reshape(-1, 3).contiguous().float(), rays_d.reshape(-1, 3).contiguous().float()
        n_rays = ro.shape[0]
        t0, t1 = self._intervals(ro, rd, space_cache)                     # [n_rays, S]
        S = t0.shape[1]
        ray_indices = torch.arange(n_rays, device=ro.device).repeat_interleave(S)
        t0f, t1f = t0.reshape(-1).contiguous(), t1.reshape(-1).contiguous()
        t_mid, t_len = ((t0f + t1f) * 0.5)[:, None], (t1f - t0f)[:, None]
        t_dirs = rd[ray_indices]
        positions = ro[ray_indices] + t_dirs * t_mid
        t_light = light_positions.reshape(-1, 1, 3).expand(-1, H * W * S, -1).reshape(-1, 3)
        geo, rgb, bg = self._shade(positions, t_dirs, t_light, rays_d, space_cache, text_embed, V, extra)
        alpha = self.get_alpha(geo["sdf"], geo["normal"], t_dirs, t_len)[:, 0]
        blend = bg if bg_color is None else bg_color
        if blend.dim() == 2 and blend.shape[0] == V:                      # one colour per view
            blend = blend[:, None, None, :].expand(-1, H, W, -1)
        weights, opacity, depth, fg, z_var, comp, comp_normal = self._composite(
            alpha, rgb, blend.reshape(n_rays, -1).float(), t0f, t1f, geo.get("normal"), n_rays, S)
        img = lambda x, c: x.reshape(V, H, W, c)
        out = {"comp_rgb": img(comp, 3), "comp_rgb_fg": img(fg, 3), "comp_rgb_bg": img(bg, 3), "opacity": img(opacity, 1),
               "depth": img(depth, 1), "z_variance": img(z_var, 1)}
        if comp_normal is not None:
            out["comp_normal"] = img(comp_normal, 3)
        if self.training:
            out.update(weights=weights[:, None], t_points=t_mid, t_intervals=t_len, t_dirs=t_dirs, ray_indices=ray_indices, points=positions, **geo)
            out["inv_std"] = self.variance.inv_std
        return out

    def forward(self, rays_o, rays_d, light_positions, bg_color=None, noise=None, space_cache=None, text_embed=None, **kwargs):
        V = rays_o.shape[0]
        cache = self._cache_per_view(space_cache, noise, text_embed, V)
        if self.training or _cache_batch(cache) == V:
            return self._render(rays_o, rays_d, light_positions, bg_color, cache, text_embed, kwargs)
        # evaluation of many views of ONE prompt (orbit videos): view by view against the same cache entry
        if _cache_batch(cache) != 1:
            raise AssertionError("batch_size of space_cache must be 1 or equal to batch_size of rays_o")
        views = [self._render(rays_o[v:v + 1], rays_d[v:v + 1], light_positions[v:v + 1], None if bg_color is None else bg_color[v:v + 1],
                              cache, text_embed, kwargs) for v in range(V)]
        return {k: torch.cat([o[k] for o in views], dim=0) for k in views[0]}

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False) -> None:
        pass

    def train(self, mode=True):
        self.randomized = mode and self.cfg.randomized
        if hasattr(self.geometry, "train"):
            self.geometry.train(mode)
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        if hasattr(self.geometry, "eval"):
            self.geometry.eval()
        return super().train(False)
