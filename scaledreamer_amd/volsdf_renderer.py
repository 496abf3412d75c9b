"""`generative-space-volsdf-volume-renderer`
(custom/amortized/models/renderers/generative_space_volsdf_volume_renderer.py:36-462, built on NeuSVolumeRenderer,
threestudio/models/renderers/neus_volume_renderer.py:19-96) on the HIP path: importance-sampled VolSDF rendering of a
generator-conditioned SDF field.  Per step and ray: 128 proposal intervals (no-grad SDF pass) -> transmittance cdf -> 64
resampled edges -> 193 merged intervals -> field with finite-difference normals -> alpha compositing.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import partial
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nerfacc_api
from .estimators import ImportanceEstimator
from .registry import register
from .renderer import VolumeRenderer, chunk_batch, validate_empty_rays


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


def chunk_batch_custom(func, chunk_size: int, points: torch.Tensor, **kwargs):
    """custom/amortized/models/renderers/utils.py `chunk_batch`: split the POINT axis (dim 1) of [B, Np, 3]."""
    if chunk_size <= 0 or points.shape[1] <= chunk_size:
        return func(points, **kwargs)
    outs = [func(points[:, i:i + chunk_size], **kwargs) for i in range(0, points.shape[1], chunk_size)]
    B = points.shape[0]
    return {k: torch.cat([o[k].view(B, -1, o[k].shape[-1]) for o in outs], 1).reshape(-1, outs[0][k].shape[-1]) for k in outs[0]}


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
        if self.cfg.estimator == "importance":
            self.estimator = ImportanceEstimator()
        else:
            raise NotImplementedError(f"Estimator {self.cfg.estimator} not implemented for generative-space-volsdf-volume-renderer")
        self.chunk_training = self.cfg.train_chunk_size > 0
        self.cos_anneal_ratio = 1.0
        self.randomized = self.cfg.randomized

    def get_alpha(self, sdf, normal, dirs, dists):
        """neus_volume_renderer.py:93-117"""
        inv_std = self.variance(sdf)
        if self.cfg.use_volsdf:
            return torch.abs(dists.detach()) * volsdf_density(sdf, inv_std)
        true_cos = (dirs * normal).sum(-1, keepdim=True)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - self.cos_anneal_ratio) + F.relu(-true_cos) * self.cos_anneal_ratio)
        next_sdf, prev_sdf = sdf + iter_cos * dists * 0.5, sdf - iter_cos * dists * 0.5
        prev_cdf, next_cdf = torch.sigmoid(prev_sdf * inv_std), torch.sigmoid(next_sdf * inv_std)
        return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)

    def forward(self, rays_o, rays_d, light_positions, bg_color=None, noise=None, space_cache=None, text_embed=None, **kwargs):
        batch_size = rays_o.shape[0]
        bs_cache = text_embed.shape[0] if text_embed is not None else batch_size
        if space_cache is None:
            space_cache = self.geometry.generate_space_cache(styles=noise, text_embed=text_embed)
        if self.training:
            if bs_cache != batch_size:
                assert batch_size > bs_cache
                if torch.is_tensor(space_cache):
                    space_cache = space_cache.repeat_interleave(batch_size // bs_cache, dim=0)
                else:
                    raise NotImplementedError
            return self._forward(rays_o=rays_o, rays_d=rays_d, light_positions=light_positions, bg_color=bg_color, noise=noise,
                                 space_cache=space_cache, text_embed=text_embed, **kwargs)
        if bs_cache != batch_size:
            assert bs_cache == 1, "batch_size of space_cache must be 1 or equal to batch_size of rays_o"
            func = partial(self._forward, space_cache=space_cache, noise=noise, text_embed=text_embed)
            out = chunk_batch(func, 1, rays_o=rays_o, rays_d=rays_d, light_positions=light_positions, bg_color=bg_color, **kwargs)
            if "inv_std" in out:
                out["inv_std"] = out["inv_std"][0]
            return out
        return self._forward(rays_o=rays_o, rays_d=rays_d, light_positions=light_positions, bg_color=bg_color, noise=noise,
                             space_cache=space_cache, text_embed=text_embed, **kwargs)

    def _forward(self, rays_o, rays_d, light_positions, bg_color=None, noise=None, space_cache=None, text_embed=None, **kwargs):
        batch_size, height, width = rays_o.shape[:3]
        ro, rd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        lp = light_positions.reshape(-1, 1, 1, 3).expand(-1, height, width, -1).reshape(-1, 3)
        n_rays = ro.shape[0]
        if space_cache is None:
            assert noise is not None, "Either space_cache or noise must be provided"
            space_cache = self.geometry.generate_space_cache(styles=noise, text_embed=text_embed)
        if torch.is_tensor(space_cache):
            assert space_cache.shape[0] == batch_size, "space_cache must have the same batch size as rays_o"
        if self.cfg.estimator != "importance":
            raise NotImplementedError

        def prop_sigma_fn(t_starts, t_ends, proposal_network, space_cache):
            if torch.is_tensor(space_cache):
                B = space_cache.shape[0]
            elif isinstance(space_cache, dict):
                v = next(iter(space_cache.values()))
                B = v.shape[0] if torch.is_tensor(v) else v[0].shape[0]
            else:
                raise ValueError("space_cache must be a tensor or a dict")
            if not self.cfg.use_volsdf:
                raise ValueError("Currently only VolSDF supports importance sampling.")
            positions = ro.unsqueeze(-2) + rd.unsqueeze(-2) * (t_starts + t_ends)[..., None] / 2.0
            with torch.no_grad():
                if self.training and not self.chunk_training:
                    geo_out = self.geometry(positions.reshape(B, -1, 3), space_cache=space_cache, output_normal=False)
                else:
                    geo_out = chunk_batch_custom(partial(proposal_network, space_cache=space_cache, output_normal=False),
                                                 self.cfg.train_chunk_size if self.training else self.cfg.eval_chunk_size,
                                                 positions.reshape(B, -1, 3))
                inv_std = self.variance(geo_out["sdf"])
                return volsdf_density(geo_out["sdf"], inv_std).reshape(positions.shape[:2])

        t_starts_, t_ends_ = self.estimator.sampling(
            prop_sigma_fns=[partial(prop_sigma_fn, proposal_network=self.geometry, space_cache=space_cache)],
            prop_samples=[self.cfg.num_samples_per_ray_importance], num_samples=self.cfg.num_samples_per_ray, n_rays=n_rays,
            near_plane=self.cfg.near_plane, far_plane=self.cfg.far_plane, sampling_type="uniform", stratified=self.randomized)
        per_ray = t_starts_.shape[1]
        ray_indices = torch.arange(n_rays, device=ro.device).unsqueeze(-1).expand(-1, per_ray).flatten()
        t_starts_, t_ends_ = t_starts_.flatten(), t_ends_.flatten()
        ray_indices, t_starts_, t_ends_ = validate_empty_rays(ray_indices, t_starts_, t_ends_)
        ray_indices = ray_indices.long()
        t_starts, t_ends = t_starts_[..., None], t_ends_[..., None]
        t_origins, t_dirs, t_light = ro[ray_indices], rd[ray_indices], lp[ray_indices]
        t_positions = (t_starts + t_ends) / 2.0
        positions = t_origins + t_dirs * t_positions
        t_intervals = t_ends - t_starts

        hyper_bg = hasattr(self.background, "enabling_hypernet") and self.background.enabling_hypernet
        if self.training and not self.chunk_training:
            geo_out = self.geometry(positions.reshape(batch_size, -1, 3), space_cache=space_cache, output_normal=True)
            rgb_fg_all = self.material(viewdirs=t_dirs, positions=positions, light_positions=t_light, **geo_out, **kwargs)
            comp_rgb_bg = self.background(dirs=rays_d, text_embed=text_embed) if hyper_bg else self.background(dirs=rays_d)
        else:
            geo_out = chunk_batch_custom(partial(self.geometry, space_cache=space_cache, output_normal=True),
                                         self.cfg.train_chunk_size if self.training else self.cfg.eval_chunk_size,
                                         positions.reshape(batch_size, -1, 3))
            rgb_fg_all = chunk_batch(self.material, self.cfg.eval_chunk_size, viewdirs=t_dirs, positions=positions,
                                     light_positions=t_light, **geo_out)
            comp_rgb_bg = self.background(dirs=rays_d, text_embed=text_embed) if hyper_bg else self.background(dirs=rays_d)

        alpha = self.get_alpha(geo_out["sdf"], geo_out["normal"], t_dirs, t_intervals)
        # every ray holds exactly `per_ray` samples: packed_info is known without a bincount
        packed = torch.stack([torch.arange(n_rays, device=ro.device, dtype=torch.int32) * per_ray,
                              torch.full((n_rays,), per_ray, device=ro.device, dtype=torch.int32)], dim=-1) \
            if ray_indices.shape[0] == n_rays * per_ray else None
        weights_, _ = nerfacc_api.render_weight_from_alpha(alpha[..., 0], packed_info=packed, ray_indices=ray_indices, n_rays=n_rays)
        weights = weights_[..., None]
        acc = partial(nerfacc_api.accumulate_along_rays, weights[..., 0], ray_indices=ray_indices, n_rays=n_rays)
        opacity = acc(values=None)
        depth = acc(values=t_positions)
        comp_rgb_fg = acc(values=rgb_fg_all)
        z_variance = acc(values=(t_positions - depth[ray_indices]) ** 2)
        if bg_color is None:
            bg_color = comp_rgb_bg
        if bg_color.shape[:-1] == (batch_size, height, width):
            bg_color = bg_color.reshape(batch_size * height * width, -1)
        comp_rgb = comp_rgb_fg + bg_color * (1.0 - opacity)
        out = {
            "comp_rgb": comp_rgb.view(batch_size, height, width, -1),
            "comp_rgb_fg": comp_rgb_fg.view(batch_size, height, width, -1),
            "comp_rgb_bg": comp_rgb_bg.view(batch_size, height, width, -1),
            "opacity": opacity.view(batch_size, height, width, 1),
            "depth": depth.view(batch_size, height, width, 1),
            "z_variance": z_variance.view(batch_size, height, width, 1),
        }
        if "normal" in geo_out:
            comp_normal = F.normalize(acc(values=geo_out["normal"]), dim=-1)
            comp_normal = torch.lerp(torch.zeros_like(comp_normal), (comp_normal.detach() + 1.0) / 2.0, opacity)
            out["comp_normal"] = comp_normal.view(batch_size, height, width, 3)
        if self.training:
            out.update({"weights": weights, "t_points": t_positions, "t_intervals": t_intervals, "t_dirs": t_dirs,
                        "ray_indices": ray_indices, "points": positions, **geo_out})
            out["inv_std"] = self.variance.inv_std
        return out

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
