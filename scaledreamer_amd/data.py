"""`random-camera-datamodule`: the per-step random camera batch of the reference
(threestudio/data/uncond.py:28-344 RandomCameraIterableDataset.collate; ray generation utils/ops.py:183-269).
CPU work, <1 ms per step; the batch dict has the reference's keys, shapes and units (degrees for
elevation/azimuth)."""
from __future__ import annotations

import bisect
import math
import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple

import torch
import torch.nn.functional as F

from .base import Updateable
from .config import parse_structured
from .registry import register


def get_ray_directions(H: int, W: int, focal: float, use_pixel_centers: bool = True) -> torch.Tensor:
    pc = 0.5 if use_pixel_centers else 0
    cx, cy = W / 2, H / 2
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32) + pc, torch.arange(H, dtype=torch.float32) + pc, indexing="xy")
    return torch.stack([(i - cx) / focal, -(j - cy) / focal, -torch.ones_like(i)], -1)


def get_rays(directions: torch.Tensor, c2w: torch.Tensor, normalize: bool = True):
    """directions [B,H,W,3], c2w [B,4,4] -> rays_o, rays_d [B,H,W,3]   (ops.py:249-269, keepdim=True)"""
    rays_d = (directions[:, :, :, None, :] * c2w[:, None, None, :3, :3]).sum(-1)
    rays_o = c2w[:, None, None, :3, 3].expand(rays_d.shape)
    if normalize:
        rays_d = F.normalize(rays_d, dim=-1)
    return rays_o, rays_d


def get_projection_matrix(fovy: torch.Tensor, aspect_wh: float, near: float, far: float) -> torch.Tensor:
    b = fovy.shape[0]
    m = torch.zeros(b, 4, 4, dtype=torch.float32)
    m[:, 0, 0] = 1.0 / (torch.tan(fovy / 2.0) * aspect_wh)
    m[:, 1, 1] = -1.0 / torch.tan(fovy / 2.0)
    m[:, 2, 2] = -(far + near) / (far - near)
    m[:, 2, 3] = -2.0 * far * near / (far - near)
    m[:, 3, 2] = -1.0
    return m


def get_mvp_matrix(c2w: torch.Tensor, proj_mtx: torch.Tensor) -> torch.Tensor:
    w2c = torch.zeros(c2w.shape[0], 4, 4).to(c2w)
    w2c[:, :3, :3] = c2w[:, :3, :3].permute(0, 2, 1)
    w2c[:, :3, 3:] = -c2w[:, :3, :3].permute(0, 2, 1) @ c2w[:, :3, 3:]
    w2c[:, 3, 3] = 1.0
    return proj_mtx @ w2c


@dataclass
class RandomCameraDataModuleConfig:
    height: Any = 64
    width: Any = 64
    batch_size: Any = 1
    resolution_milestones: List[int] = field(default_factory=lambda: [])
    eval_height: int = 512
    eval_width: int = 512
    eval_batch_size: int = 1
    n_val_views: int = 1
    n_test_views: int = 120
    elevation_range: Tuple[float, float] = (-10, 90)
    azimuth_range: Tuple[float, float] = (-180, 180)
    camera_distance_range: Tuple[float, float] = (1, 1.5)
    fovy_range: Tuple[float, float] = (40, 70)
    camera_perturb: float = 0.1
    center_perturb: float = 0.2
    up_perturb: float = 0.02
    light_position_perturb: float = 1.0
    light_distance_range: Tuple[float, float] = (0.8, 1.5)
    eval_elevation_deg: float = 15.0
    eval_camera_distance: float = 1.5
    eval_fovy_deg: float = 70.0
    light_sample_strategy: str = "dreamfusion"
    batch_uniform_azimuth: bool = True
    progressive_until: int = 0
    rays_d_normalize: bool = True


@register("random-camera-datamodule")
class RandomCameraIterableDataset(Updateable):
    def __init__(self, cfg: Any) -> None:
        self.cfg: RandomCameraDataModuleConfig = parse_structured(RandomCameraDataModuleConfig, cfg)
        as_list = lambda v: [v] if isinstance(v, int) else list(v)
        self.heights, self.widths, self.batch_sizes = as_list(self.cfg.height), as_list(self.cfg.width), as_list(self.cfg.batch_size)
        assert len(self.heights) == len(self.widths) == len(self.batch_sizes)
        if len(self.heights) == 1:
            self.resolution_milestones = [-1]
        else:
            assert len(self.heights) == len(self.cfg.resolution_milestones) + 1
            self.resolution_milestones = [-1] + list(self.cfg.resolution_milestones)
        self.directions_unit_focals = [get_ray_directions(H=h, W=w, focal=1.0) for h, w in zip(self.heights, self.widths)]
        self.height, self.width, self.batch_size = self.heights[0], self.widths[0], self.batch_sizes[0]
        self.directions_unit_focal = self.directions_unit_focals[0]
        self.elevation_range = list(self.cfg.elevation_range)
        self.azimuth_range = list(self.cfg.azimuth_range)
        self.camera_distance_range = list(self.cfg.camera_distance_range)
        self.fovy_range = list(self.cfg.fovy_range)

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        i = bisect.bisect_right(self.resolution_milestones, global_step) - 1
        self.height, self.width, self.batch_size = self.heights[i], self.widths[i], self.batch_sizes[i]
        self.directions_unit_focal = self.directions_unit_focals[i]
        r = min(1.0, global_step / (self.cfg.progressive_until + 1))
        e = self.cfg.eval_elevation_deg
        self.elevation_range = [(1 - r) * e + r * self.cfg.elevation_range[0], (1 - r) * e + r * self.cfg.elevation_range[1]]
        self.azimuth_range = [r * self.cfg.azimuth_range[0], r * self.cfg.azimuth_range[1]]

    def __iter__(self):
        while True:
            yield {}

    def collate(self, batch=None) -> Dict[str, Any]:
        B, c = self.batch_size, self.cfg
        if random.random() < 0.5:  # uniform in elevation (biased towards the poles)
            elevation_deg = torch.rand(B) * (self.elevation_range[1] - self.elevation_range[0]) + self.elevation_range[0]
            elevation = elevation_deg * math.pi / 180
        else:  # uniform on the sphere
            lo, hi = (math.sin(v / 180.0 * math.pi) for v in self.elevation_range)
            elevation = torch.asin(torch.rand(B) * (hi - lo) + lo)
            elevation_deg = elevation / math.pi * 180.0
        if c.batch_uniform_azimuth:
            azimuth_deg = (torch.rand(B) + torch.arange(B)) / B * (self.azimuth_range[1] - self.azimuth_range[0]) + self.azimuth_range[0]
        else:
            azimuth_deg = torch.rand(B) * (self.azimuth_range[1] - self.azimuth_range[0]) + self.azimuth_range[0]
        azimuth = azimuth_deg * math.pi / 180
        camera_distances = torch.rand(B) * (self.camera_distance_range[1] - self.camera_distance_range[0]) + self.camera_distance_range[0]
        camera_positions = torch.stack([camera_distances * torch.cos(elevation) * torch.cos(azimuth),
                                        camera_distances * torch.cos(elevation) * torch.sin(azimuth),
                                        camera_distances * torch.sin(elevation)], dim=-1)
        center = torch.zeros_like(camera_positions)
        up = torch.as_tensor([0, 0, 1], dtype=torch.float32)[None, :].repeat(B, 1)
        camera_positions = camera_positions + (torch.rand(B, 3) * 2 * c.camera_perturb - c.camera_perturb)
        center = center + torch.randn(B, 3) * c.center_perturb
        up = up + torch.randn(B, 3) * c.up_perturb
        fovy_deg = torch.rand(B) * (self.fovy_range[1] - self.fovy_range[0]) + self.fovy_range[0]
        fovy = fovy_deg * math.pi / 180
        light_distances = torch.rand(B) * (c.light_distance_range[1] - c.light_distance_range[0]) + c.light_distance_range[0]
        if c.light_sample_strategy == "dreamfusion":
            light_direction = F.normalize(camera_positions + torch.randn(B, 3) * c.light_position_perturb, dim=-1)
            light_positions = light_direction * light_distances[:, None]
        elif c.light_sample_strategy == "magic3d":
            local_z = F.normalize(camera_positions, dim=-1)
            local_x = F.normalize(torch.stack([local_z[:, 1], -local_z[:, 0], torch.zeros_like(local_z[:, 0])], dim=-1), dim=-1)
            local_y = F.normalize(torch.cross(local_z, local_x, dim=-1), dim=-1)
            rot = torch.stack([local_x, local_y, local_z], dim=-1)
            la = torch.rand(B) * math.pi * 2 - math.pi
            le = torch.rand(B) * math.pi / 3 + math.pi / 6
            local = torch.stack([light_distances * torch.cos(le) * torch.cos(la), light_distances * torch.cos(le) * torch.sin(la),
                                 light_distances * torch.sin(le)], dim=-1)
            light_positions = (rot @ local[:, :, None])[:, :, 0]
        else:
            raise ValueError(f"Unknown light sample strategy: {c.light_sample_strategy}")
        lookat = F.normalize(center - camera_positions, dim=-1)
        right = F.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
        up = F.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
        c2w3x4 = torch.cat([torch.stack([right, up, -lookat], dim=-1), camera_positions[:, :, None]], dim=-1)
        c2w = torch.cat([c2w3x4, torch.zeros_like(c2w3x4[:, :1])], dim=1)
        c2w[:, 3, 3] = 1.0
        focal_length = 0.5 * self.height / torch.tan(0.5 * fovy)
        directions = self.directions_unit_focal[None].repeat(B, 1, 1, 1)
        directions[:, :, :, :2] = directions[:, :, :, :2] / focal_length[:, None, None, None]
        rays_o, rays_d = get_rays(directions, c2w, normalize=c.rays_d_normalize)
        proj_mtx = get_projection_matrix(fovy, self.width / self.height, 0.01, 100.0)
        return {"rays_o": rays_o, "rays_d": rays_d, "mvp_mtx": get_mvp_matrix(c2w, proj_mtx),
                "camera_positions": camera_positions, "c2w": c2w, "light_positions": light_positions,
                "elevation": elevation_deg, "azimuth": azimuth_deg, "camera_distances": camera_distances,
                "height": self.height, "width": self.width, "fovy": fovy, "proj_mtx": proj_mtx}


@dataclass
class RandomMultiviewCameraDataModuleConfig(RandomCameraDataModuleConfig):
    relative_radius: bool = True
    n_view: int = 1
    zoom_range: Tuple[float, float] = (1.0, 1.0)


@register("mvdream-random-multiview-camera-datamodule")
class RandomMultiviewCameraIterableDataset(RandomCameraIterableDataset):
    """threestudio/data/uncond_multiview.py:29-255: groups of `n_view` cameras sharing elevation / fovy / distance /
    perturbations / light, azimuths spread evenly over the range; distance relative to 1/tan(fovy/2).  The order of
    the RNG draws is the reference's (elevation, azimuth, fovy, distance, zoom, perturbs, light)."""

    def __init__(self, cfg: Any) -> None:
        cfg_mv = parse_structured(RandomMultiviewCameraDataModuleConfig, cfg)
        super().__init__({k: getattr(cfg_mv, k) for k in RandomCameraDataModuleConfig.__dataclass_fields__})
        self.cfg = cfg_mv
        self.zoom_range = list(self.cfg.zoom_range)

    def collate(self, batch=None) -> Dict[str, Any]:
        c, V = self.cfg, self.cfg.n_view
        assert self.batch_size % V == 0, f"batch_size ({self.batch_size}) must be dividable by n_view ({V})!"
        R, B = self.batch_size // V, self.batch_size
        rep = lambda v: v.repeat_interleave(V, dim=0)
        if random.random() < 0.5:
            elevation_deg = rep(torch.rand(R) * (self.elevation_range[1] - self.elevation_range[0]) + self.elevation_range[0])
            elevation = elevation_deg * math.pi / 180
        else:
            lo, hi = ((v + 90.0) / 180.0 for v in self.elevation_range)
            elevation = rep(torch.asin(2 * (torch.rand(R) * (hi - lo) + lo) - 1.0))
            elevation_deg = elevation / math.pi * 180.0
        azimuth_deg = (torch.rand(R).reshape(-1, 1) + torch.arange(V).reshape(1, -1)).reshape(-1) / V * (
            self.azimuth_range[1] - self.azimuth_range[0]) + self.azimuth_range[0]
        azimuth = azimuth_deg * math.pi / 180
        fovy_deg = rep(torch.rand(R) * (self.fovy_range[1] - self.fovy_range[0]) + self.fovy_range[0])
        fovy = fovy_deg * math.pi / 180
        camera_distances = rep(torch.rand(R) * (self.camera_distance_range[1] - self.camera_distance_range[0]) + self.camera_distance_range[0])
        if c.relative_radius:
            camera_distances = 1 / torch.tan(0.5 * fovy) * camera_distances
        zoom = rep(torch.rand(R) * (self.zoom_range[1] - self.zoom_range[0]) + self.zoom_range[0])
        fovy, fovy_deg = fovy * zoom, fovy_deg * zoom
        camera_positions = torch.stack([camera_distances * torch.cos(elevation) * torch.cos(azimuth),
                                        camera_distances * torch.cos(elevation) * torch.sin(azimuth),
                                        camera_distances * torch.sin(elevation)], dim=-1)
        center = torch.zeros_like(camera_positions)
        up = torch.as_tensor([0, 0, 1], dtype=torch.float32)[None, :].repeat(B, 1)
        camera_positions = camera_positions + rep(torch.rand(R, 3) * 2 * c.camera_perturb - c.camera_perturb)
        center = center + rep(torch.randn(R, 3) * c.center_perturb)
        up = up + rep(torch.randn(R, 3) * c.up_perturb)
        light_distances = rep(torch.rand(R) * (c.light_distance_range[1] - c.light_distance_range[0]) + c.light_distance_range[0])
        if c.light_sample_strategy == "dreamfusion":
            light_direction = F.normalize(camera_positions + rep(torch.randn(R, 3)) * c.light_position_perturb, dim=-1)
            light_positions = light_direction * light_distances[:, None]
        elif c.light_sample_strategy == "magic3d":
            local_z = F.normalize(camera_positions, dim=-1)
            local_x = F.normalize(torch.stack([local_z[:, 1], -local_z[:, 0], torch.zeros_like(local_z[:, 0])], dim=-1), dim=-1)
            local_y = F.normalize(torch.cross(local_z, local_x, dim=-1), dim=-1)
            rot = torch.stack([local_x, local_y, local_z], dim=-1)
            la = rep(torch.rand(R) * math.pi - 2 * math.pi)      # (sic) uncond_multiview.py:190-191
            le = rep(torch.rand(R) * math.pi / 3 + math.pi / 6)
            local = torch.stack([light_distances * torch.cos(le) * torch.cos(la), light_distances * torch.cos(le) * torch.sin(la),
                                 light_distances * torch.sin(le)], dim=-1)
            light_positions = (rot @ local[:, :, None])[:, :, 0]
        else:
            raise ValueError(f"Unknown light sample strategy: {c.light_sample_strategy}")
        lookat = F.normalize(center - camera_positions, dim=-1)
        right = F.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
        up = F.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
        c2w3x4 = torch.cat([torch.stack([right, up, -lookat], dim=-1), camera_positions[:, :, None]], dim=-1)
        c2w = torch.cat([c2w3x4, torch.zeros_like(c2w3x4[:, :1])], dim=1)
        c2w[:, 3, 3] = 1.0
        focal_length = 0.5 * self.height / torch.tan(0.5 * fovy)
        directions = self.directions_unit_focal[None].repeat(B, 1, 1, 1)
        directions[:, :, :, :2] = directions[:, :, :, :2] / focal_length[:, None, None, None]
        rays_o, rays_d = get_rays(directions, c2w, normalize=True)
        proj_mtx = get_projection_matrix(fovy, self.width / self.height, 0.1, 1000.0)
        return {"rays_o": rays_o, "rays_d": rays_d, "mvp_mtx": get_mvp_matrix(c2w, proj_mtx),
                "camera_positions": camera_positions, "c2w": c2w, "light_positions": light_positions,
                "elevation": elevation_deg, "azimuth": azimuth_deg, "camera_distances": camera_distances,
                "height": self.height, "width": self.width, "fovy": fovy_deg}
