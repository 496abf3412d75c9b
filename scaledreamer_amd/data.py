"""`random-camera-datamodule` / `mvdream-random-multiview-camera-datamodule`: the per-step random camera batch of the reference
(threestudio/data/uncond.py:28-344 RandomCameraIterableDataset.collate, threestudio/data/uncond_multiview.py:29-255; ray
generation threestudio/utils/ops.py:183-269).

Built in three separate stages instead of the reference's ~60 CPU tensor ops per step:
  1. DRAW    every random number of the step, in the reference's order (that order is the contract: seeded batches are pinned key
             by key by tests/golden/camera_*.npz).  A draw plan — a list of (name, kind, width) — describes the stream; the
             single-view and the multi-view sampler differ only in their plans.
  2. CAMERAS elevation / azimuth / distance / look-at frame / projection for the B <= 8 cameras of a step: a few dozen scalars,
             computed on the host in float64 numpy and emitted as the reference's float32 tensors (degrees for elevation / azimuth).
  3. RAYS    rays_o / rays_d [B,H,W,3] are generated ON THE DEVICE from c2w and the focal lengths (asd_generate_rays): no CPU ray
             tensors, no H2D copy of 2 x B*H*W*3 floats per step.  Without a GPU `collate()` raises like every HIP op; `cameras()`
             (stages 1-2) is plain host logic.
The batch dict has the reference's keys, shapes and units."""
from __future__ import annotations

import bisect
import math
import random
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .base import Updateable
from .config import parse_structured
from .registry import register


def _unit(v: np.ndarray) -> np.ndarray:
    """F.normalize(dim=-1): v / max(||v||, 1e-12)"""
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-12)


def _spherical(radius, elevation, azimuth) -> np.ndarray:
    """z-up: +x at azimuth 0 (uncond.py:193-204)"""
    return np.stack([radius * np.cos(elevation) * np.cos(azimuth), radius * np.cos(elevation) * np.sin(azimuth), radius * np.sin(elevation)], -1)


def look_at(positions: np.ndarray, center: np.ndarray, up: np.ndarray) -> np.ndarray:
    """camera-to-world [B,4,4] of an OpenGL camera (-z forward) at `positions` looking at `center` (uncond.py:297-312)"""
    fwd = _unit(center - positions)
    right = _unit(np.cross(fwd, up))
    true_up = _unit(np.cross(right, fwd))
    c2w = np.zeros((positions.shape[0], 4, 4))
    c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3], c2w[:, 3, 3] = right, true_up, -fwd, positions, 1.0
    return c2w


def projection(fovy: np.ndarray, aspect_wh: float, near: float, far: float) -> np.ndarray:
    """OpenGL perspective matrices with the reference's flipped y (ops.py:222-236)"""
    m = np.zeros((fovy.shape[0], 4, 4))
    m[:, 0, 0] = 1.0 / (np.tan(fovy / 2.0) * aspect_wh)
    m[:, 1, 1] = -1.0 / np.tan(fovy / 2.0)
    m[:, 2, 2] = -(far + near) / (far - near)
    m[:, 2, 3] = -2.0 * far * near / (far - near)
    m[:, 3, 2] = -1.0
    return m


def mvp(c2w: np.ndarray, proj: np.ndarray) -> np.ndarray:
    """proj @ inverse(c2w) with the rigid inverse written out (ops.py:239-246)"""
    w2c = np.zeros_like(c2w)
    rt = np.transpose(c2w[:, :3, :3], (0, 2, 1))
    w2c[:, :3, :3], w2c[:, :3, 3], w2c[:, 3, 3] = rt, -(rt @ c2w[:, :3, 3:])[..., 0], 1.0
    return proj @ w2c


def rays_from_cameras(c2w: torch.Tensor, focal: torch.Tensor, H: int, W: int, normalize: bool = True, device=None):
    """device rays of the cameras (asd_generate_rays); c2w / focal may live on the host (two tiny uploads)"""
    from . import ops

    dev = torch.device(device) if device is not None else (c2w.device if c2w.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    return ops.generate_rays(c2w.to(dev, non_blocking=True), focal.to(dev, non_blocking=True), H, W, normalize)


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


@dataclass
class RandomMultiviewCameraDataModuleConfig(RandomCameraDataModuleConfig):
    relative_radius: bool = True
    n_view: int = 1
    zoom_range: Tuple[float, float] = (1.0, 1.0)


@register("random-camera-datamodule")
class RandomCameraIterableDataset(Updateable):
    NEAR_FAR = (0.01, 100.0)
    CONFIG = RandomCameraDataModuleConfig

    def __init__(self, cfg: Any) -> None:
        self.cfg = parse_structured(self.CONFIG, cfg)
        as_list = lambda v: [v] if isinstance(v, int) else list(v)
        self.heights, self.widths, self.batch_sizes = as_list(self.cfg.height), as_list(self.cfg.width), as_list(self.cfg.batch_size)
        assert len(self.heights) == len(self.widths) == len(self.batch_sizes)
        if len(self.heights) == 1:
            self.resolution_milestones = [-1]
        else:
            assert len(self.heights) == len(self.cfg.resolution_milestones) + 1
            self.resolution_milestones = [-1] + list(self.cfg.resolution_milestones)
        self.height, self.width, self.batch_size = self.heights[0], self.widths[0], self.batch_sizes[0]
        self.elevation_range = list(self.cfg.elevation_range)
        self.azimuth_range = list(self.cfg.azimuth_range)
        self.camera_distance_range = list(self.cfg.camera_distance_range)
        self.fovy_range = list(self.cfg.fovy_range)
        self.ray_device: Optional[torch.device] = None      # None: the current CUDA device

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        i = bisect.bisect_right(self.resolution_milestones, global_step) - 1     # resolution milestones (uncond.py:106-116)
        self.height, self.width, self.batch_size = self.heights[i], self.widths[i], self.batch_sizes[i]
        r = min(1.0, global_step / (self.cfg.progressive_until + 1))             # progressive view range (:118-141)
        e = self.cfg.eval_elevation_deg
        self.elevation_range = [(1 - r) * e + r * self.cfg.elevation_range[0], (1 - r) * e + r * self.cfg.elevation_range[1]]
        self.azimuth_range = [r * self.cfg.azimuth_range[0], r * self.cfg.azimuth_range[1]]

    def __iter__(self):
        while True:
            yield {}

    # ---- stage 1: the random stream ---------------------------------------------------------------------------------------------
    def groups(self) -> Tuple[int, int]:
        """(independently drawn cameras, views sharing each draw)"""
        return self.batch_size, 1

    def draw_plan(self) -> List[Tuple[str, str, int]]:
        """(name, 'u' uniform [0,1) | 'n' standard normal, width) in the order uncond.py:143-290 consumes torch's CPU generator"""
        light = [("light_dir", "n", 3)] if self.cfg.light_sample_strategy == "dreamfusion" else [("light_azimuth", "u", 1), ("light_elevation", "u", 1)]
        return [("elevation", "u", 1), ("azimuth", "u", 1), ("distance", "u", 1), ("position_jitter", "u", 3), ("center_jitter", "n", 3),
                ("up_jitter", "n", 3), ("fovy", "u", 1), ("light_distance", "u", 1)] + light

    def draw(self) -> Dict[str, np.ndarray]:
        if self.cfg.light_sample_strategy not in ("dreamfusion", "magic3d"):
            raise ValueError(f"Unknown light sample strategy: {self.cfg.light_sample_strategy}")
        R, _ = self.groups()
        out: Dict[str, Any] = {"coin": random.random()}          # python's generator first (uncond.py:147), then torch's
        for name, kind, width in self.draw_plan():
            shape = (R,) if width == 1 else (R, width)
            out[name] = (torch.rand(shape) if kind == "u" else torch.randn(shape)).numpy()          # float32, as drawn
        return out

    # ---- stage 2: the cameras ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _range(u, lo_hi):
        """u * (hi - lo) + lo in float32 like the reference: where the result cancels (azimuth near 0 of a +-180 range) the float32
        rounding of this step is visible in the batch, so it is part of the contract; everything downstream runs in float64"""
        return np.float32(u) * np.float32(lo_hi[1] - lo_hi[0]) + np.float32(lo_hi[0])

    def _elevation_deg(self, d) -> np.ndarray:
        if d["coin"] < 0.5:       # uniform in the angle (biased towards the poles)
            return self._range(d["elevation"], self.elevation_range)
        lo, hi = (math.sin(v / 180.0 * math.pi) for v in self.elevation_range)       # uniform on the sphere
        return np.degrees(np.arcsin(self._range(d["elevation"], (lo, hi))))

    def _azimuth_deg(self, d, R: int, V: int) -> np.ndarray:
        u = d["azimuth"]
        if self.cfg.batch_uniform_azimuth:            # stratified over the batch (uncond.py:176-186)
            u = (u + np.arange(R, dtype=np.float32)) / np.float32(R)
        return self._range(u, self.azimuth_range)

    def _distance_fovy(self, d):
        fovy_deg = self._range(d["fovy"], self.fovy_range)
        return self._range(d["distance"], self.camera_distance_range), fovy_deg, fovy_deg

    def cameras(self) -> Dict[str, Any]:
        """stages 1-2: every key of the reference batch except the rays (host tensors, float32)"""
        c = self.cfg
        R, V = self.groups()
        d = self.draw()
        spread = lambda a: np.repeat(a, V, axis=0)       # the views of a group share the draw
        elevation_deg = spread(self._elevation_deg(d))
        azimuth_deg = self._azimuth_deg(d, R, V)
        distance, fovy_deg, fovy_out_deg = (spread(v) for v in self._distance_fovy(d))
        elevation_deg, azimuth_deg, distance, fovy_deg = (np.asarray(v, np.float64) for v in (elevation_deg, azimuth_deg, distance, fovy_deg))
        elevation, azimuth, fovy = np.radians(elevation_deg), np.radians(azimuth_deg), np.radians(fovy_deg)
        d = {k: (np.asarray(v, np.float64) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        positions = _spherical(distance, elevation, azimuth) + spread(d["position_jitter"] * 2 * c.camera_perturb - c.camera_perturb)
        center = spread(d["center_jitter"] * c.center_perturb)
        up = np.array([0.0, 0.0, 1.0])[None] + spread(d["up_jitter"] * c.up_perturb)
        light_distance = spread(self._range(d["light_distance"], c.light_distance_range).astype(np.float64))
        if c.light_sample_strategy == "dreamfusion":      # a point light near the camera direction
            light = _unit(positions + spread(d["light_dir"]) * c.light_position_perturb) * light_distance[:, None]
        else:                                             # magic3d: a light in the camera's local upper hemisphere
            z = _unit(positions)
            x = _unit(np.stack([z[:, 1], -z[:, 0], np.zeros_like(z[:, 0])], -1))
            y = _unit(np.cross(z, x))
            la, le = (spread(a) for a in self._light_angles(d))
            local = _spherical(light_distance, le, la)
            light = x * local[:, :1] + y * local[:, 1:2] + z * local[:, 2:3]
        c2w = look_at(positions, center, up)
        proj = projection(fovy, self.width / self.height, *self.NEAR_FAR)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        out = {"mvp_mtx": f32(mvp(c2w, proj)), "camera_positions": f32(positions), "c2w": f32(c2w), "light_positions": f32(light),
               "elevation": f32(elevation_deg), "azimuth": f32(azimuth_deg), "camera_distances": f32(distance),
               "height": self.height, "width": self.width, "focal_length": f32(0.5 * self.height / np.tan(0.5 * fovy))}
        out.update(self._fovy_keys(f32, fovy, fovy_out_deg, proj))
        return out

    def _light_angles(self, d):
        return d["light_azimuth"] * math.pi * 2 - math.pi, d["light_elevation"] * math.pi / 3 + math.pi / 6

    def _fovy_keys(self, f32, fovy, fovy_deg, proj):
        return {"fovy": f32(fovy), "proj_mtx": f32(proj)}             # radians (uncond.py:341)

    # ---- stage 3: rays on the device --------------------------------------------------------------------------------------------
    def collate(self, batch=None) -> Dict[str, Any]:
        out = self.cameras()
        out["rays_o"], out["rays_d"] = rays_from_cameras(out["c2w"], out.pop("focal_length"), self.height, self.width,
                                                         self.cfg.rays_d_normalize, device=self.ray_device)
        return out


@register("mvdream-random-multiview-camera-datamodule")
class RandomMultiviewCameraIterableDataset(RandomCameraIterableDataset):
    """threestudio/data/uncond_multiview.py:29-255: groups of `n_view` cameras share every draw (elevation, fovy, distance, zoom,
    perturbations, light); the azimuths of a group are spread evenly over the range from one draw; the distance is relative to
    1 / tan(fovy / 2); the batch reports fovy in degrees."""
    NEAR_FAR = (0.1, 1000.0)
    CONFIG = RandomMultiviewCameraDataModuleConfig

    def __init__(self, cfg: Any) -> None:
        super().__init__(cfg)
        self.zoom_range = list(self.cfg.zoom_range)

    def groups(self) -> Tuple[int, int]:
        V = self.cfg.n_view
        assert self.batch_size % V == 0, f"batch_size ({self.batch_size}) must be dividable by n_view ({V})!"
        return self.batch_size // V, V

    def draw_plan(self):
        light = [("light_dir", "n", 3)] if self.cfg.light_sample_strategy == "dreamfusion" else [("light_azimuth", "u", 1), ("light_elevation", "u", 1)]
        return [("elevation", "u", 1), ("azimuth", "u", 1), ("fovy", "u", 1), ("distance", "u", 1), ("zoom", "u", 1), ("position_jitter", "u", 3),
                ("center_jitter", "n", 3), ("up_jitter", "n", 3), ("light_distance", "u", 1)] + light

    def _elevation_deg(self, d) -> np.ndarray:
        if d["coin"] < 0.5:
            return self._range(d["elevation"], self.elevation_range)
        lo, hi = ((v + 90.0) / 180.0 for v in self.elevation_range)      # uniform on the sphere, parameterised on [0,1] (uncond_multiview.py:67-79)
        return np.degrees(np.arcsin(2 * self._range(d["elevation"], (lo, hi)) - 1.0))

    def _azimuth_deg(self, d, R: int, V: int) -> np.ndarray:
        return self._range((d["azimuth"][:, None] + np.arange(V, dtype=np.float32)[None]).reshape(-1) / np.float32(V), self.azimuth_range)

    def _distance_fovy(self, d):
        fovy_deg = self._range(d["fovy"], self.fovy_range)
        distance = self._range(d["distance"], self.camera_distance_range)
        if self.cfg.relative_radius:
            distance = distance / np.tan(0.5 * np.radians(fovy_deg.astype(np.float64)))
        zoom = self._range(d["zoom"], self.zoom_range)
        return distance, fovy_deg * zoom, fovy_deg * zoom

    def _light_angles(self, d):
        return d["light_azimuth"] * math.pi - 2 * math.pi, d["light_elevation"] * math.pi / 3 + math.pi / 6   # (sic) :190-191

    def _fovy_keys(self, f32, fovy, fovy_deg, proj):
        return {"fovy": f32(fovy_deg)}                                   # degrees (uncond_multiview.py:254)
