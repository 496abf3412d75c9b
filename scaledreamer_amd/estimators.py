"""`ImportanceEstimator` (threestudio/models/estimators.py:15-118) on the HIP path, plus the nerfacc surface it binds:
nerfacc.data_specs.RayIntervals, nerfacc.pdf.importance_sampling, nerfacc.volrend.render_transmittance_from_density.
The reference's control flow is kept call for call; the resampling, the transmittance cdf and the final cat + sort run in
the kernels of csrc/amortized.hip.  nerfacc draws the stratified jitter from its own Philox stream (unpinned): here it is
`jitter_fn(n_rays, device)` (default torch.rand), injectable by tests."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops


@dataclass
class RayIntervals:
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None


@dataclass
class RaySamples:
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor] = None


def importance_sampling(intervals: RayIntervals, cdfs: torch.Tensor, n_intervals_per_ray: int, stratified: bool = False,
                        jitter: Optional[torch.Tensor] = None) -> Tuple[RayIntervals, RaySamples]:
    if intervals.packed_info is not None or not isinstance(n_intervals_per_ray, int):
        raise NotImplementedError("packed (ragged) intervals: the amortized renderer uses dense [n_rays, edges] tensors")
    if stratified and jitter is None:
        jitter = torch.rand(cdfs.shape[0], device=cdfs.device)
    edges = ops.importance_resample(intervals.vals, cdfs, n_intervals_per_ray, jitter if stratified else None)
    return RayIntervals(vals=edges), RaySamples(vals=0.5 * (edges[:, 1:] + edges[:, :-1]))


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
    if packed_info is not None or ray_indices is not None or prefix_trans is not None:
        raise NotImplementedError("dense [n_rays, n_samples] tensors only")
    sd = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sd)
    trans = torch.exp(-(torch.cumsum(sd, dim=-1) - sd))
    return trans, alphas


def _transform_stot(transform_type: str, s_vals: torch.Tensor, t_min, t_max) -> torch.Tensor:
    if transform_type == "uniform":
        fn = ifn = lambda x: x
    elif transform_type == "lindisp":
        fn = ifn = lambda x: 1 / x
    else:
        raise ValueError(f"Unknown transform_type: {transform_type}")
    s_min, s_max = fn(t_min), fn(t_max)
    return ifn(s_vals * s_max + (1 - s_vals) * s_min)


class ImportanceEstimator(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("_dev", torch.zeros(0), persistent=False)
        self.jitter_fn: Callable = lambda n, device: torch.rand(n, device=device)

    @property
    def device(self):
        return self._dev.device

    @torch.no_grad()
    def sampling(self, prop_sigma_fns: List[Callable], prop_samples: List[int], num_samples: int, n_rays: int, near_plane: float,
                 far_plane: float, sampling_type: str = "uniform", stratified: bool = False,
                 requires_grad: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        assert len(prop_sigma_fns) == len(prop_samples), "The number of proposal networks and the number of samples should be the same."
        dev = self.device
        cdfs = torch.cat([torch.zeros((n_rays, 1), device=dev), torch.ones((n_rays, 1), device=dev)], dim=-1)
        intervals = RayIntervals(vals=cdfs)
        t_vals = None
        for level_fn, level_samples in zip(prop_sigma_fns, prop_samples):
            intervals, _ = importance_sampling(intervals, cdfs, level_samples, stratified, self.jitter_fn(n_rays, dev) if stratified else None)
            t_vals = _transform_stot(sampling_type, intervals.vals, near_plane, far_plane)
            t_starts, t_ends = t_vals[..., :-1], t_vals[..., 1:]
            with torch.set_grad_enabled(requires_grad):
                sigmas = level_fn(t_starts, t_ends)
                assert sigmas.shape == t_starts.shape
                cdfs = ops.transmittance_cdf(t_vals, sigmas)      # 1 - cat([trans, 0]) in one pass
        intervals, _ = importance_sampling(intervals, cdfs, num_samples, stratified, self.jitter_fn(n_rays, dev) if stratified else None)
        t_vals_fine = _transform_stot(sampling_type, intervals.vals, near_plane, far_plane)
        t_vals = ops.merge_sorted(t_vals, t_vals_fine)            # sort(cat([t_vals, t_vals_fine]))
        return t_vals[..., :-1], t_vals[..., 1:]
