"""CPU restatement of NeRFVolumeRenderer.forward (+ its backward) composed from the oracle primitives.

TEST INFRASTRUCTURE ONLY (see asd_oracle.c).  Follows threestudio/models/renderers/nerf_volume_renderer.py:
118-386 for the occgrid estimator with grid_prune=True / prune_alpha_threshold=True, training mode:
sampling (:139-180) -> positions (:269-279) -> geometry / material / background (:281-292) -> weights and
accumulations (:312-364).  Used (a) to pin the oracle against the goldens produced by the reference's own
glue (tests/test_goldens_cpu.py) and (b) as the CPU baseline of bench.py.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O


def forward(P: dict) -> dict:
    """P: rays_o/rays_d [Nr,3], jitter [Nr]|None, occs [32768], binaries, spp, radius, grid, w1d,w2d,w1f,w2f,
    bgrid, bw0,bw1,bw2.  Returns the output dictionary (flattened per-ray arrays) + a context for backward."""
    radius = float(P.get("radius", 1.0))
    rays_o = np.ascontiguousarray(P["rays_o"].reshape(-1, 3), np.float32)
    rays_d = np.ascontiguousarray(P["rays_d"].reshape(-1, 3), np.float32)
    nr = rays_o.shape[0]
    m, mb = O.grid_meta(), O.grid_meta(4, 2, 19, 4, 4.0)
    fc = O.field_cfg(radius=radius)
    mc = O.march_cfg(radius=radius, num_samples_per_ray=int(P["spp"]))
    diag = (3 * (2 * radius) ** 2) ** 0.5
    mc.max_steps = int(diag / float(mc.step)) + 3
    bits = O.pack_bits(np.asarray(P["binaries"]))
    count, offset, ray_idx, t0, t1, pts = O.march(mc, rays_o, rays_d, bits, P.get("jitter"))
    # sigma_fn + visibility pruning (nerfacc sampling with alpha_thre=0.01, early_stop_eps=1e-4)
    sig_c = O.field_density(m, fc, P["grid"], P["w1d"], P["w2d"], pts)
    alpha_thre = min(0.01, float(np.asarray(P["occs"], np.float32).mean(dtype=np.float32)))
    keep, kcount = O.prune(sig_c, t0, t1, offset, count, 1e-4, alpha_thre)
    sel = keep.astype(bool)
    ray_idx, t0, t1 = ray_idx[sel].astype(np.int64), t0[sel], t1[sel]
    koff = np.concatenate([[0], np.cumsum(kcount)[:-1]]).astype(np.int32)
    if ray_idx.size == 0:  # validate_empty_rays (utils/ops.py:514-520)
        ray_idx, t0, t1 = np.zeros(1, np.int64), np.zeros(1, np.float32), np.zeros(1, np.float32)
        kcount = np.zeros(nr, np.int32); kcount[0] = 1
        koff = np.ones(nr, np.int32); koff[0] = 0
    t_pos = (t0 + t1) / np.float32(2.0)
    positions = rays_o[ray_idx] + rays_d[ray_idx] * t_pos[:, None]
    sigma, feats, normal, enc = O.field_fwd(m, fc, P["grid"], P["w1d"], P["w2d"], P["w1f"], P["w2f"], positions, True)
    rgb = (1.0 / (1.0 + np.exp(-feats.astype(np.float32)))).astype(np.float32)  # no_material.py:48
    bg = O.envmap_fwd(mb, P["bgrid"], P["bw0"], P["bw1"], P["bw2"], rays_d)
    comp = O.composite_fwd(sigma, t0, t1, rgb, koff, kcount, bg)
    out = dict(comp_rgb=comp["comp_rgb"], comp_rgb_fg=comp["rgb_fg"], comp_rgb_bg=bg, opacity=comp["opacity"][:, None],
               depth=comp["depth"][:, None], z_variance=comp["z_var"][:, None], weights=comp["weights"][:, None],
               t_points=t_pos[:, None], t_intervals=(t1 - t0)[:, None], t_dirs=rays_d[ray_idx], ray_indices=ray_idx,
               points=positions, density=sigma[:, None], features=feats, normal=normal, shading_normal=normal)
    ctx = dict(m=m, mb=mb, fc=fc, sigma=sigma, t0=t0, t1=t1, rgb=rgb, koff=koff, kcount=kcount, bg=bg, comp=comp,
               positions=positions, rays_d=rays_d, n_candidates=int(count.sum()))
    return out, ctx


def backward(P: dict, ctx: dict, d_comp_rgb=None, d_opacity=None, d_depth=None, d_z_var=None, d_weights=None,
             d_normal=None, d_rgb_fg=None) -> dict:
    """Gradients of every trainable tensor given upstream gradients of the renderer outputs."""
    flat = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).reshape(-1) if a.ndim <= 2 and a.shape[-1] == 1 else np.ascontiguousarray(a, np.float32)
    d_sigma, d_rgb, d_bg = O.composite_bwd(ctx["sigma"], ctx["t0"], ctx["t1"], ctx["rgb"], ctx["koff"], ctx["kcount"],
                                           ctx["bg"], ctx["comp"], d_comp_rgb=flat(d_comp_rgb), d_rgb_fg=flat(d_rgb_fg),
                                           d_opacity=flat(d_opacity), d_depth=flat(d_depth), d_z_var=flat(d_z_var),
                                           d_weights=flat(d_weights))
    d_feat = (d_rgb * ctx["rgb"] * (1.0 - ctx["rgb"])).astype(np.float32)
    dgrid, dw1d, dw2d, dw1f, dw2f = O.field_bwd(ctx["m"], ctx["fc"], P["grid"], P["w1d"], P["w2d"], P["w1f"], P["w2f"],
                                                 ctx["positions"], d_sigma, d_feat, d_normal)
    dbgrid, dbw0, dbw1, dbw2 = O.envmap_bwd(ctx["mb"], P["bgrid"], P["bw0"], P["bw1"], P["bw2"], ctx["rays_d"], d_bg)
    return dict(grid=dgrid, w1d=dw1d, w2d=dw2d, w1f=dw1f, w2f=dw2f, bgrid=dbgrid, bw0=dbw0, bw1=dbw1, bw2=dbw2)
