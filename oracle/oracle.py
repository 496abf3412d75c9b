"""numpy front-end of oracle/asd_oracle.c (liboracle.so).

TEST INFRASTRUCTURE ONLY — see the header of asd_oracle.c.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by scaledreamer_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
MAXL = 16


class GridMeta(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("n_params", C.c_uint32),
                ("reserved", C.c_uint32), ("scale", C.c_float * MAXL), ("resolution", C.c_uint32 * MAXL),
                ("offset", C.c_uint32 * MAXL), ("size", C.c_uint32 * MAXL), ("dense", C.c_uint32 * MAXL)]


class FieldCfg(C.Structure):
    _fields_ = [("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("radius", C.c_float),
                ("bias_mode", C.c_int32), ("bias_value", C.c_float), ("blob_scale", C.c_float),
                ("blob_std", C.c_float), ("activation", C.c_int32), ("fd_eps", C.c_float),
                ("n_hidden", C.c_int32), ("n_feature_dims", C.c_int32), ("field_mode", C.c_int32)]


class MarchCfg(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("resolution", C.c_int32), ("near_plane", C.c_float),
                ("far_plane", C.c_float), ("step", C.c_float), ("max_steps", C.c_int32)]


_lib = None


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_grid_meta_init.restype = C.c_uint32
        _lib.orc_grid_meta_init.argtypes = [C.POINTER(GridMeta), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_double]
        _lib.orc_march.restype = C.c_int32
        _lib.orc_prune.restype = C.c_int32
    return _lib


def _p(a):
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
              per_level_scale=1.447269237440378) -> GridMeta:
    m = GridMeta()
    lib().orc_grid_meta_init(C.byref(m), n_levels, n_features, log2_hashmap_size, base_resolution,
                             float(per_level_scale))
    return m


def field_cfg(radius=1.0, bias_mode=1, bias_value=0.0, blob_scale=10.0, blob_std=0.5, activation=0, fd_eps=0.01,
              n_hidden=64, n_feature_dims=3, field_mode=0) -> FieldCfg:
    c = FieldCfg()
    for d in range(3):
        c.bbox_min[d] = -radius
        c.bbox_max[d] = radius
    c.radius, c.bias_mode, c.bias_value = radius, bias_mode, bias_value
    c.blob_scale, c.blob_std, c.activation, c.fd_eps = blob_scale, blob_std, activation, fd_eps
    c.n_hidden, c.n_feature_dims, c.field_mode = n_hidden, n_feature_dims, field_mode
    return c


def march_cfg(radius=1.0, resolution=32, near=0.0, far=1e10, num_samples_per_ray=512) -> MarchCfg:
    c = MarchCfg()
    for d in range(3):
        c.aabb[d] = -radius
        c.aabb[3 + d] = radius
    c.resolution, c.near_plane, c.far_plane = resolution, near, far
    # render_step_size (reference nerf_volume_renderer.py:66-68)
    c.step = np.float32(1.732 * 2 * radius / num_samples_per_ray)
    c.max_steps = int(num_samples_per_ray) + 2
    return c


# ------------------------------------------------------------------------------------------------
def hashgrid_fwd(m: GridMeta, params, x):
    params, x = _f(params), _f(x)
    out = np.empty((x.shape[0], m.n_levels * 2), np.float32)
    lib().orc_hashgrid_fwd(C.byref(m), _p(params), _p(x), C.c_int32(x.shape[0]), _p(out))
    return out


def hashgrid_bwd(m: GridMeta, x, dout):
    x, dout = _f(x), _f(dout)
    dparams = np.zeros(m.n_params, np.float32)
    lib().orc_hashgrid_bwd(C.byref(m), _p(x), _p(dout), C.c_int32(x.shape[0]), _p(dparams))
    return dparams


def field_density(m, c, grid, w1d, w2d, points):
    grid, w1d, w2d, points = _f(grid), _f(w1d), _f(w2d), _f(points)
    sigma = np.empty(points.shape[0], np.float32)
    lib().orc_field_density(C.byref(m), C.byref(c), _p(grid), _p(w1d), _p(w2d), _p(points),
                            C.c_int32(points.shape[0]), _p(sigma))
    return sigma


def field_fwd(m, c, grid, w1d, w2d, w1f, w2f, points, want_normal=True, want_fd_grad=False):
    grid, w1d, w2d, w1f, w2f, points = map(_f, (grid, w1d, w2d, w1f, w2f, points))
    n = points.shape[0]
    sigma = np.empty(n, np.float32)
    feats = np.empty((n, c.n_feature_dims), np.float32)
    normal = np.empty((n, 3), np.float32) if want_normal else None
    fd_grad = np.empty((n, 3), np.float32) if want_fd_grad else None
    enc = np.empty((n, m.n_levels * 2), np.float32)
    lib().orc_field_fwd(C.byref(m), C.byref(c), _p(grid), _p(w1d), _p(w2d), _p(w1f), _p(w2f), _p(points),
                        C.c_int32(n), _p(sigma), _p(feats), _p(normal), _p(fd_grad), _p(enc))
    if want_fd_grad:
        return sigma, feats, normal, fd_grad, enc
    return sigma, feats, normal, enc


def field_bwd(m, c, grid, w1d, w2d, w1f, w2f, points, d_sigma=None, d_features=None, d_normal=None, d_fd_grad=None):
    grid, w1d, w2d, w1f, w2f, points = map(_f, (grid, w1d, w2d, w1f, w2f, points))
    d_sigma, d_features, d_normal, d_fd_grad = _f(d_sigma), _f(d_features), _f(d_normal), _f(d_fd_grad)
    dgrid = np.zeros(m.n_params, np.float32)
    dw1d, dw2d = np.zeros_like(w1d), np.zeros_like(w2d)
    dw1f, dw2f = np.zeros_like(w1f), np.zeros_like(w2f)
    lib().orc_field_bwd(C.byref(m), C.byref(c), _p(grid), _p(w1d), _p(w2d), _p(w1f), _p(w2f), _p(points),
                        C.c_int32(points.shape[0]), _p(d_sigma), _p(d_features), _p(d_normal), _p(d_fd_grad), _p(dgrid),
                        _p(dw1d), _p(dw2d), _p(dw1f), _p(dw2f))
    return dgrid, dw1d, dw2d, dw1f, dw2f


# ---- amortized path: importance sampling, VolSDF density, voxel / tri-plane samplers ----------------------------
def importance_resample(vals, cdfs, n_out, jitter=None):
    vals, cdfs, jitter = _f(vals), _f(cdfs), _f(jitter)
    n_rays, e_in = vals.shape
    out = np.empty((n_rays, n_out + 1), np.float32)
    lib().orc_importance_resample(_p(vals), _p(cdfs), C.c_int32(n_rays), C.c_int32(e_in), C.c_int32(n_out), _p(jitter), _p(out))
    return out


def transmittance_cdf(t_edges, sigma):
    t_edges, sigma = _f(t_edges), _f(sigma)
    n_rays, S = sigma.shape
    cdf = np.empty((n_rays, S + 1), np.float32)
    lib().orc_transmittance_cdf(_p(t_edges), _p(sigma), C.c_int32(n_rays), C.c_int32(S), _p(cdf))
    return cdf


def merge_sorted(a, b):
    a, b = _f(a), _f(b)
    out = np.empty((a.shape[0], a.shape[1] + b.shape[1]), np.float32)
    lib().orc_merge_sorted(_p(a), C.c_int32(a.shape[1]), _p(b), C.c_int32(b.shape[1]), C.c_int32(a.shape[0]), _p(out))
    return out


def volsdf_density(sdf, inv_std):
    sdf = _f(sdf)
    out = np.empty_like(sdf)
    lib().orc_volsdf_density(_p(sdf), C.c_int64(sdf.size), C.c_float(inv_std), _p(out))
    return out


def voxel_sample_fwd(voxel_cl, points):
    voxel_cl, points = _f(voxel_cl), _f(points)
    B, D, H, W, Cc = voxel_cl.shape
    M = points.shape[1]
    out = np.empty((B, M, Cc), np.float32)
    lib().orc_voxel_sample_fwd(_p(voxel_cl), *(C.c_int32(v) for v in (B, D, H, W, Cc)), _p(points), C.c_int32(M), _p(out))
    return out


def voxel_sample_bwd(d_out, points, shape):
    d_out, points = _f(d_out), _f(points)
    B, D, H, W, Cc = shape
    dv = np.zeros(shape, np.float32)
    lib().orc_voxel_sample_bwd(_p(d_out), *(C.c_int32(v) for v in (B, D, H, W, Cc)), _p(points), C.c_int32(points.shape[1]), _p(dv))
    return dv


def triplane_sample_fwd(planes_cl, points, coord_scale=1.0):
    planes_cl, points = _f(planes_cl), _f(points)
    B, _, H, W, Cc = planes_cl.shape
    M = points.shape[1]
    out = np.empty((B, M, 3 * Cc), np.float32)
    lib().orc_triplane_sample_fwd(_p(planes_cl), *(C.c_int32(v) for v in (B, H, W, Cc)), _p(points), C.c_int32(M),
                                  C.c_float(coord_scale), _p(out))
    return out


def triplane_sample_bwd(d_out, points, shape, coord_scale=1.0):
    d_out, points = _f(d_out), _f(points)
    B, _, H, W, Cc = shape
    dp = np.zeros(shape, np.float32)
    lib().orc_triplane_sample_bwd(_p(d_out), *(C.c_int32(v) for v in (B, H, W, Cc)), _p(points), C.c_int32(points.shape[1]),
                                  C.c_float(coord_scale), _p(dp))
    return dp


def envmap_fwd(m, grid, w0, w1, w2, dirs):
    grid, w0, w1, w2, dirs = map(_f, (grid, w0, w1, w2, dirs))
    color = np.empty((dirs.shape[0], 3), np.float32)
    lib().orc_envmap_fwd(C.byref(m), _p(grid), _p(w0), _p(w1), _p(w2), C.c_int32(w1.shape[0]), _p(dirs),
                         C.c_int32(dirs.shape[0]), _p(color))
    return color


def envmap_bwd(m, grid, w0, w1, w2, dirs, d_color):
    grid, w0, w1, w2, dirs, d_color = map(_f, (grid, w0, w1, w2, dirs, d_color))
    dgrid = np.zeros(m.n_params, np.float32)
    dw0, dw1, dw2 = np.zeros_like(w0), np.zeros_like(w1), np.zeros_like(w2)
    lib().orc_envmap_bwd(C.byref(m), _p(grid), _p(w0), _p(w1), _p(w2), C.c_int32(w1.shape[0]), _p(dirs),
                         _p(d_color), C.c_int32(dirs.shape[0]), _p(dgrid), _p(dw0), _p(dw1), _p(dw2))
    return dgrid, dw0, dw1, dw2


def pack_bits(binaries: np.ndarray) -> np.ndarray:
    """bool[res^3] (nerfacc binaries, flattened ix,iy,iz) -> uint32 words, bit i of word i>>5."""
    b = np.ascontiguousarray(binaries.reshape(-1).astype(np.uint8))
    pad = (-b.size) % 32
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    return np.packbits(b.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).copy()


def march(c: MarchCfg, rays_o, rays_d, occ_bits, jitter=None):
    rays_o, rays_d, jitter = _f(rays_o), _f(rays_d), _f(jitter)
    occ_bits = np.ascontiguousarray(occ_bits, dtype=np.uint32)
    n = rays_o.shape[0]
    count = np.zeros(n, np.int32)
    total = lib().orc_march(C.byref(c), _p(rays_o), _p(rays_d), C.c_int32(n), _p(occ_bits), _p(jitter), _p(count),
                            None, None, None, None)
    ray_idx = np.empty(total, np.int32)
    t0 = np.empty(total, np.float32)
    t1 = np.empty(total, np.float32)
    pts = np.empty((total, 3), np.float32)
    lib().orc_march(C.byref(c), _p(rays_o), _p(rays_d), C.c_int32(n), _p(occ_bits), _p(jitter), _p(count),
                    _p(ray_idx), _p(t0), _p(t1), _p(pts))
    offset = np.concatenate([[0], np.cumsum(count)[:-1]]).astype(np.int32) if n else np.zeros(0, np.int32)
    return count, offset, ray_idx, t0, t1, pts


def prune(sigma, t0, t1, offset, count, early_stop_eps=1e-4, alpha_thre=0.0):
    sigma, t0, t1 = _f(sigma), _f(t0), _f(t1)
    offset = np.ascontiguousarray(offset, np.int32)
    count = np.ascontiguousarray(count, np.int32)
    keep = np.zeros(sigma.shape[0], np.uint8)
    kept = np.zeros(count.shape[0], np.int32)
    lib().orc_prune(_p(sigma), _p(t0), _p(t1), _p(offset), _p(count), C.c_int32(count.shape[0]),
                    C.c_float(early_stop_eps), C.c_float(alpha_thre), _p(keep), _p(kept))
    return keep, kept


def composite_fwd(sigma, t0, t1, rgb, offset, count, bg, mode=0):
    sigma, t0, t1, rgb, bg = map(_f, (sigma, t0, t1, rgb, bg))
    offset = np.ascontiguousarray(offset, np.int32)
    count = np.ascontiguousarray(count, np.int32)
    n, nr = sigma.shape[0], count.shape[0]
    w = np.zeros(n, np.float32)
    op, dp, zv = (np.zeros(nr, np.float32) for _ in range(3))
    fg, comp = np.zeros((nr, 3), np.float32), np.zeros((nr, 3), np.float32)
    lib().orc_composite_fwd(C.c_int32(mode), _p(sigma), _p(t0), _p(t1), _p(rgb), _p(offset), _p(count),
                            C.c_int32(nr), _p(bg), _p(w), _p(op), _p(dp), _p(fg), _p(zv), _p(comp))
    return dict(weights=w, opacity=op, depth=dp, rgb_fg=fg, z_var=zv, comp_rgb=comp)


def composite_bwd(sigma, t0, t1, rgb, offset, count, bg, fwd, d_comp_rgb=None, d_rgb_fg=None, d_opacity=None,
                  d_depth=None, d_z_var=None, d_weights=None, mode=0):
    sigma, t0, t1, rgb, bg = map(_f, (sigma, t0, t1, rgb, bg))
    offset = np.ascontiguousarray(offset, np.int32)
    count = np.ascontiguousarray(count, np.int32)
    ups = [_f(a) for a in (d_comp_rgb, d_rgb_fg, d_opacity, d_depth, d_z_var, d_weights)]
    n, nr = sigma.shape[0], count.shape[0]
    d_sigma = np.zeros(n, np.float32)
    d_rgb = np.zeros((n, 3), np.float32)
    d_bg = np.zeros((nr, 3), np.float32)
    lib().orc_composite_bwd(C.c_int32(mode), _p(sigma), _p(t0), _p(t1), _p(rgb), _p(offset), _p(count),
                            C.c_int32(nr), _p(bg), _p(fwd["weights"]), _p(fwd["opacity"]), _p(fwd["depth"]),
                            *[_p(u) for u in ups], _p(d_sigma), _p(d_rgb), _p(d_bg))
    return d_sigma, d_rgb, d_bg


def occgrid_update(occs, cell_idx, occ_new, decay=0.95, occ_thre=0.01):
    occs = np.ascontiguousarray(occs, np.float32).copy()
    cell_idx = np.ascontiguousarray(cell_idx, np.int32)
    occ_new = _f(occ_new)
    n = occs.shape[0]
    bits = np.zeros((n + 31) // 32, np.uint32)
    binaries = np.zeros(n, np.uint8)
    lib().orc_occgrid_update(_p(occs), C.c_int32(n), _p(cell_idx), _p(occ_new), C.c_int32(cell_idx.shape[0]),
                             C.c_float(decay), C.c_float(occ_thre), _p(bits), _p(binaries))
    return occs, bits, binaries


def generate_rays(c2w, focal, H: int, W: int, normalize: bool = True):
    """c2w [B,4,4], focal [B] (pixels) -> rays_o, rays_d [B,H,W,3]  (orc_generate_rays: ops.py:183-269)"""
    c2w = np.ascontiguousarray(c2w, np.float32)
    focal = np.ascontiguousarray(focal, np.float32)
    B = c2w.shape[0]
    ro, rd = np.empty((B, H, W, 3), np.float32), np.empty((B, H, W, 3), np.float32)
    fp = C.POINTER(C.c_float)
    lib().orc_generate_rays(c2w.ctypes.data_as(fp), focal.ctypes.data_as(fp), C.c_int32(B), C.c_int32(H), C.c_int32(W), C.c_int32(int(normalize)),
                            ro.ctypes.data_as(fp), rd.ctypes.data_as(fp))
    return ro, rd
