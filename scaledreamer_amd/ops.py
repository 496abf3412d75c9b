"""Raw (non-autograd) tensor-level wrappers of the C ABI in include/asd_hip.h.

Every function takes CUDA(HIP) float32 tensors, allocates the outputs with torch (so the caching allocator
and the current stream own them) and enqueues the kernels on the current stream.  No synchronisation.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import FieldCfg, GridMeta, MarchCfg, check, f32, i32, lib, ptr, stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.AsdError("the HIP path needs device tensors (there is no CPU fallback)")


class _Keep:
    """fp32-contiguous view of an argument as a device pointer; the converted tensor is held until this object dies, i.e. until
    after the launch has been enqueued.  (`ptr(t.contiguous())` alone frees the copy as soon as ptr() returns: the caching
    allocator then hands the same block to the next conversion in the argument list and two pointers alias.)"""

    def __init__(self):
        self.held = []

    def __call__(self, t: Optional[torch.Tensor], dtype=torch.float32):
        t = _c(t, dtype)
        if t is not None:
            self.held.append(t)
        return ptr(t)


def _c(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


# ---- hash grid --------------------------------------------------------------------------------
def hashgrid_fwd(meta: GridMeta, params: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _need_cuda(params, x)
    x = _c(x)
    out = torch.empty((x.shape[0], meta.n_levels * meta.n_features), device=x.device, dtype=torch.float32)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_hashgrid_fwd(C.byref(meta), k(params), ptr(x), i32(x.shape[0]), ptr(out), stream()))
    return out


def hashgrid_bwd(meta: GridMeta, x: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    _need_cuda(x, dout)
    x, dout = _c(x), _c(dout)
    dparams = torch.zeros(meta.n_params, device=x.device, dtype=torch.float32)
    check(lib().asd_hashgrid_bwd(C.byref(meta), ptr(x), ptr(dout), i32(x.shape[0]), ptr(dparams), stream()))
    return dparams


# ---- field ------------------------------------------------------------------------------------
def field_density(meta, cfg: FieldCfg, grid, w1d, w2d, points, n_dev: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _need_cuda(grid, points)
    points = _c(points)
    n = points.shape[0]
    sigma = out if out is not None else torch.empty(n, device=points.device, dtype=torch.float32)
    check(lib().asd_field_density(C.byref(meta), C.byref(cfg), ptr(grid), ptr(w1d), ptr(w2d), ptr(points), i32(n),
                                  ptr(n_dev), ptr(sigma), stream()))
    return sigma


def field_fwd(meta, cfg: FieldCfg, grid, w1d, w2d, w1f, w2f, points, want_normal: bool,
              n_dev: Optional[torch.Tensor] = None, want_fd_grad: bool = False):
    """-> (sigma|sdf, features, normal, enc)   or, with want_fd_grad, (sdf, features, normal, fd_grad, enc)."""
    _need_cuda(grid, points)
    points = _c(points)
    n, dev = points.shape[0], points.device
    sigma = torch.empty(n, device=dev, dtype=torch.float32)
    feats = torch.empty((n, cfg.n_feature_dims), device=dev, dtype=torch.float32) if cfg.n_feature_dims > 0 else None
    normal = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_normal else None
    fd_grad = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_fd_grad else None
    enc = torch.empty((n, meta.n_levels * 2), device=dev, dtype=torch.float32)
    check(lib().asd_field_fwd(C.byref(meta), C.byref(cfg), ptr(grid), ptr(w1d), ptr(w2d), ptr(w1f), ptr(w2f),
                              ptr(points), i32(n), ptr(n_dev), ptr(sigma), ptr(feats), ptr(normal), ptr(fd_grad), ptr(enc),
                              stream()))
    if want_fd_grad:
        return sigma, feats, normal, fd_grad, enc
    return sigma, feats, normal, enc


def field_bwd(meta, cfg: FieldCfg, grid, w1d, w2d, w1f, w2f, points, enc, sigma, d_sigma, d_features, d_normal,
              d_grid: torch.Tensor, n_dev: Optional[torch.Tensor] = None, d_fd_grad=None):
    """Accumulates into d_grid (atomics); returns (dw1d, dw2d, dw1f, dw2f)."""
    points = _c(points)
    n, dev = points.shape[0], points.device
    nf = C.c_int64(0)
    check(lib().asd_field_bwd_workspace(C.byref(cfg), i32(n), i32(int(d_normal is not None or d_fd_grad is not None)), C.byref(nf)))
    ws = torch.empty(nf.value, device=dev, dtype=torch.float32)
    H, Cf = cfg.n_hidden, cfg.n_feature_dims
    dw1d = torch.zeros((H, 32), device=dev, dtype=torch.float32)
    dw2d = torch.zeros((1, H), device=dev, dtype=torch.float32)
    dw1f = torch.zeros((H, 32), device=dev, dtype=torch.float32) if Cf > 0 else None
    dw2f = torch.zeros((Cf, H), device=dev, dtype=torch.float32) if Cf > 0 else None
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_field_bwd(C.byref(meta), C.byref(cfg), ptr(grid), ptr(w1d), ptr(w2d), ptr(w1f), ptr(w2f),
                              ptr(points), ptr(enc), ptr(sigma), i32(n), ptr(n_dev), k(d_sigma),
                              k(d_features), k(d_normal), k(d_fd_grad), ptr(d_grid), ptr(dw1d), ptr(dw2d), ptr(dw1f),
                              ptr(dw2f), ptr(ws), stream()))
    return dw1d, dw2d, dw1f, dw2f


# ---- background -------------------------------------------------------------------------------
def envmap_fwd(meta, grid, w0, w1, w2, dirs) -> torch.Tensor:
    _need_cuda(grid, dirs)
    dirs = _c(dirs)
    color = torch.empty((dirs.shape[0], 3), device=dirs.device, dtype=torch.float32)
    check(lib().asd_envmap_fwd(C.byref(meta), ptr(grid), ptr(w0), ptr(w1), ptr(w2), i32(w1.shape[0]), ptr(dirs),
                               i32(dirs.shape[0]), ptr(color), stream()))
    return color


def envmap_bwd(meta, grid, w0, w1, w2, dirs, d_color):
    dirs, d_color = _c(dirs), _c(d_color)
    dev = dirs.device
    dgrid = torch.zeros(meta.n_params, device=dev, dtype=torch.float32)
    dw0, dw1, dw2 = torch.zeros_like(w0), torch.zeros_like(w1), torch.zeros_like(w2)
    check(lib().asd_envmap_bwd(C.byref(meta), ptr(grid), ptr(w0), ptr(w1), ptr(w2), i32(w1.shape[0]), ptr(dirs),
                               ptr(d_color), i32(dirs.shape[0]), ptr(dgrid), ptr(dw0), ptr(dw1), ptr(dw2), stream()))
    return dgrid, dw0, dw1, dw2


# ---- marching ---------------------------------------------------------------------------------
def scan_i32(count: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    offset = torch.empty_like(count)
    total = torch.empty(1, device=count.device, dtype=torch.int32)
    check(lib().asd_scan_i32(ptr(count), i32(count.shape[0]), ptr(offset), ptr(total), stream()))
    return offset, total


def march(cfg: MarchCfg, rays_o, rays_d, occ_bits, jitter=None, n_max: Optional[int] = None):
    """Two-pass marcher. Returns (count, offset, total_dev, ray_idx, t0, t1, points); the sample arrays are
    sized n_max (default: exact, which costs one host sync)."""
    _need_cuda(rays_o, rays_d, occ_bits)
    rays_o, rays_d, jitter = _c(rays_o), _c(rays_d), _c(jitter)
    nr, dev = rays_o.shape[0], rays_o.device
    count = torch.empty(nr, device=dev, dtype=torch.int32)
    check(lib().asd_march_count(C.byref(cfg), ptr(rays_o), ptr(rays_d), i32(nr), ptr(occ_bits), ptr(jitter),
                                ptr(count), stream()))
    offset, total = scan_i32(count)
    if n_max is None:
        n_max = int(total.item())
    ray_idx = torch.empty(n_max, device=dev, dtype=torch.int32)
    t0 = torch.empty(n_max, device=dev, dtype=torch.float32)
    t1 = torch.empty(n_max, device=dev, dtype=torch.float32)
    pts = torch.empty((n_max, 3), device=dev, dtype=torch.float32)
    check(lib().asd_march_write(C.byref(cfg), ptr(rays_o), ptr(rays_d), i32(nr), ptr(occ_bits), ptr(jitter),
                                ptr(offset), ptr(ray_idx), ptr(t0), ptr(t1), ptr(pts), stream()))
    return count, offset, total, ray_idx, t0, t1, pts


def prune(sigma, t0, t1, offset, count, early_stop_eps: float, alpha_thre: float):
    nr = count.shape[0]
    keep = torch.empty(sigma.shape[0], device=sigma.device, dtype=torch.uint8)
    kept = torch.empty(nr, device=sigma.device, dtype=torch.int32)
    check(lib().asd_prune_count(ptr(sigma), ptr(t0), ptr(t1), ptr(offset), ptr(count), i32(nr), f32(early_stop_eps),
                                f32(alpha_thre), ptr(keep), ptr(kept), stream()))
    return keep, kept


def compact(rays_o, rays_d, offset, count, keep, t0, t1, kept_offset, n_out: int, zero_ray_idx: bool = False):
    """n_out may be an upper bound of the kept count (rows behind the kept ones stay unwritten); zero_ray_idx: those rows of ray_idx
    then read 0, so the buffer stays a valid index tensor"""
    nr, dev = count.shape[0], t0.device
    ray_idx = (torch.zeros if zero_ray_idx else torch.empty)(n_out, device=dev, dtype=torch.int64)
    t0o = torch.empty(n_out, device=dev, dtype=torch.float32)
    t1o = torch.empty(n_out, device=dev, dtype=torch.float32)
    pts = torch.empty((n_out, 3), device=dev, dtype=torch.float32)
    dirs = torch.empty((n_out, 3), device=dev, dtype=torch.float32)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_compact(k(rays_o), k(rays_d), i32(nr), ptr(offset), ptr(count), ptr(keep), ptr(t0),
                            ptr(t1), ptr(kept_offset), ptr(ray_idx), ptr(t0o), ptr(t1o), ptr(pts), ptr(dirs),
                            stream()))
    return ray_idx, t0o, t1o, pts, dirs


def occgrid_update(occs, cell_idx, occ_new, decay: float, occ_thre: float, occ_bits, binaries):
    scratch = torch.empty(2, device=occs.device, dtype=torch.float32)
    n_up = 0 if cell_idx is None else cell_idx.shape[0]
    check(lib().asd_occgrid_update(ptr(occs), i32(occs.numel()), ptr(cell_idx), ptr(occ_new), i32(n_up), f32(decay),
                                   f32(occ_thre), ptr(occ_bits), ptr(binaries), ptr(scratch), stream()))


def generate_rays(c2w: torch.Tensor, focal: torch.Tensor, H: int, W: int, normalize: bool = True):
    """c2w [B,4,4], focal [B] (pixels), device tensors -> rays_o, rays_d [B,H,W,3] (include/asd_hip.h: asd_generate_rays)"""
    _need_cuda(c2w, focal)
    c2w, focal = _c(c2w), _c(focal)
    B = c2w.shape[0]
    rays_o = torch.empty((B, H, W, 3), device=c2w.device, dtype=torch.float32)
    rays_d = torch.empty((B, H, W, 3), device=c2w.device, dtype=torch.float32)
    check(lib().asd_generate_rays(ptr(c2w), ptr(focal), i32(B), i32(H), i32(W), i32(int(normalize)), ptr(rays_o), ptr(rays_d), stream()))
    return rays_o, rays_d


# ---- compositing ------------------------------------------------------------------------------
def composite_fwd(sigma, t0, t1, rgb, offset, count, bg, mode: int = 0):
    nr, n, dev = count.shape[0], sigma.shape[0], sigma.device
    w = torch.empty(n, device=dev, dtype=torch.float32)
    op = torch.empty(nr, device=dev, dtype=torch.float32)
    dp = torch.empty(nr, device=dev, dtype=torch.float32)
    zv = torch.empty(nr, device=dev, dtype=torch.float32)
    fg = torch.empty((nr, 3), device=dev, dtype=torch.float32)
    comp = torch.empty((nr, 3), device=dev, dtype=torch.float32)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_composite_fwd(i32(mode), k(sigma), ptr(t0), ptr(t1), k(rgb), ptr(offset), ptr(count),
                                  i32(nr), k(bg), ptr(w), ptr(op), ptr(dp), ptr(fg), ptr(zv), ptr(comp),
                                  stream()))
    return dict(weights=w, opacity=op, depth=dp, rgb_fg=fg, z_var=zv, comp_rgb=comp)


def composite_bwd(sigma, t0, t1, rgb, offset, count, bg, fwd, d_comp_rgb=None, d_rgb_fg=None, d_opacity=None,
                  d_depth=None, d_z_var=None, d_weights=None, mode: int = 0):
    nr, n, dev = count.shape[0], sigma.shape[0], sigma.device
    d_sigma = torch.empty(n, device=dev, dtype=torch.float32)
    d_rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    d_bg = torch.empty((nr, 3), device=dev, dtype=torch.float32)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_composite_bwd(i32(mode), k(sigma), ptr(t0), ptr(t1), k(rgb), ptr(offset), ptr(count),
                                  i32(nr), k(bg), ptr(fwd["weights"]), ptr(fwd["opacity"]), ptr(fwd["depth"]),
                                  k(d_comp_rgb), k(d_rgb_fg), k(d_opacity), k(d_depth),
                                  k(d_z_var), k(d_weights), ptr(d_sigma), ptr(d_rgb), ptr(d_bg), stream()))
    return d_sigma, d_rgb, d_bg


def pack_bits(binaries: torch.Tensor) -> torch.Tensor:
    """bool[res^3] -> uint32 words (bit i of word i>>5), stored as int32 on the device."""
    b = binaries.reshape(-1).to(torch.int64)
    pad = (-b.numel()) % 32
    if pad:
        b = torch.cat([b, b.new_zeros(pad)])
    w = (b.view(-1, 32) << torch.arange(32, device=b.device, dtype=torch.int64)).sum(1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
    return w.to(torch.int32).contiguous()


# ---- amortized path: importance sampling, voxel / tri-plane samplers ------------------------------------------
def importance_resample(vals: torch.Tensor, cdfs: torch.Tensor, n_out: int, jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vals, cdfs [n_rays, e_in] -> resampled edges [n_rays, n_out + 1] (include/asd_hip.h: asd_importance_resample)."""
    _need_cuda(vals, cdfs)
    vals, cdfs = _c(vals), _c(cdfs)
    n_rays, e_in = vals.shape
    out = torch.empty((n_rays, n_out + 1), device=vals.device, dtype=torch.float32)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_importance_resample(ptr(vals), ptr(cdfs), i32(n_rays), i32(e_in), i32(n_out), k(jitter), ptr(out), stream()))
    return out


def transmittance_cdf(t_edges: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    _need_cuda(t_edges, sigma)
    t_edges, sigma = _c(t_edges), _c(sigma)
    n_rays, S = sigma.shape
    assert t_edges.shape == (n_rays, S + 1)
    cdf = torch.empty((n_rays, S + 1), device=sigma.device, dtype=torch.float32)
    check(lib().asd_transmittance_cdf(ptr(t_edges), ptr(sigma), i32(n_rays), i32(S), ptr(cdf), stream()))
    return cdf


def merge_sorted(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_cuda(a, b)
    a, b = _c(a), _c(b)
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), device=a.device, dtype=torch.float32)
    check(lib().asd_merge_sorted(ptr(a), i32(a.shape[1]), ptr(b), i32(b.shape[1]), i32(a.shape[0]), ptr(out), stream()))
    return out


def relayout(x: torch.Tensor) -> torch.Tensor:
    """[batch, rows, cols] -> [batch, cols, rows] (fp32)."""
    _need_cuda(x)
    x = _c(x)
    b, r, c = x.shape
    y = torch.empty((b, c, r), device=x.device, dtype=torch.float32)
    check(lib().asd_relayout_f32(ptr(x), i32(b), i32(r), i32(c), ptr(y), stream()))
    return y


def voxel_sample_fwd(voxel_cl: torch.Tensor, points: torch.Tensor) -> torch.Tensor:
    """voxel_cl [B,D,H,W,C], points [B,M,3] -> [B,M,C]"""
    _need_cuda(voxel_cl, points)
    B, D, H, W, Cc = voxel_cl.shape
    points = _c(points)
    M = points.shape[1]
    out = torch.empty((B, M, Cc), device=points.device, dtype=torch.float32)
    check(lib().asd_voxel_sample_fwd(ptr(voxel_cl), i32(B), i32(D), i32(H), i32(W), i32(Cc), ptr(points), i32(M), ptr(out), stream()))
    return out


def voxel_sample_bwd(d_out: torch.Tensor, points: torch.Tensor, shape) -> torch.Tensor:
    B, D, H, W, Cc = shape
    d_voxel = torch.zeros(shape, device=d_out.device, dtype=torch.float32)
    points = _c(points)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_voxel_sample_bwd(k(d_out), i32(B), i32(D), i32(H), i32(W), i32(Cc), ptr(points), i32(points.shape[1]),
                                     ptr(d_voxel), stream()))
    return d_voxel


def triplane_sample_fwd(planes_cl: torch.Tensor, points: torch.Tensor, coord_scale: float) -> torch.Tensor:
    """planes_cl [B,3,H,W,C], points [B,M,3] -> [B,M,3C]"""
    _need_cuda(planes_cl, points)
    B, _, H, W, Cc = planes_cl.shape
    points = _c(points)
    M = points.shape[1]
    out = torch.empty((B, M, 3 * Cc), device=points.device, dtype=torch.float32)
    check(lib().asd_triplane_sample_fwd(ptr(planes_cl), i32(B), i32(H), i32(W), i32(Cc), ptr(points), i32(M), f32(coord_scale), ptr(out),
                                        stream()))
    return out


def triplane_sample_bwd(d_out: torch.Tensor, points: torch.Tensor, shape, coord_scale: float) -> torch.Tensor:
    B, _, H, W, Cc = shape
    d_planes = torch.zeros(shape, device=d_out.device, dtype=torch.float32)
    points = _c(points)
    k = _Keep()   # converted temporaries must outlive the launch (a freed block would be handed to the next conversion)
    check(lib().asd_triplane_sample_bwd(k(d_out), i32(B), i32(H), i32(W), i32(Cc), ptr(points), i32(points.shape[1]),
                                        f32(coord_scale), ptr(d_planes), stream()))
    return d_planes


# ---- generator backbone: split-fp16 3x3x3 convolution + layer tail on channel-last fp32 volumes (csrc/conv3d.hip) ----------------------
_conv3d_ws = {}     # (device, stream) -> grow-only workspace (uint8): the passes of one convolution share it, one conv runs at a time on a stream
_zero_pages = {}


def _ws(device, nbytes: int) -> torch.Tensor:
    """the convolution workspace of the CURRENT stream.  When it has to grow, the old buffer may still be read by kernels queued on that
    stream: record_stream keeps the caching allocator from handing it out before they have run."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    cur = _conv3d_ws.get(key)
    if cur is None or cur.numel() < nbytes:
        if cur is not None:
            cur.record_stream(torch.cuda.current_stream(device))
        cur = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        _conv3d_ws[key] = cur
    return cur


def free_workspaces() -> None:
    """release the grow-only convolution workspaces (e.g. before evaluation or torch.cuda.empty_cache()): the 128^3 weight-gradient planes and
    slabs otherwise stay resident for the life of the process"""
    for key, buf in list(_conv3d_ws.items()):
        buf.record_stream(torch.cuda.ExternalStream(key[1], device=key[0]))     # the stream the workspace was keyed by, not the current one
    _conv3d_ws.clear()


def _zero_page(device) -> torch.Tensor:
    z = _zero_pages.get(device)
    if z is None:
        z = _zero_pages[device] = torch.zeros(64, device=device, dtype=torch.float32)
    return z


def new_amax(device) -> torch.Tensor:
    """a zeroed device word that a producer kernel fills with the bit pattern of max|output| (include/asd_hip.h: amax_out)"""
    return torch.zeros(1, device=device, dtype=torch.int32)


def absmax(x: torch.Tensor) -> torch.Tensor:
    """the max|x| word of a tensor whose producer left none (asd_absmax_f32)"""
    x = _c(x)
    a = new_amax(x.device)
    check(lib().asd_absmax_f32(ptr(x), C.c_int64(x.numel()), _ap(a), stream()))
    return a


def _ap(t: Optional[torch.Tensor]):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _epilogue(bias=None, noise=None, noise_strength=None, act: bool = False, gain: float = 1.0, clamp: float = 0.0, amax_out=None):
    """-> (Conv3dEpilogue, tensors to keep alive)"""
    keep = [_c(t) for t in (bias, noise, noise_strength)]
    ep = L.Conv3dEpilogue(ptr(keep[0]).value, ptr(keep[1]).value, ptr(keep[2]).value, int(bool(act)), float(gain), float(clamp), _ap(amax_out).value)
    return ep, keep


def conv3d_fwd(x: torch.Tensor, w: torch.Tensor, bias=None, noise=None, noise_strength=None, act: bool = False, gain: float = 1.0,
               clamp: float = 0.0, amax_x=None, amax_out=None) -> torch.Tensor:
    """x [N,D,H,W,Cin] fp32 channel-last, w [N,Cout,Cin,3,3,3] (per sample) or [Cout,Cin,3,3,3] -> act(conv + noise * ns + bias) [N,D,H,W,Cout].
    amax_x: the producer's max|x| word if it left one (else one more pass over x); amax_out: a zeroed word that receives max|y|"""
    _need_cuda(x, w)
    x, w = _c(x), _c(w)
    cout = w.shape[-5]
    N, D, H, W, Cin = x.shape
    d = L.Conv3dDesc(int(N), int(D), int(H), int(W), int(Cin), int(cout), _ap(amax_x).value, None)
    y = torch.empty((N, D, H, W, cout), device=x.device, dtype=torch.float32)
    nb = lib().asd_conv3d_workspace_bytes(C.byref(d), i32(0))
    ws = _ws(x.device, nb)
    ep, keep = _epilogue(bias, noise, noise_strength, act, gain, clamp, amax_out)
    stride = cout * Cin * 27 if w.dim() == 6 else 0
    check(lib().asd_conv3d_fwd(C.byref(d), ptr(x), ptr(w), C.c_int64(stride), ptr(y), C.byref(ep), ptr(ws), C.c_int64(nb), stream()))
    return y


def conv3d_dgrad(dy: torch.Tensor, w: torch.Tensor, cin: int, amax_dy=None) -> torch.Tensor:
    _need_cuda(dy, w)
    dy, w = _c(dy), _c(w)
    N, D, H, W, cout = dy.shape
    d = L.Conv3dDesc(int(N), int(D), int(H), int(W), int(cin), int(cout), None, _ap(amax_dy).value)
    dx = torch.empty((N, D, H, W, cin), device=dy.device, dtype=torch.float32)
    nb = lib().asd_conv3d_workspace_bytes(C.byref(d), i32(1))
    ws = _ws(dy.device, nb)
    stride = cout * cin * 27 if w.dim() == 6 else 0
    check(lib().asd_conv3d_dgrad(C.byref(d), ptr(dy), ptr(w), C.c_int64(stride), ptr(dx), ptr(ws), C.c_int64(nb), stream()))
    return dx


def conv3d_wgrad(x: torch.Tensor, dy: torch.Tensor, amax_x=None, amax_dy=None) -> torch.Tensor:
    """-> dw [N,Cout,Cin,3,3,3] (one gradient per sample: the modulated convolution has per-sample weights)"""
    _need_cuda(x, dy)
    x, dy = _c(x), _c(dy)
    N, D, H, W, cin = x.shape
    cout = dy.shape[4]
    d = L.Conv3dDesc(int(N), int(D), int(H), int(W), int(cin), int(cout), _ap(amax_x).value, _ap(amax_dy).value)
    dw = torch.empty((N, cout, cin, 3, 3, 3), device=x.device, dtype=torch.float32)
    nb = lib().asd_conv3d_workspace_bytes(C.byref(d), i32(2))
    ws = _ws(x.device, nb)
    check(lib().asd_conv3d_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(dw), C.c_int64(cout * cin * 27), ptr(ws), C.c_int64(nb),
                                 ptr(_zero_page(x.device)), stream()))
    return dw


def layer_act_bwd(dy: torch.Tensor, y: torch.Tensor, gain: float, clamp: float, want_bias: bool = True, want_rowsum: bool = True,
                  sub: Optional[torch.Tensor] = None, amax_out=None, act_mask: Optional[torch.Tensor] = None):
    """dz = dy * act'(y - sub) on [..., C] (act_mask: the branch bits upsample3d_fwd recorded, instead of reading them off y - sub);
    -> (dz, d_bias [C] | None, d_rowsum [rows] | None)"""
    dy, y, sub = _c(dy), _c(y), _c(sub)
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    dz = torch.empty_like(y)
    d_bias = torch.empty(Cc, device=y.device, dtype=torch.float32) if want_bias else None
    d_rowsum = torch.empty(rows, device=y.device, dtype=torch.float32) if want_rowsum else None
    check(lib().asd_layer_act_bwd(ptr(dy), ptr(y), ptr(sub), ptr(act_mask), C.c_int64(rows), i32(Cc), f32(gain), f32(clamp), ptr(dz), ptr(d_bias),
                                  ptr(d_rowsum), _ap(amax_out), stream()))
    return dz, d_bias, d_rowsum


def modulated_weights_fwd(weight: torch.Tensor, styles: torch.Tensor, gain: float = 1.0, demodulate: bool = True):
    """weight [Cout, Cin, *k], styles [N, Cin] -> (wm [N, Cout, Cin, *k], dcoef [N, Cout]): weight * styles * gain, demodulated per (n, co)"""
    _need_cuda(weight, styles)
    weight, styles = _c(weight), _c(styles)
    cout, cin = weight.shape[0], weight.shape[1]
    K = weight.numel() // (cout * cin)
    N = styles.shape[0]
    wm = torch.empty((N,) + tuple(weight.shape), device=weight.device, dtype=torch.float32)
    dcoef = torch.empty((N, cout), device=weight.device, dtype=torch.float32)
    check(lib().asd_modulated_weights_fwd(ptr(weight), ptr(styles), i32(N), i32(cout), i32(cin), i32(K), f32(gain), i32(int(demodulate)), ptr(wm),
                                          ptr(dcoef), stream()))
    return wm, dcoef


def modulated_weights_bwd(d_wm: torch.Tensor, wm: torch.Tensor, weight: torch.Tensor, styles: torch.Tensor, dcoef: torch.Tensor, gain: float = 1.0,
                          demodulate: bool = True):
    """-> (d_weight like weight, d_styles [N, Cin])"""
    d_wm, wm, weight, styles = _c(d_wm), _c(wm), _c(weight), _c(styles)
    cout, cin = weight.shape[0], weight.shape[1]
    K = weight.numel() // (cout * cin)
    N = styles.shape[0]
    d_weight = torch.empty_like(weight)
    d_styles = torch.empty_like(styles)
    check(lib().asd_modulated_weights_bwd(ptr(d_wm), ptr(wm), ptr(weight), ptr(styles), ptr(dcoef), i32(N), i32(cout), i32(cin), i32(K), f32(gain),
                                          i32(int(demodulate)), ptr(d_weight), ptr(d_styles), stream()))
    return d_weight, d_styles


def upsample3d_fwd(x: torch.Tensor, bias=None, noise=None, noise_strength=None, act: bool = False, gain: float = 1.0, clamp: float = 0.0,
                   add: Optional[torch.Tensor] = None, amax_out=None, act_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N,r,r,r,C] -> act(trilinear 2x (align_corners) + noise * ns + bias) + add, [N,2r,2r,2r,C]; act_mask: uint8 [N (2r)^3 C / 4] that
    receives the activation's branch per element (for layer_act_bwd)"""
    _need_cuda(x)
    x, add = _c(x), _c(add)
    N, r, r2, r3, Cc = x.shape
    assert r == r2 == r3, "cubic volumes"
    y = torch.empty((N, 2 * r, 2 * r, 2 * r, Cc), device=x.device, dtype=torch.float32)
    ep, keep = _epilogue(bias, noise, noise_strength, act, gain, clamp, amax_out)
    check(lib().asd_upsample3d_fwd(ptr(x), i32(N), i32(r), i32(Cc), C.byref(ep), ptr(add), ptr(y), ptr(act_mask), stream()))
    return y


def upsample3d_bwd(dy: torch.Tensor) -> torch.Tensor:
    dy = _c(dy)
    N, R, _, _, Cc = dy.shape
    r = R // 2
    dx = torch.empty((N, r, r, r, Cc), device=dy.device, dtype=torch.float32)
    ws = torch.empty(6 * N * r * r * r * Cc, device=dy.device, dtype=torch.float32)
    check(lib().asd_upsample3d_bwd(ptr(dy), i32(N), i32(r), i32(Cc), ptr(dx), ptr(ws), stream()))
    return dx


def torgb_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N,D,H,W,Cin], w [N,32,Cin] (per-sample modulated weights), bias [32] -> x w^T + bias (+ add) [N,D,H,W,32], exact fp32"""
    _need_cuda(x, w)
    x, w, bias, add = _c(x), _c(w), _c(bias), _c(add)
    N, Cin = x.shape[0], x.shape[-1]
    rows = x.numel() // Cin // N
    y = torch.empty((*x.shape[:-1], 32), device=x.device, dtype=torch.float32)
    for n in range(N):
        check(lib().asd_torgb_fwd(ptr(x[n]), C.c_int64(rows), i32(Cin), ptr(w[n]), ptr(bias), ptr(None if add is None else add[n]), ptr(y[n]),
                                  C.c_void_p(0), stream()))
    return y


def torgb_bwd(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor, need_dx: bool = True):
    """-> (dx [N,D,H,W,Cin] | None, dw [N,32,Cin], d_bias [32])"""
    x, dy, w = _c(x), _c(dy), _c(w)
    N, Cin = x.shape[0], x.shape[-1]
    rows = x.numel() // Cin // N
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((N, 32, Cin), device=x.device, dtype=torch.float32)
    db = torch.empty((N, 32), device=x.device, dtype=torch.float32)
    for n in range(N):
        check(lib().asd_torgb_bwd(ptr(x[n]), ptr(dy[n]), C.c_int64(rows), i32(Cin), ptr(w[n]), C.c_void_p(0), ptr(None if dx is None else dx[n]),
                                  ptr(dw[n]), ptr(db[n]), stream()))
    return dx, dw, db.sum(0)


# ---- fused sampled-volume field (3DConv-net): trilinear sample -> MLP heads -> bias -> finite differences in one kernel -----------------
def voxfield_fwd(voxel_cl: torch.Tensor, cfg: FieldCfg, w1s, w2s, w1f, w2f, points, want_normal: bool, want_features: bool = True,
                 save_enc: bool = True):
    """voxel_cl [D,H,W,32] (one batch entry), points [n,3] world coordinates -> (sdf [n], features [n,3] | None, normal, fd_grad, enc [n,32] | None)"""
    _need_cuda(voxel_cl, points)
    voxel_cl, points = _c(voxel_cl), _c(points)
    D, H, W, Cc = voxel_cl.shape
    n, dev = points.shape[0], points.device
    sdf = torch.empty(n, device=dev, dtype=torch.float32)
    feats = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_features and cfg.n_feature_dims == 3 else None
    normal = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_normal else None
    fdg = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_normal else None
    enc = torch.empty((n, 32), device=dev, dtype=torch.float32) if save_enc else None
    check(lib().asd_voxfield_fwd(ptr(voxel_cl), i32(D), i32(H), i32(W), i32(Cc), C.byref(cfg), ptr(w1s), ptr(w2s), ptr(w1f), ptr(w2f), ptr(points), i32(n),
                                 ptr(sdf), ptr(feats), ptr(normal), ptr(fdg), ptr(enc), stream()))
    return sdf, feats, normal, fdg, enc


def voxfield_bwd(voxel_cl, cfg: FieldCfg, w1s, w2s, w1f, w2f, points, enc, sdf, d_sdf, d_features, d_normal, d_fd_grad, d_voxel: torch.Tensor):
    """accumulates into d_voxel [D,H,W,32]; returns (dw1s, dw2s, dw1f, dw2f)"""
    voxel_cl, points = _c(voxel_cl), _c(points)
    D, H, W, Cc = voxel_cl.shape
    n, dev = points.shape[0], points.device
    nf = C.c_int64(0)
    check(lib().asd_voxfield_bwd_workspace(C.byref(cfg), i32(n), i32(int(d_normal is not None or d_fd_grad is not None)), C.byref(nf)))
    ws = torch.empty(nf.value, device=dev, dtype=torch.float32)
    dw1s = torch.zeros((64, 32), device=dev, dtype=torch.float32)
    dw2s = torch.zeros((1, 64), device=dev, dtype=torch.float32)
    dw1f = torch.zeros((64, 32), device=dev, dtype=torch.float32)
    dw2f = torch.zeros((3, 64), device=dev, dtype=torch.float32)
    k = _Keep()
    check(lib().asd_voxfield_bwd(ptr(voxel_cl), i32(D), i32(H), i32(W), i32(Cc), C.byref(cfg), ptr(w1s), ptr(w2s), ptr(w1f), ptr(w2f), ptr(points),
                                 ptr(enc), ptr(sdf), i32(n), k(d_sdf), k(d_features), k(d_normal), k(d_fd_grad), ptr(d_voxel), ptr(dw1s), ptr(dw2s),
                                 ptr(dw1f), ptr(dw2f), ptr(ws), stream()))
    return dw1s, dw2s, dw1f, dw2f


# ---- fused tri-plane field (Triplane-transformer-sdf): three plane lookups -> 96 -> 64 -> 64 -> 1 | 3 heads -> bias -> finite differences ------
def _ptr6(ts):
    return (C.c_void_p * 6)(*[t.data_ptr() for t in ts])


def trifield_fwd(planes_cl: torch.Tensor, cfg: FieldCfg, weights6, points, want_normal: bool, want_features: bool = True):
    """planes_cl [3,H,W,32] (one batch entry); weights6 = (sdf W1^T [96,64], W2 [64,64], W3 [1,64], feature W1^T, W2, W3 [3,64]) contiguous fp32;
    points [n,3] world coordinates -> (sdf [n], features [n,3] | None, normal, fd_grad)"""
    _need_cuda(planes_cl, points)
    planes_cl, points = _c(planes_cl), _c(points)
    _, H, W, Cc = planes_cl.shape
    n, dev = points.shape[0], points.device
    sdf = torch.empty(n, device=dev, dtype=torch.float32)
    feats = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_features else None
    normal = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_normal else None
    fdg = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_normal else None
    nf = C.c_int64(0)
    check(lib().asd_trifield_fwd_workspace(i32(H), i32(W), C.byref(nf)))
    ws = torch.empty(nf.value, device=dev, dtype=torch.float32)
    check(lib().asd_trifield_fwd(ptr(planes_cl), i32(H), i32(W), i32(Cc), C.byref(cfg), _ptr6(weights6), ptr(points), i32(n), ptr(sdf), ptr(feats),
                                 ptr(normal), ptr(fdg), ptr(ws), stream()))
    return sdf, feats, normal, fdg


def trifield_bwd(planes_cl, cfg: FieldCfg, weights6, points, sdf, d_sdf, d_features, d_normal, d_fd_grad, d_planes: torch.Tensor):
    """accumulates into d_planes [3,H,W,32]; returns the six weight gradients (first layers in the NATIVE [64,96] layout)"""
    planes_cl, points = _c(planes_cl), _c(points)
    _, H, W, Cc = planes_cl.shape
    n, dev = points.shape[0], points.device
    nf = C.c_int64(0)
    check(lib().asd_trifield_bwd_workspace(i32(H), i32(W), i32(n), i32(int(d_normal is not None or d_fd_grad is not None)), C.byref(nf)))
    ws = torch.empty(nf.value, device=dev, dtype=torch.float32)
    dws = [torch.zeros(sh, device=dev, dtype=torch.float32) for sh in ((64, 96), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))]
    k = _Keep()
    check(lib().asd_trifield_bwd(ptr(planes_cl), i32(H), i32(W), i32(Cc), C.byref(cfg), _ptr6(weights6), ptr(points), ptr(sdf), i32(n), k(d_sdf),
                                 k(d_features), k(d_normal), k(d_fd_grad), ptr(d_planes), _ptr6(dws), ptr(ws), stream()))
    return dws
