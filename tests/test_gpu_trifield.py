"""The fused tri-plane field (include/asd_hip.h: asd_trifield_fwd / _bwd; csrc/trifield_mfma.hip: split-fp16 products on the matrix pipe with the
probes of the finite-difference normal carried as differences) straight at the C ABI against a float64 restatement of
Triplane-transformer-sdf.forward (custom/amortized/models/geometry/triplane_transformer.py:139-240): F.grid_sample lookups, two VanillaMLP
heads, sphere bias, finite-difference sdf_grad / normal, and every gradient - planes and the six head weights.

Tolerances.  Outputs: fp32 class (1e-5 of the norm).  Gradients: a ReLU whose pre-activation lies within fp32 rounding of zero switches a whole
row's contribution on or off against float64 (one row of n is n^-1/2 of a random-sign sum: 2e-2 at n = 3001), in ANY fp32 evaluation; the
seeds below have no such row at n = 3001 (1e-4 bound, observed 2e-6), the larger sizes are bounded by what one or two flips cost."""
import ctypes as C
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GUARD = 1 << 16


def _cfg():
    from scaledreamer_amd import _lib

    f = _lib.FieldCfg()
    for d in range(3):
        f.bbox_min[d], f.bbox_max[d] = -2.0, 2.0
    f.radius, f.bias_mode, f.bias_value = 2.0, _lib.ASD_BIAS_SPHERE, 0.8
    f.blob_scale, f.blob_std, f.activation = 0.0, 1.0, _lib.ASD_ACT_NONE
    f.fd_eps, f.n_hidden, f.n_feature_dims, f.field_mode = 0.01, 64, 3, _lib.ASD_FIELD_SDF
    return f


def _ref64(planes_cl, ws, pts, gs=None):
    c = planes_cl.double().permute(0, 3, 1, 2)[None].clone().requires_grad_(gs is not None)
    w = [x.double().clone().requires_grad_(gs is not None) for x in ws]

    def enc(p):
        u = p.double()[None] / 2.0
        proj = [u[..., [0, 1]], u[..., [0, 2]], u[..., [2, 1]]]
        return torch.cat([F.grid_sample(c[:, k], proj[k][:, None], mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, 0].t() for k in range(3)], -1)

    def sdf_of(p):
        return torch.relu(torch.relu(enc(p) @ w[0].t()) @ w[1].t()) @ w[2].t() + (p.double().pow(2).sum(-1, keepdim=True).sqrt() - 0.8)

    s = sdf_of(pts)
    f = torch.relu(torch.relu(enc(pts) @ w[3].t()) @ w[4].t()) @ w[5].t()
    sg = torch.cat([(sdf_of((pts + 0.01 * torch.eye(3, device=pts.device)[k]).clamp(-2.0, 2.0)) - s) / 0.01 for k in range(3)], -1)
    out = {"sdf": s[:, 0], "features": f, "sdf_grad": sg, "normal": F.normalize(sg, dim=-1)}
    if gs is None:
        return out
    sum((out[k] * gs[k].double()).sum() for k in gs).backward()
    return out, c.grad[0].permute(0, 2, 3, 1), [x.grad for x in w]


def _problem(n, seed=3, wscale=2.0, hw=(64, 64)):
    g = torch.Generator().manual_seed(seed)
    planes = (torch.randn(3, hw[0], hw[1], 32, generator=g) * 0.5).cuda()
    ws = [(torch.randn(o, i, generator=g) * (wscale / i) ** 0.5).cuda() for o, i in ((64, 96), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))]
    w6 = (ws[0].t().contiguous(), ws[1], ws[2], ws[3].t().contiguous(), ws[4], ws[5])
    pts = (torch.rand(n, 3, generator=g) * 4.4 - 2.2).cuda()               # some points outside the box: zero padding, clamped probes
    gs = {k: torch.randn(n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}
    gs["sdf"] = gs["sdf"][:, 0].contiguous()
    return planes, ws, w6, pts, gs


_l2 = lambda a, b: float((a.detach().double() - b.detach()).norm() / b.detach().norm().clamp_min(1e-30))


@pytest.mark.parametrize("n", [1, 15, 3001, 50001])
def test_forward_against_float64(n):
    from scaledreamer_amd import ops

    planes, ws, w6, pts, _ = _problem(n)
    r = _ref64(planes, ws, pts)
    for want_normal in (True, False):
        sdf, feats, normal, fdg = ops.trifield_fwd(planes, _cfg(), w6, pts, want_normal, True)
        assert _l2(sdf, r["sdf"]) < 1e-5 and _l2(feats, r["features"]) < 1e-5
        if want_normal:
            assert _l2(fdg, r["sdf_grad"]) < 5e-5 and _l2(normal, r["normal"]) < 5e-5      # (a difference of two fp32 values over eps = 0.01)
    assert _l2(ops.trifield_fwd(planes, _cfg(), w6, pts, False, False)[0], r["sdf"]) < 1e-5


@pytest.mark.parametrize("n,tol", [(15, 1e-4), (3001, 1e-4), (50001, 5e-3)])
@pytest.mark.parametrize("mode", ["all", "no_normal", "sdf_only"])
def test_backward_against_float64(n, tol, mode):
    from scaledreamer_amd import ops

    planes, ws, w6, pts, gs = _problem(n)
    keys = {"all": ("sdf", "features", "normal", "sdf_grad"), "no_normal": ("sdf", "features"), "sdf_only": ("sdf",)}[mode]
    g = {k: gs[k] for k in keys}
    sdf = ops.trifield_fwd(planes, _cfg(), w6, pts, True, True)[0]
    dpl = torch.zeros_like(planes)
    dws = ops.trifield_bwd(planes, _cfg(), w6, pts, sdf, g.get("sdf"), g.get("features"), g.get("normal"), g.get("sdf_grad"), dpl)
    _, cref, wref = _ref64(planes, ws, pts, g)
    assert _l2(dpl, cref) < tol, f"planes {_l2(dpl, cref):.1e}"
    for i, (a, b) in enumerate(zip(dws, wref)):
        if b is None or float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0
        else:
            assert _l2(a, b) < tol, f"dW{i} {_l2(a, b):.1e}"


def test_rectangular_planes():
    """H != W (the reference's planes are square; the C ABI is not): lookups, scatter bins and the sort's cell keys"""
    from scaledreamer_amd import ops

    planes, ws, w6, pts, gs = _problem(3001, seed=3, hw=(40, 56))       # (seeds 5, 12, 13 have a ReLU pre-activation within rounding of zero: 1e-3 .. 7e-3 in BOTH fp32 forms)
    sdf, feats, normal, fdg = ops.trifield_fwd(planes, _cfg(), w6, pts, True, True)
    r, cref, wref = _ref64(planes, ws, pts, gs)
    assert _l2(sdf, r["sdf"]) < 1e-5 and _l2(feats, r["features"]) < 1e-5 and _l2(fdg, r["sdf_grad"]) < 5e-5
    dpl = torch.zeros_like(planes)
    dws = ops.trifield_bwd(planes, _cfg(), w6, pts, sdf, gs["sdf"], gs["features"], gs["normal"], gs["sdf_grad"], dpl)
    assert _l2(dpl, cref) < 2e-4
    for a, b in zip(dws, wref):
        assert _l2(a, b) < 2e-4


def test_backward_in_chunks(monkeypatch):
    """the pass walks the samples in chunks (1 M by default): three chunks of 1 100 samples, the last one ragged, against one chunk"""
    from scaledreamer_amd import ops

    planes, ws, w6, pts, gs = _problem(3001)
    sdf = ops.trifield_fwd(planes, _cfg(), w6, pts, True, True)[0]
    res = []
    for chunk in (None, "1100"):
        if chunk:
            monkeypatch.setenv("ASD_TRI_CHUNK", chunk)
        dpl = torch.zeros_like(planes)
        dws = ops.trifield_bwd(planes, _cfg(), w6, pts, sdf, gs["sdf"], gs["features"], gs["normal"], gs["sdf_grad"], dpl)
        res.append([dpl] + list(dws))
    for a, b in zip(*res):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5


def test_module_default_initialisation_scale():
    """weights of the size torch's Linear init draws (+-0.1): the norm-bound scales of the split must not cost accuracy at that end either"""
    from scaledreamer_amd import ops

    planes, ws, w6, pts, gs = _problem(3001, seed=9, wscale=0.3)
    sdf, feats, normal, fdg = ops.trifield_fwd(planes, _cfg(), w6, pts, True, True)
    r, cref, wref = _ref64(planes, ws, pts, gs)
    assert _l2(sdf, r["sdf"]) < 1e-5 and _l2(feats, r["features"]) < 1e-5 and _l2(fdg, r["sdf_grad"]) < 5e-5
    dpl = torch.zeros_like(planes)
    dws = ops.trifield_bwd(planes, _cfg(), w6, pts, sdf, gs["sdf"], gs["features"], gs["normal"], gs["sdf_grad"], dpl)
    assert _l2(dpl, cref) < 2e-4
    for a, b in zip(dws, wref):
        assert _l2(a, b) < 2e-4


def test_repeatable_up_to_the_order_of_atomic_sums():
    from scaledreamer_amd import ops

    planes, ws, w6, pts, gs = _problem(20011, seed=4)
    ref = None
    for it in range(6):
        sdf, feats, normal, fdg = ops.trifield_fwd(planes, _cfg(), w6, pts, True, True)
        dpl = torch.zeros_like(planes)
        dws = ops.trifield_bwd(planes, _cfg(), w6, pts, sdf, gs["sdf"], gs["features"], gs["normal"], gs["sdf_grad"], dpl)
        cur = [sdf, feats, normal, fdg, dpl] + list(dws)
        if ref is None:
            ref = [t.clone() for t in cur]
            continue
        for k, (a, b) in enumerate(zip(cur, ref)):
            if k < 4:
                assert torch.equal(a, b), "the forward pass has no atomics: bit-identical"
            else:
                assert float((a - b).abs().max() / b.abs().max()) < 2e-5


def test_passes_stay_inside_their_buffers():
    from scaledreamer_amd import _lib

    planes, ws, w6, pts, gs = _problem(3001)
    n, cfg, L = pts.shape[0], _cfg(), _lib.lib()

    def guarded(nbytes):
        buf = torch.full((nbytes + 2 * GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
        return buf, buf[GUARD:GUARD + nbytes]

    intact = lambda buf, nbytes: bool((buf[:GUARD] == 0xA5).all()) and bool((buf[GUARD + nbytes:] == 0xA5).all())
    ptr6 = lambda ts: (C.c_void_p * 6)(*[t.data_ptr() for t in ts])
    nf = C.c_int64(0)
    _lib.check(L.asd_trifield_fwd_workspace(_lib.i32(64), _lib.i32(64), C.byref(nf)))
    bufs = {k: guarded(sz) for k, sz in (("ws", nf.value * 4), ("sdf", n * 4), ("feat", n * 12), ("normal", n * 12), ("fdg", n * 12))}
    p = lambda k: C.c_void_p(bufs[k][1].data_ptr())
    _lib.check(L.asd_trifield_fwd(_lib.ptr(planes), _lib.i32(64), _lib.i32(64), _lib.i32(32), C.byref(cfg), ptr6(w6), _lib.ptr(pts), _lib.i32(n), p("sdf"),
                                  p("feat"), p("normal"), p("fdg"), p("ws"), _lib.stream()))
    torch.cuda.synchronize()
    for k, (buf, view) in bufs.items():
        assert intact(buf, view.numel()), f"forward wrote outside {k}"
    sdf = bufs["sdf"][1].view(torch.float32).clone()
    _lib.check(L.asd_trifield_bwd_workspace(_lib.i32(64), _lib.i32(64), _lib.i32(n), _lib.i32(1), C.byref(nf)))
    shapes = ((64, 96), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))
    b2 = {"ws": guarded(nf.value * 4), "dpl": guarded(planes.numel() * 4)}
    b2.update({f"dw{i}": guarded(a * b * 4) for i, (a, b) in enumerate(shapes)})
    for k in b2:
        if k != "ws":
            b2[k][1].zero_()
    dws = [b2[f"dw{i}"][1].view(torch.float32) for i in range(6)]
    _lib.check(L.asd_trifield_bwd(_lib.ptr(planes), _lib.i32(64), _lib.i32(64), _lib.i32(32), C.byref(cfg), ptr6(w6), _lib.ptr(pts), _lib.ptr(sdf), _lib.i32(n),
                                  _lib.ptr(gs["sdf"]), _lib.ptr(gs["features"]), _lib.ptr(gs["normal"]), _lib.ptr(gs["sdf_grad"]),
                                  C.c_void_p(b2["dpl"][1].data_ptr()), ptr6(dws), C.c_void_p(b2["ws"][1].data_ptr()), _lib.stream()))
    torch.cuda.synchronize()
    for k, (buf, view) in b2.items():
        assert intact(buf, view.numel()), f"backward wrote outside {k}"
    assert all(torch.isfinite(d).all() for d in dws) and torch.isfinite(b2["dpl"][1].view(torch.float32)).all()
