"""GPU parity of the tri-plane transformer's building blocks (csrc/tritx.hip), straight at the C ABI, against float64 restatements of the
reference's torch ops (custom/amortized/extern/triplane_transformer_modules.py:34-187: nn.Linear, nn.GELU, nn.LayerNorm, diffusers'
Attention = softmax(q k^T / sqrt(d)) v) — the reference trains this generator in fp32, so the bar is fp32-class: errors are quoted
relative to the output range and compared with what torch's own fp32 ops leave on the same problem."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from scaledreamer_amd import _lib as L

    return L


def _ws(n_floats):
    return torch.empty(int(n_floats), device="cuda", dtype=torch.float32)


def pack_weight(w, fwd=True, bwd=False):
    """(plane_w, inv_w, plane_wt, inv_wt) of an nn.Linear weight [N, K]"""
    L = _lib()
    N, K = w.shape
    Np = (N + 63) // 64 * 64
    pw = torch.empty((N, 3 * K), device="cuda", dtype=torch.float16) if fwd else None
    iw = torch.empty(N, device="cuda") if fwd else None
    pt = torch.empty((K, 3 * Np), device="cuda", dtype=torch.float16) if bwd else None
    it = torch.empty(K, device="cuda") if bwd else None
    ws = _ws(max(K, N) + 64)
    L.check(L.lib().asd_tx_pack_weight(L.ptr(w.contiguous()), L.i32(N), L.i32(K), L.ptr(pw), L.ptr(iw), L.ptr(pt), L.ptr(it), L.ptr(ws), L.stream()))
    return pw, iw, pt, it


def linear(x, plane, inv, N, bias=None, mode=0, aux=None, residual=None):
    L = _lib()
    M, K = x.shape
    y = torch.empty((M, N), device="cuda")
    ws = _ws(L.lib().asd_tx_linear_workspace(L.i32(M), L.i32(N), L.i32(K)))
    L.check(L.lib().asd_tx_linear(L.ptr(x), L.i32(M), L.i32(K), L.i32(K), L.ptr(plane), L.ptr(inv), L.i32(N), L.ptr(bias), L.i32(mode), L.ptr(aux),
                                  L.ptr(residual), L.i32(N), L.ptr(y), L.i32(N), L.ptr(ws), L.stream()))
    return y


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("M,K,N", [(3072, 768, 768), (3072, 768, 3072), (77, 1024, 1536), (3072, 3072, 768), (200, 64, 128)])
def test_linear_forward_gelu_and_input_gradient(M, K, N):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g) * 3       # rows of uneven magnitude
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g)
    pw, iw, pt, it = pack_weight(w, True, N % 64 == 0)
    ref = x.double() @ w.double().t() + b.double()
    y = linear(x, pw, iw, N, bias=b, residual=res)
    e_hip, e_torch = rel(y, ref + res.double()), rel(x @ w.t() + b + res, ref + res.double())
    assert e_hip < 4e-6, (e_hip, e_torch)
    # GELU epilogue + saved pre-activation
    aux = torch.empty(M, N, device="cuda")
    h = linear(x, pw, iw, N, bias=b, mode=1, aux=aux)
    assert rel(aux, ref) < 4e-6 and rel(h, torch.nn.functional.gelu(ref)) < 4e-6
    if N % 64 == 0:
        # input gradient through the GELU: dx = (dh * gelu'(u)) w  — as the transformer's backward does it: mode 2 on the PREVIOUS layer's
        # product, here simply dx = dy w with the transposed planes, and the GELU' factor on a second product
        dy = torch.randn(M, N, device="cuda", generator=g)
        dx = linear(dy, pt, it, K)
        assert rel(dx, dy.double() @ w.double()) < 4e-6      # 22-bit operands, up to 3072 terms per sum
        u = torch.randn(M, K, device="cuda", generator=g)
        dxg = linear(dy, pt, it, K, mode=2, aux=u)
        ud = u.double()
        gp = 0.5 * (1 + torch.erf(ud / 2 ** 0.5)) + ud * torch.exp(-0.5 * ud * ud) / (2 * np.pi) ** 0.5
        assert rel(dxg, (dy.double() @ w.double()) * gp) < 4e-6


@pytest.mark.parametrize("M,N,K", [(3072, 768, 768), (3072, 2304, 768), (77, 1536, 1024), (3072, 768, 3072), (130, 64, 128)])
def test_linear_weight_gradient(M, N, K):
    L = _lib()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    dy = torch.randn(M, N, device="cuda", generator=g) * torch.logspace(-4, 0, N, device="cuda")       # columns spanning four decades
    x = torch.randn(M, K, device="cuda", generator=g) * (1 + 5 * torch.rand(K, device="cuda", generator=g))
    dw, db = torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")
    ws = _ws(L.lib().asd_tx_wgrad_workspace(L.i32(M), L.i32(N), L.i32(K)))
    L.check(L.lib().asd_tx_linear_wgrad(L.ptr(dy), L.i32(N), L.ptr(x), L.i32(K), L.i32(M), L.i32(N), L.i32(K), L.ptr(dw), L.ptr(db), L.ptr(ws), L.stream()))
    ref = dy.double().t() @ x.double()
    # per output row (= column of dy, each with its own scale): relative to that row's range
    err = float(((dw.double() - ref).abs().amax(1) / ref.abs().amax(1)).max())
    err_t = float((((dy.t() @ x).double() - ref).abs().amax(1) / ref.abs().amax(1)).max())
    assert err < 5e-6, (err, err_t)
    assert rel(db, dy.double().sum(0)) < 1e-5


def test_layernorm_forward_backward():
    L = _lib()
    M, D = 3072, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, D, device="cuda", generator=g) * 2 + 0.3
    gamma, beta = torch.randn(D, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
    y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
    L.check(L.lib().asd_tx_layernorm_fwd(L.ptr(x), L.i32(M), L.i32(D), L.ptr(gamma), L.ptr(beta), L.f32(1e-6), L.ptr(y), L.ptr(stats), L.stream()))
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-6)
    assert rel(y, ref.detach()) < 1e-6
    dy, dres = torch.randn(M, D, device="cuda", generator=g), torch.randn(M, D, device="cuda", generator=g)
    ref.backward(dy.double())
    dx, dg, db = torch.empty_like(x), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    L.check(L.lib().asd_tx_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(gamma), L.i32(M), L.i32(D), L.ptr(dres), L.ptr(dx), L.ptr(dg), L.ptr(db), L.stream()))
    assert rel(dx, xd.grad + dres.double()) < 2e-6
    assert rel(dg, gd.grad) < 1e-5 and rel(db, bd.grad) < 1e-5
