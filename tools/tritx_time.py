"""Timings of the tri-plane transformer's building blocks at the shipped shapes (csrc/tritx.hip): python tools/tritx_time.py
(HIP events around 10 back-to-back calls after 3 warm-up calls; whole C-ABI entry, i.e. including the operand-plane passes)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import test_gpu_tritx as T
from scaledreamer_amd import _lib as L


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


H, D = 16, 768
g = torch.Generator(device="cuda").manual_seed(0)
for name, Lq, Lk in (("self", 3072, 3072), ("cross", 3072, 77)):
    q = torch.randn(Lq, D, device="cuda", generator=g)
    k, v = torch.randn(Lk, D, device="cuda", generator=g), torch.randn(Lk, D, device="cuda", generator=g)
    d_o = torch.randn(Lq, D, device="cuda", generator=g)
    o, lse, ws = T.attention_fwd(q, k, v, H)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    P = lambda t: C.c_void_p(t.data_ptr())
    fwd = lambda: L.check(L.lib().asd_tx_attention_fwd(P(q), L.i32(D), P(k), L.i32(D), P(v), L.i32(D), L.i32(Lq), L.i32(Lk), L.i32(H), P(o), L.i32(D), P(lse), P(ws), L.stream()))
    bwd = lambda: L.check(L.lib().asd_tx_attention_bwd(P(q), L.i32(D), P(k), L.i32(D), P(v), L.i32(D), P(o), L.i32(D), P(d_o), L.i32(D), P(lse), L.i32(Lq), L.i32(Lk),
                                                       L.i32(H), P(dq), L.i32(D), P(dk), L.i32(D), P(dv), L.i32(D), P(ws), L.stream()))
    flops = 4.0 * Lq * Lk * D
    tf, tb = timed(fwd), timed(bwd)
    print(f"attention {name:5s} fwd {tf:8.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s fp32-equivalent)   bwd {tb:8.1f} us ({3.5 * flops / tb / 1e6:6.1f} TFLOP/s incl. recomputation)")
for M, K, N in ((3072, 768, 768), (3072, 768, 2304), (3072, 768, 3072), (3072, 3072, 768), (77, 1024, 1536)):
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    dy = torch.randn(M, N, device="cuda", generator=g)
    pw, iw, pt, it = T.pack_weight(w, True, True)
    y = torch.empty(M, N, device="cuda")
    ws = T._ws(L.lib().asd_tx_linear_workspace(L.i32(M), L.i32(N), L.i32(K)))
    lin = lambda: L.check(L.lib().asd_tx_linear(L.ptr(x), L.i32(M), L.i32(K), L.i32(K), L.ptr(pw), L.ptr(iw), L.i32(N), None, L.i32(0), None, None, L.i32(0), L.ptr(y), L.i32(N), L.ptr(ws), L.stream()))
    dw = torch.empty(N, K, device="cuda")
    ws2 = T._ws(L.lib().asd_tx_wgrad_workspace(L.i32(M), L.i32(N), L.i32(K)))
    wg = lambda: L.check(L.lib().asd_tx_linear_wgrad(L.ptr(dy), L.i32(N), L.ptr(x), L.i32(K), L.i32(M), L.i32(N), L.i32(K), L.ptr(dw), None, L.ptr(ws2), L.stream()))
    t1, t2, t3 = timed(lin), timed(wg), timed(lambda: x @ w.t())
    fl = 2.0 * M * N * K
    print(f"linear {M}x{K}->{N}: fwd {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF/s fp32-eq)  wgrad {t2:7.1f} us ({fl / t2 / 1e6:6.1f})  torch fp32 matmul {t3:7.1f} us ({fl / t3 / 1e6:6.1f})")

M, Dm = 3072, 768
x, dy, dres = (torch.randn(M, Dm, device="cuda", generator=g) for _ in range(3))
gamma, beta = torch.randn(Dm, device="cuda", generator=g), torch.randn(Dm, device="cuda", generator=g)
y, stats, dx = torch.empty_like(x), torch.empty(M, 2, device="cuda"), torch.empty_like(x)
dg, db = torch.zeros(Dm, device="cuda"), torch.zeros(Dm, device="cuda")
lnf = lambda: L.check(L.lib().asd_tx_layernorm_fwd(L.ptr(x), L.i32(M), L.i32(Dm), L.ptr(gamma), L.ptr(beta), L.f32(1e-6), L.ptr(y), L.ptr(stats), L.stream()))
lnb = lambda: L.check(L.lib().asd_tx_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(stats), L.ptr(gamma), L.i32(M), L.i32(Dm), L.ptr(dres), L.ptr(dx), L.ptr(dg), L.ptr(db), L.stream()))
print(f"layernorm {M}x{Dm}: fwd {timed(lnf):6.1f} us  bwd {timed(lnb):6.1f} us  (ASD_TX_LN_ROWS={os.environ.get('ASD_TX_LN_ROWS', '2')})")
