"""GEGLU projections (act = 2) of the three transformer widths under every admissible tile: time with cold weights / activations.
    python tools/geglu_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H
from small_m_conv import timeit

torch.manual_seed(0)
for M, C in [(20480, 320), (5120, 640), (1280, 1280)]:
    N, K = 8 * C, C
    pool = 6
    a = [torch.randn(M, K, device="cuda").half() for _ in range(pool)]
    w = [(torch.randn(N, K, device="cuda") * K ** -0.5).half() for _ in range(pool)]
    b = torch.randn(N, device="cuda").half()
    out = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
    plan = timeit(lambda i: H.gemm(a[i % pool], w[i % pool], bias=b, act=2, out=out))
    rows = []
    for t in range(20):
        try:
            us = timeit(lambda i: H.gemm(a[i % pool], w[i % pool], bias=b, act=2, out=out, tile_cfg=t + 1, split_k=1), reps=12)
        except Exception as e:
            continue
        rows.append((us, t))
    rows.sort()
    print((M, N, K), f"plan {plan:.1f} us | " + " | ".join(f"cfg{t} {H.TILE_BM[t]}x{H.TILE_BN[t]}: {us:.1f}" for us, t in rows[:8]), flush=True)
