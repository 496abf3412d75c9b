"""Plain GEMM with the shape of the 8x8-level convolutions (M = 320, N = 1280, K = 11520) and cold weights: is the conv's im2col addressing
what the 28 us are spent on?   python tools/small_m_plain.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H
from small_m_conv import timeit

for M, N, K in [(320, 1280, 11520), (320, 1280, 2560), (1280, 1280, 1280)]:
    pool = int(400e6 // (N * K * 2)) + 1
    a = torch.randn(M, K, device="cuda").half()
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).half() for _ in range(pool)]
    rows = []
    for t in (0, 1, 2, 7, 12, 15, 16, 17, 18, 19):
        for sk in (1, 2, 4, 5, 8, 10, 16):
            try:
                us = timeit(lambda i: H.gemm(a, ws[i % pool], tile_cfg=t + 1, split_k=sk), reps=12)
            except Exception:
                continue
            rows.append((us, t, sk))
    rows.sort()
    print((M, N, K), "best: " + " | ".join(f"cfg{t} s{sk}: {us:.1f}" for us, t, sk in rows[:8]), flush=True)
