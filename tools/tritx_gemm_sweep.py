"""tile configuration x split-K sweep of asd_gemm_f16 (fp32 result) over the K-concatenated split-fp16 products of the tri-plane transformer
(csrc/tritx.hip: tx_gemm): prints the best plan per shape.   python tools/tritx_gemm_sweep.py   (GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scaledreamer_amd import _lib as L

TILES = ["128x64", "128x128", "256x64", "256x128", "128x320", "256x256", "256x320", "320x128"]
BN = [64, 128, 64, 128, 320, 256, 320, 128]
SHAPES = [(3072, 768, 2304), (3072, 2304, 2304), (3072, 3072, 2304), (3072, 768, 9216), (3072, 768, 6912), (77, 1536, 3072), (3072, 128, 2304), (3072, 768, 384),
          (768, 768, 9216), (2304, 768, 9216), (3072, 768, 9216), (768, 3072, 9216), (1536, 1024, 384), (768, 128, 9216)]
zero = torch.zeros(64, device="cuda")


def run(a, w, c, ws, M, N, K, cfg, sk):
    g = L.GemmArgs()
    g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), c.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldw, g.ldc = M, N, K, K, K, N
    g.out_f32, g.split_k, g.tile_cfg = 1, sk, cfg + 1
    g.zero_page, g.workspace = zero.data_ptr(), ws.data_ptr()
    return L.lib().asd_gemm_f16(C.byref(g), L.stream())


def timeit(fn, reps=10):
    for _ in range(2):
        if fn() != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, N, K in SHAPES:
    a, w = torch.randn(M, K, device="cuda").half(), torch.randn(N, K, device="cuda").half()
    c, ws = torch.empty(M, N, device="cuda"), torch.empty(8 * M * N, device="cuda")
    res = []
    for t in range(len(TILES)):
        if N % BN[t] != 0 and BN[t] not in (64, 128):
            continue
        for sk in (1, 2, 3, 4, 6, 8):
            if K % (64 * sk) != 0:
                continue
            us = timeit(lambda: run(a, w, c, ws, M, N, K, t, sk))
            if us is not None:
                res.append((us, TILES[t], t + 1, sk))
    res.sort()
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}: " + "  ".join(f"{n} sk{sk} {us:6.1f}us ({fl / us / 1e6:5.0f}TF)" for us, n, _, sk in res[:4]) + f"   PLAN {{{M}, {N}, {K}, {res[0][2]}, {res[0][3]}}},")
