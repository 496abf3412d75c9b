"""Chosen tile configurations against the tuned plan on plain-GEMM shapes of the step (back-to-back launches behind a spin kernel).
    python tools/tile_ab.py 25,26"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda._sleep(200000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "25,26").split(",")]
torch.manual_seed(0)
shapes = [(20480, 320, 320, 1), (20480, 320, 320, 0), (20480, 320, 1280, 1), (20480, 640, 320, 0), (8192, 320, 320, 1), (8192, 640, 320, 0), (20480, 320, 640, 0),
          (20480, 320, 960, 0), (5120, 640, 640, 1), (5120, 640, 2560, 1), (5120, 1280, 640, 0), (1280, 1280, 1280, 1)]
for M, N, K, res in shapes:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda").half()
    r = torch.randn(M, N, device="cuda").half() if res else None
    fl = 2.0 * M * N * K
    ref = H.gemm(a, w, bias=b, residual=r)
    us = timeit(lambda: H.gemm(a, w, bias=b, residual=r))
    out = [f"plan {H.plan_table().get((M, N, K, K))}: {us:6.1f} us"]
    for t in tiles:
        if N % H.TILE_BN[t] and H.TILE_BN[t] != 64:
            continue
        y = H.gemm(a, w, bias=b, residual=r, tile_cfg=t + 1, split_k=1)
        err = float((y.float() - ref.float()).abs().max())
        us = timeit(lambda: H.gemm(a, w, bias=b, residual=r, tile_cfg=t + 1, split_k=1))
        out.append(f"tile{t} {H.TILE_BM[t]}x{H.TILE_BN[t]}: {us:6.1f} us d{err:.0e}")
    print(f"{(M, N, K)} res={res}: " + " | ".join(out), flush=True)
