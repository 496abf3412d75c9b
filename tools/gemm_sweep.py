"""Sweep the tile configurations (and split-K) of asd_gemm_f16 over the GEMM shapes of one ASD step; prints the best
configuration per shape next to the cost model's choice.   python tools/gemm_sweep.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib
from scaledreamer_amd.diffusion import hip_ops as H

TILES = ["128x64", "128x128", "256x64", "256x128", "128x320", "256x256", "256x320", "320x128", "win64", "win128", "win64x2", "win128x2", "64x64"]
BN = [64, 128, 64, 128, 320, 256, 320, 128, 64, 128, 64, 128, 64]


def timeit(fn, reps=20):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e5 + 2.5e4 * reps))   # park the GPU while the launches are queued: GPU time, not the host's launch rate
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def r(*s):
    return torch.randn(*s, device="cuda").half()


import os
shapes = [("gemm", 1280, 1280, 1280), ("gemm", 1280, 1280, 5120), ("gemm", 1280, 10240, 1280), ("gemm", 20480, 320, 320), ("gemm", 5120, 640, 640), ("gemm", 5120, 640, 2560),
          ("gemm", 20480, 320, 1280), ("gemm", 320, 1280, 1280), ("gemm", 320, 1280, 5120), ("gemm", 5120, 1280, 640)] if os.environ.get("SWEEP_SMALL") else [("conv", 5, 64, 320, 320), ("conv", 5, 64, 640, 320), ("conv", 5, 64, 960, 320), ("conv", 5, 32, 640, 640), ("conv", 5, 32, 1280, 640),
          ("conv", 5, 16, 1280, 1280), ("conv", 5, 16, 2560, 1280), ("conv", 5, 8, 1280, 1280), ("conv", 5, 8, 2560, 1280),
          ("conv", 1, 512, 128, 128), ("conv", 1, 256, 256, 256), ("conv", 1, 128, 512, 512), ("conv", 1, 64, 512, 512),
          ("conv", 1, 64, 128, 128), ("conv", 1, 256, 128, 128), ("conv", 4, 32, 320, 320),
          ("gemm", 20480, 320, 320), ("gemm", 20480, 2560, 320), ("gemm", 20480, 320, 1280), ("gemm", 5120, 640, 640), ("gemm", 5120, 5120, 640),
          ("gemm", 5120, 640, 2560), ("gemm", 1280, 1280, 1280), ("gemm", 1280, 10240, 1280), ("gemm", 1280, 1280, 5120), ("gemm", 320, 1280, 1280),
          ("gemm", 320, 10240, 1280), ("gemm", 320, 1280, 5120), ("gemm", 400, 12480, 1024), ("gemm", 4096, 4096, 512), ("gemm", 4096, 512, 4096)]
for sh in shapes:
    if sh[0] == "conv":
        _, B, hw, cin, cout = sh
        x, w = r(B, hw, hw, cin), H.pack_conv3x3_weight(r(cout, cin, 3, 3))
        M, N, K = B * hw * hw, cout, 9 * cin
        run = lambda sk: H.conv3x3(x, w, split_k=sk)
    else:
        _, M, N, K = sh
        a, w = r(M, K), r(N, K)
        run = lambda sk: H.gemm(a, w, split_k=sk)
    lib().asd_gemm_force_tile(C.c_int32(-1))
    sk0 = H.default_split(M, N, K)
    base = timeit(lambda: run(sk0))
    res = []
    for t, name in enumerate(TILES):
        if N % BN[t] != 0 and not (BN[t] == 64):
            continue
        if 8 <= t <= 11 and (sh[0] != "conv" or sh[2] % 16 != 0 or sh[3] % 64 != 0):
            continue
        lib().asd_gemm_force_tile(C.c_int32(t))
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (K // sk < 256):
                continue
            if 8 <= t <= 11 and sk > sh[3] // 64:
                continue
            res.append((timeit(lambda: run(sk), reps=8), name, sk))
    lib().asd_gemm_force_tile(C.c_int32(-1))
    res.sort()
    fl = 2.0 * M * N * K
    top = "  ".join(f"{n}/s{k}:{u:.0f}" for u, n, k in res[:4])
    print(f"{str(sh):34s} model(split {sk0}) {base:7.1f} us {fl / base / 1e6:6.0f} TF/s | best {res[0][0]:7.1f} us {fl / res[0][0] / 1e6:6.0f} TF/s | {top}")
