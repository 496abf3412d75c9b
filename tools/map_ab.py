"""Plain GEMM timings over tile configurations for the library named by ASD_HIP_LIB (A/B of block -> tile mappings)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib, LIB_PATH
from scaledreamer_amd.diffusion import hip_ops as H

TILES = {0: "128x64", 1: "128x128", 2: "256x64", 3: "256x128", 4: "128x320", 5: "256x256", 6: "256x320", 7: "320x128"}
BN = {0: 64, 1: 128, 2: 64, 3: 128, 4: 320, 5: 256, 6: 320, 7: 128}


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e5 + 2.5e4 * reps))   # park the GPU while the launches are queued: GPU time, not the host's launch rate
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("lib", LIB_PATH)
for (M, N, K) in [(8192, 8192, 8192), (20480, 320, 320), (20480, 2560, 320), (20480, 320, 1280), (5120, 640, 640), (5120, 5120, 640), (5120, 640, 2560), (1280, 1280, 1280), (1280, 10240, 1280), (16384, 4096, 1024)]:
    a, w = torch.randn(M, K, device="cuda").half(), torch.randn(N, K, device="cuda").half()
    out = []
    for t, name in TILES.items():
        if N % BN[t]:
            continue
        lib().asd_gemm_force_tile(C.c_int32(t))
        us = timeit(lambda: H.gemm(a, w, split_k=1))
        out.append(f"{name}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s")
    lib().asd_gemm_force_tile(C.c_int32(-1))
    print(f"{(M, N, K)}: " + " | ".join(out), flush=True)
