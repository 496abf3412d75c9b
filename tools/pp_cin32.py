"""VAE conv_in (3 -> 128 channels at 512^2, input padded to 32 channels) on the ping-pong window kernel (one 32-channel chunk) against the
implicit-GEMM plan: error and time.    python tools/pp_cin32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib
from scaledreamer_amd.diffusion import hip_ops as H


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda._sleep(200000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


torch.manual_seed(0)
for B, hw, cin, cout in [(1, 512, 32, 128), (4, 256, 32, 128)]:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    x[..., 3:] = 0
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 27 ** -0.5)
    bias = torch.randn(cout, device="cuda").half()
    lib().asd_gemm_force_tile(C.c_int32(2))
    ref = H.conv3x3(x, w, bias=bias, split_k=1).float()
    lib().asd_gemm_force_tile(C.c_int32(-1))
    out = [f"plan {timeit(lambda: H.conv3x3(x, w, bias=bias)):6.1f} us"]
    for t in (0, 1, 2, 3, 13, 20, 23):
        if t == 13:
            continue
        y = H.conv3x3(x, w, bias=bias, split_k=1, tile_cfg=t + 1).float()
        err = float((y - ref).abs().max() / ref.abs().max())
        us = timeit(lambda: H.conv3x3(x, w, bias=bias, split_k=1, tile_cfg=t + 1))
        out.append(f"cfg{t}: {us:6.1f} us e{err:.0e}")
    print((B, hw, cin, cout), " | ".join(out), flush=True)
