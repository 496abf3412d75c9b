"""Time the LDS-window 3x3 convolution (forced tile) on the conv shapes of one ASD step for the library named by ASD_HIP_LIB
(A/B of kernel variants behind the same C ABI) and check it against the implicit-GEMM kernel.   python tools/win_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib, LIB_PATH
from scaledreamer_amd.diffusion import hip_ops as H


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("lib", LIB_PATH)
torch.manual_seed(0)
shapes = [(1, 512, 128, 128), (1, 256, 256, 256), (1, 256, 128, 256), (1, 128, 512, 512), (1, 64, 512, 512), (5, 64, 320, 320), (5, 64, 640, 320),
          (5, 64, 960, 320), (5, 32, 640, 640), (5, 32, 1280, 640), (5, 16, 1280, 1280), (5, 16, 2560, 1280)]
for B, hw, cin, cout in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    lib().asd_gemm_force_tile(C.c_int32(2))
    ref = H.conv3x3(x, w, split_k=1).float()
    out = []
    for t, name in ((8, "win64"), (9, "win128"), (10, "win64x2"), (11, "win128x2")):
        if t in (9, 11) and cout % 128:
            continue
        lib().asd_gemm_force_tile(C.c_int32(t))
        for sk in (1, 2, 4, 5, 8):
            if sk > cin // 64 or (cin // 64) % sk:
                continue
            y = H.conv3x3(x, w, split_k=sk).float()
            err = float((y - ref).abs().max() / ref.abs().max())
            us = timeit(lambda: H.conv3x3(x, w, split_k=sk))
            out.append(f"{name}/s{sk}: {us:6.1f} us {fl / us / 1e6:5.0f} TF/s err {err:.1e}")
    lib().asd_gemm_force_tile(C.c_int32(-1))
    print(f"{(B, hw, cin, cout)}: " + " | ".join(out), flush=True)
