"""Ping-pong window convolution (csrc/gemm_pp.hip, tile configurations 20-24) against the tuned plan and the other window kernels on the
3x3 stride-1 convolution shapes of one ASD step: error vs the implicit-GEMM kernel, time behind a spin kernel, PFLOP/s.
    python tools/pp_ab.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib, LIB_PATH
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


print("lib", LIB_PATH)
torch.manual_seed(0)
# (batch, H = W, Cin, Cout): VAE encoder levels, UNet levels at batch 5 / 2
shapes = [(1, 512, 128, 128), (1, 256, 128, 256), (1, 256, 256, 256), (1, 128, 256, 512), (1, 128, 512, 512), (1, 64, 512, 512),
          (5, 64, 320, 320), (2, 64, 320, 320), (5, 64, 640, 320), (5, 64, 960, 320), (5, 32, 320, 640), (5, 32, 640, 640), (5, 32, 1280, 640),
          (5, 32, 960, 640), (5, 16, 640, 1280), (5, 16, 1280, 1280), (5, 16, 2560, 1280)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = [(1, 512, 128, 128), (1, 256, 256, 256), (5, 64, 320, 320)]
names = {8: "win64", 9: "win128", 10: "win64x2", 11: "win128x2", 13: "w4/64", 14: "w4/128", 20: "pp512x128", 21: "pp256x256", 22: "pp256x320",
         23: "pp256x128", 24: "pp256x160"}
for B, hw, cin, cout in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * (9 * cin) ** -0.5)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    lib().asd_gemm_force_tile(C.c_int32(2))
    ref = H.conv3x3(x, w, split_k=1).float()
    lib().asd_gemm_force_tile(C.c_int32(-1))
    us = timeit(lambda: H.conv3x3(x, w))
    out = [f"plan: {us:6.1f} us {fl / us / 1e9:5.2f} PF"]
    for t in (9, 11, 14, 10, 13, 20, 21, 22, 23, 24):
        bn, bm = H.TILE_BN[t], H.TILE_BM[t]
        if cout % bn or (t in H.PP_TILES and hw % (bm // 16)):
            continue
        best = None
        for sk in (1, 2, 4, 5, 8, 10):
            if sk > 1 and (cin // 64 < 2 * sk or (B * hw * hw // bm) * (cout // bn) * sk > 1536):
                continue
            y = H.conv3x3(x, w, split_k=sk, tile_cfg=t + 1).float()
            err = float((y - ref).abs().max() / ref.abs().max())
            us = timeit(lambda: H.conv3x3(x, w, split_k=sk, tile_cfg=t + 1))
            if err > 2e-3:
                out.append(f"{names[t]}/s{sk}: WRONG err {err:.1e}")
            if best is None or us < best[0]:
                best = (us, sk, err)
        if best:
            out.append(f"{names[t]}/s{best[1]}: {best[0]:6.1f} us {fl / best[0] / 1e9:5.2f} PF e{best[2]:.0e}")
    print(f"{(B, hw, cin, cout)}: " + " | ".join(out), flush=True)
