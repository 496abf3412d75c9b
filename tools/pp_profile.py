"""Cycle accounting inside the ping-pong window convolution (conv3x3_pp_kernel built with -DASD_PP_PROFILE): per wave, s_memtime ticks
spent in the prologue, the load/read segments, at the two barriers of a phase, in the MFMA segments and in the epilogue.
    python tools/pp_profile.py   (GPU box; builds tools/bin/libasd_ppprof.so first)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "tools", "bin", "libasd_ppprof.so")
if not os.environ.get("ASD_HIP_LIB"):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "ppprof", "-DASD_PP_PROFILE", "gemm_pp.hip"])
    os.environ["ASD_HIP_LIB"] = os.path.join(ROOT, "scaledreamer_amd", "variants", "libasd_hip_ppprof.so")
    os.execv(sys.executable, [sys.executable] + sys.argv)
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from scaledreamer_amd._lib import GemmArgs, check, lib, stream
from scaledreamer_amd.diffusion import hip_ops as H

dev = torch.device("cuda", 0)
shapes = [(1, 512, 128, 128, 20), (1, 512, 128, 128, 23), (1, 256, 256, 256, 21), (1, 256, 256, 256, 23), (5, 64, 320, 320, 24), (5, 64, 640, 320, 22)]
for (B, hw, cin, cout, cfg) in shapes:
    x = torch.randn(B, hw, hw, cin, device=dev).half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device=dev).half() * 0.02)
    y = torch.empty(B * hw * hw, cout, device=dev, dtype=torch.float16)
    bn, bm = H.TILE_BN[cfg], H.TILE_BM[cfg]
    items = (B * hw * hw // bm) * (cout // bn)
    ws = torch.zeros(items * 8 * 16, device=dev, dtype=torch.int64)
    g = GemmArgs()
    g.A, g.W, g.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
    g.M, g.N, g.K = B * hw * hw, cout, 9 * cin
    g.lda, g.ldw, g.ldc = 0, 9 * cin, cout
    g.rows_per_group = 1
    g.conv, g.Hin, g.Win, g.Cin, g.Hout, g.Wout, g.stride, g.pad, g.upsample = 1, hw, hw, cin, hw, hw, 1, 1, 0
    g.zero_page = H.zero_page(dev).data_ptr()
    g.tile_cfg, g.split_k = cfg + 1, 1
    g.workspace = ws.data_ptr()
    for _ in range(3):
        check(lib().asd_gemm_f16(C.byref(g), stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); check(lib().asd_gemm_f16(C.byref(g), stream())); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = ws.cpu().numpy().reshape(items, 8, 16).astype(np.float64)
    wall0, wall1 = t[..., 8], t[..., 9]
    start, end, pro, tl, b1, tm, b2, epi = (t[..., i] for i in range(8))
    # wall_clock64 is a chip-wide constant 100 MHz counter: shader clock during the kernel = s_memtime ticks per wall tick, kernel span
    # and the number of blocks alive at a time from the wall stamps
    mhz = float(np.median((end - start)[:, 0] / np.maximum(wall1 - wall0, 1)[:, 0])) * 100.0
    span = (wall1.max() - wall0.min()) * (mhz / 100.0)
    conc = float((wall1 - wall0)[:, 0].sum() / (wall1.max() - wall0.min()))
    nph = (cin // 32) * 9 * (2 if (bm // 64) * (bn // 32) >= 24 else 1)
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(f"{B}x{hw}^2 {cin}->{cout} cfg{cfg} {grp}: {us:.1f} us, span {span:.0f} shader cycles at {mhz:.0f} MHz (s_memtime ticks per wall_clock64 tick), "
              f"{len(t)} blocks, {conc:.0f} alive on average; per wave: life {(end - start)[:, sl].mean():.0f} "
              f"prologue {pro[:, sl].mean():.0f} epilogue {epi[:, sl].mean():.0f}; per phase ({nph}): load/read {tl[:, sl].mean() / nph:.0f}  barrier1 {b1[:, sl].mean() / nph:.0f}  "
              f"mfma {tm[:, sl].mean() / nph:.0f}  barrier2 {b2[:, sl].mean() / nph:.0f}", flush=True)
