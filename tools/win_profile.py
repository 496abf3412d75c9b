"""Cycle accounting inside the window convolution (conv3x3_win2_kernel built with -DASD_WIN_PROFILE): per wave, s_memtime stamps
at start / end of main loop / end of kernel, cycles spent at the per-tap wait+barrier, at the first wait (window + first weights)
and at window reloads.   tools/win_profile.py  (GPU box; builds tools/bin/libasd_prof.so first)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "tools", "bin", "libasd_prof.so")
if not os.environ.get("ASD_HIP_LIB"):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(ROOT, "scaledreamer_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-DASD_WIN_PROFILE",
           "-I" + os.path.join(ROOT, "include"), "-shared", "-o", out] + [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".hip")]
    subprocess.check_call(cmd)
    os.environ["ASD_HIP_LIB"] = out
    os.execv(sys.executable, [sys.executable] + sys.argv)
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
from scaledreamer_amd._lib import GemmArgs, check, lib, stream
from scaledreamer_amd.diffusion import hip_ops as H

dev = torch.device("cuda", 0)
shapes = [(1, 512, 128, 128, 11), (1, 512, 128, 128, 10), (1, 256, 256, 256, 11), (1, 128, 512, 512, 11), (5, 64, 320, 320, 10), (5, 32, 640, 640, 10)]
for (B, hw, cin, cout, cfg) in shapes:
    x = torch.randn(B, hw, hw, cin, device=dev).half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device=dev).half() * 0.02)
    y = torch.empty(B * hw * hw, cout, device=dev, dtype=torch.float16)
    bn, nw = H.TILE_BN[cfg], (4 if cfg in (13, 14) else 8)
    items = B * (hw // 16) ** 2 * ((cout + bn - 1) // bn)
    ws = torch.zeros(items * nw * 8, device=dev, dtype=torch.int64)
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
    t = ws.cpu().numpy().reshape(items, nw, 8).astype(np.float64)
    t = t[t[:, 0, 0] > 0]
    start, end, first, wait, reload_, loop_end = (t[..., i] for i in range(6))
    k0 = start.min()
    total = end.max() - k0
    steps = 9 * (cin // 64)
    dur = end - start
    us = e0.elapsed_time(e1) * 1e3
    cu = (t[:, 0, 6].astype(np.int64) >> 8) & 0xf, (t[:, 0, 6].astype(np.int64) >> 13) & 0x7, t[:, 0, 7].astype(np.int64) & 0xf
    ids = (cu[2] * 8 + cu[1]) * 16 + cu[0]
    ncu = len(set(ids.tolist()))
    spans, busy, conc = [], [], []
    for c in set(ids.tolist()):
        m = ids == c
        sp = end[m].max() - start[m].min()
        spans.append(sp); busy.append(dur[m][:, 0].sum() / sp)
    total = float(np.median(spans))
    print(f"{B}x{hw}^2 {cin}->{cout} cfg{cfg}: {us:.1f} us; kernel span {total:.0f} ticks ({total / us:.0f} ticks/us); per-CU span median; resident blocks per CU {np.mean(busy):.2f} on {ncu} CUs; per wave: life {dur.mean():.0f} "
          f"(first wait {first.mean():.0f}, reloads {reload_.mean():.0f}, tap waits {wait.mean():.0f} = {wait.mean() / (steps - cin // 64):.0f}/tap, "
          f"epilogue {(end - loop_end).mean():.0f}); main-loop non-wait {(loop_end - start - first - wait - reload_).mean():.0f} = "
          f"{(loop_end - start - first - wait - reload_).mean() / steps:.0f}/tap; blocks {items}, start spread p50 {np.median(start[:, 0] - k0):.0f} max {(start[:, 0] - k0).max():.0f}")
