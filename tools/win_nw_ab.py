"""A/B of the window-convolution tile configurations on the step's 3x3 stride-1 shapes: 8 waves (10 / 11) vs 4 waves with twice
the channel extent per wave (13 / 14).  HIP events behind a spin kernel; results must agree bit for bit (same MFMA order per output).
  python tools/win_nw_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

shapes = [(1, 512, 128, 128), (1, 256, 256, 256), (1, 128, 512, 512), (5, 64, 320, 320), (5, 64, 640, 320), (5, 32, 640, 640), (1, 64, 512, 512),
          (4, 256, 128, 128), (12, 32, 320, 320)]
for (B, hw, cin, cout) in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    fl = 2.0 * B * hw * hw * cout * cin * 9
    ref, line = None, f"{B}x{hw}^2 {cin:4d}->{cout:4d}:"
    for t in (8, 9, 10, 11, 13, 14):
        bn = H.TILE_BN[t]
        if bn != 64 and cout % bn:
            continue
        y = H.conv3x3(x, w, tile_cfg=t + 1, split_k=1)
        if ref is None:
            ref = y
        ok = torch.equal(y, ref)
        torch.cuda._sleep(400_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            H.conv3x3(x, w, tile_cfg=t + 1, split_k=1)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f"  cfg{t}: {us:7.1f} us {fl / us / 1e6:6.0f} TF{'' if ok else ' MISMATCH'}"
    print(line, flush=True)
