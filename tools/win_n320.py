"""3x3 conv 320->320 at 64x64 x batch 5 (UNet, 22 launches per step) and 640->640 at 32x32: window-kernel tile configurations, BN = 64
vs BN = 128 with a partly empty third channel tile (N = 320 = 2.5 x 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H
names = {9: "win1 64", 10: "win1 128", 11: "win2 64 8w", 12: "win2 128 8w", 14: "win2 64 4w", 15: "win2 128 4w"}
for (B, hw, cin, cout) in [(5, 64, 320, 320), (5, 64, 640, 320), (5, 64, 960, 320), (5, 32, 640, 640), (5, 32, 1280, 640)]:
    xs = [torch.randn(B, hw, hw, cin, device="cuda").half() for _ in range(4)]
    ws = [H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02) for _ in range(4)]
    ref = H.conv3x3(xs[0], ws[0], tile_cfg=11, split_k=1).float()
    out = []
    for t in names:
        for sk in (1, 2):
            try:
                y = H.conv3x3(xs[0], ws[0], tile_cfg=t, split_k=sk)
            except Exception as e:
                continue
            err = float((y.float() - ref).abs().max())
            torch.cuda._sleep(2_000_000)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16):
                H.conv3x3(xs[i % 4], ws[i % 4], tile_cfg=t, split_k=sk)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 16 * 1e3
            out.append((us, f"{names[t]}/s{sk}: {us:.1f} us {2.0 * B * hw * hw * cin * cout * 9 / us / 1e6:.0f} TF/s (err {err:.1e})"))
    out.sort()
    print(f"B{B} {hw}x{hw} {cin}->{cout}: plan = {H.conv3x3.__name__}", " | ".join(o[1] for o in out[:6]), flush=True)
