"""A/B of the tile configuration of the 3-D convolution's weight-gradient GEMM (ASD_C3_WGRAD_TILE, csrc/conv3d.hip): python tools/c3_wgrad_ab.py"""
import os, subprocess, sys
code = '''
import torch, sys
sys.path.insert(0, ".")
from scaledreamer_amd import ops
for (R, cin, cout) in ((128, 64, 64), (64, 128, 64), (64, 128, 128), (32, 256, 256)):
    x = torch.randn(1, R, R, R, cin, device="cuda"); dy = torch.randn(1, R, R, R, cout, device="cuda")
    ops.conv3d_wgrad(x, dy); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.conv3d_wgrad(x, dy)
    e1.record(); torch.cuda.synchronize()
    print(R, cin, cout, "%.3f ms" % (e0.elapsed_time(e1) / 3))
'''
for tile in ("0", "4", "8", "6"):
    env = dict(os.environ, ASD_C3_WGRAD_TILE=tile)
    print("tile", tile, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
