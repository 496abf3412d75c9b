"""Launch only the roofline kernel of bench.py (3x3 conv 320->320 @64x64, batch 5, autotuned plan) 20 times — the target of
the dedicated PMC passes whose per-launch HBM bytes bench.py reports as `roofline.traffic`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

x = torch.randn(5, 64, 64, 320, device="cuda").half()
w = H.pack_conv3x3_weight(torch.randn(320, 320, 3, 3, device="cuda").half() * 0.02)
for _ in range(20):
    H.conv3x3(x, w)
torch.cuda.synchronize()
print("plan", H._plans.get((20480, 320, 2880, (64, 320, 1, 0, 1))))
