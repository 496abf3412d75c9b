"""Launch only the LDS-window convolution on two shapes (target of SQ counter passes: where do the wave cycles go?).
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -d out -- python tools/win_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from scaledreamer_amd._lib import lib
from scaledreamer_amd.diffusion import hip_ops as H

lib().asd_gemm_force_tile(C.c_int32(9))
for (B, hw, cin, cout) in [(1, 512, 128, 128), (5, 16, 2560, 1280)]:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    for _ in range(5):
        H.conv3x3(x, w, split_k=1)
    torch.cuda.synchronize()
