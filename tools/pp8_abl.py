"""(needs tools/variants/pp8_small_image_conv.patch applied) time of conv3x3_pp8 (tile 25) at split 20 / 10 with cold weights, main kernel + epilogue (events): python tools/pp8_abl.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H
from small_m_conv import timeit
B, hw, cin, cout = 5, 8, 1280, 1280
pool = 14
x = torch.randn(B, hw, hw, cin, device="cuda").half()
ws = [H.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5).half()) for _ in range(pool)]
print(os.environ.get("ASD_HIP_LIB", "product"), " ".join(f"s{sk}: {timeit(lambda i: H.conv3x3(x, ws[i % pool], tile_cfg=int(os.environ.get("PP8_TILE", "26")), split_k=sk)):.1f}" for sk in (20, 10, 5)))
