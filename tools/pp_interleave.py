"""Why are conv3x3_pp_kernel<4,4> launches 15-20 % slower inside the step than back to back?  The VAE's 512^2 ResBlock sequence
(GroupNorm apply + SiLU -> conv with statistics epilogue) against the same convolution alone, for rocprofv3 --kernel-trace --stats.
    python tools/pp_interleave.py alone|gn|gn_fresh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
hw, c = 512, 128
torch.manual_seed(0)
xs = [torch.randn(1, hw * hw, c, device="cuda").half() for _ in range(4)]
w = H.pack_conv3x3_weight(torch.randn(c, c, 3, 3, device="cuda").half() * (9 * c) ** -0.5)
gamma, beta = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
y = H.groupnorm(xs[0], gamma, beta, 1e-6, True)
for i in range(30):
    if mode == "alone":
        H.conv3x3(y.view(1, hw, hw, c), w, gn_rows=hw * hw)
    elif mode == "gn":          # the step's order: normalise the previous output, convolve it
        y = H.groupnorm(xs[0], gamma, beta, 1e-6, True)
        H.conv3x3(y.view(1, hw, hw, c), w, gn_rows=hw * hw)
    else:                       # a different input tensor every time (nothing of it in the caches)
        y = H.groupnorm(xs[i % 4], gamma, beta, 1e-6, True)
        H.conv3x3(y.view(1, hw, hw, c), w, gn_rows=hw * hw)
torch.cuda.synchronize()
