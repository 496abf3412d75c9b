"""UNet batch-5 forward only (HIP-graph replay x20) — target for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion.engine import HipBackend
dev = torch.device("cuda", 0)
be = HipBackend(dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(5, 4, 64, 64, generator=g).to(dev); t = torch.full((5,), 500, device=dev); ctx = torch.randn(5, 77, 1024, generator=g).to(dev)
for _ in range(22):
    be.unet(x, t, ctx)
torch.cuda.synchronize()
