"""Launch only the self-attention of the UNet's 64x64 level (target of SQ counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

B, L, heads = 5, 4096, 5
C_ = heads * 64
q, k, vT = (torch.randn(B * L, C_, device="cuda").half(), torch.randn(B * L, C_, device="cuda").half(), torch.randn(C_, B * L, device="cuda").half())
for _ in range(5):
    H.attention(q, k, vT, B, heads, L, L, L)
torch.cuda.synchronize()
