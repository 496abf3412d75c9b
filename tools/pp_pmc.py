"""Launch chosen (tile configuration, conv shape) pairs a few times: the target of SQ counter passes (tools/pp_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "11,14,20,23")]
for (B, hw, cin, cout) in [(1, 512, 128, 128), (1, 256, 256, 256)]:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    for c in cfgs:
        if cout % H.TILE_BN[c]:
            continue
        for _ in range(5):
            H.conv3x3(x, w, split_k=1, tile_cfg=c + 1)
    torch.cuda.synchronize()
