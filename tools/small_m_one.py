"""One small-M convolution shape under chosen (tile, split) pairs, cold weights, for rocprofv3 --kernel-trace --stats (kernel vs split-K
epilogue time).   python tools/small_m_one.py B hw cin cout  cfg:split [cfg:split ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

B, hw, cin, cout = (int(v) for v in sys.argv[1:5])
pairs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[5:]]
torch.manual_seed(0)
pool = max(2, int(400e6 // (cout * 9 * cin * 2)) + 1)
x = torch.randn(B, hw, hw, cin, device="cuda").half()
ws = [H.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5).half()) for _ in range(pool)]
res = torch.randn(B * hw * hw, cout, device="cuda").half()
for t, sk in pairs:
    for i in range(40):
        H.conv3x3(x, ws[i % pool], residual=res, tile_cfg=t + 1, split_k=sk)
    torch.cuda.synchronize()
