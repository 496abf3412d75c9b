"""Launch only the roofline kernel of bench.py (3x3 conv 128->128 @512x512 and 320->320 @64x64 batch 5, autotuned plans) 20 times each — the target of
the dedicated PMC passes whose per-launch HBM bytes bench.py reports as `roofline.traffic`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

for (B, hw, cin, cout) in [(1, 512, 128, 128), (5, 64, 320, 320)]:
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    for _ in range(20):
        H.conv3x3(x, w)
    torch.cuda.synchronize()
    print("plan", (B, hw, cin, cout), H.plan_table().get((B * hw * hw, cout, 9 * cin, (hw, cin, 1, 0, 1))))

# the plain GEMM bench.py reports as `roofline` (rocprof's top kernel: gemm_f16_kernel<64,64>) on its most frequent shape
M, N, K = 20480, 320, 320
a, w = torch.randn(M, K, device="cuda").half(), torch.randn(N, K, device="cuda").half()
for _ in range(20):
    H.gemm(a, w)
torch.cuda.synchronize()
print("plan gemm", (M, N, K), H.plan_table().get((M, N, K, K)))
