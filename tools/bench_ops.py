"""Micro-benchmark of the hand-written diffusion kernels at the SD-2.1 UNet shapes of one ASD step (batch 5).
Prints achieved TFLOP/s (MFMA-bound ops) or GB/s (bandwidth-bound ops) per launch, HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(3e5 + 2.5e4 * reps))   # park the GPU while the launches are queued: GPU time, not the host's launch rate
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def r(*s):
    return torch.randn(*s, device="cuda").half()


B = 5
print("== conv3x3 (implicit GEMM)")
for (hw, cin, cout) in [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)]:
    x, w = r(B, hw, hw, cin), H.pack_conv3x3_weight(r(cout, cin, 3, 3))
    us = timeit(lambda: H.conv3x3(x, w))
    fl = 2 * B * hw * hw * cout * cin * 9
    print(f"  {hw:3d}^2 {cin:5d}->{cout:5d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  split_k={H.default_split(B*hw*hw, cout, 9*cin)}")
print("== gemm")
for (M, N, K) in [(20480, 320, 320), (20480, 640, 320), (20480, 2560, 320), (20480, 320, 1280), (5120, 640, 640), (5120, 5120, 640), (5120, 640, 2560),
                  (1280, 1280, 1280), (1280, 10240, 1280), (1280, 1280, 5120), (320, 1280, 1280), (400, 1280, 1024), (5, 21760, 1280)]:
    a, w = r(M, K), r(N, K)
    us = timeit(lambda: H.gemm(a, w))
    print(f"  M={M:6d} N={N:6d} K={K:5d}  {us:8.1f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s  split_k={H.default_split(M, N, K)}")
print("== attention (self / cross)")
for (L, heads, lk, lks) in [(4096, 5, 4096, 4096), (1024, 10, 1024, 1024), (256, 20, 256, 256), (64, 20, 64, 64), (4096, 5, 77, 80), (1024, 10, 77, 80)]:
    C_ = heads * 64
    q, k, vT = r(B * L, C_), r(B * lks, C_), r(C_, B * lks)
    us = timeit(lambda: H.attention(q, k, vT, B, heads, L, lk, lks))
    print(f"  L={L:5d} heads={heads:3d} lk={lk:5d}  {us:8.1f} us  {4 * B * heads * L * lk * 64 / us / 1e6:7.1f} TFLOP/s")
print("== bandwidth-bound")
for (rows, c) in [(20480, 320), (5120, 640), (1280, 1280)]:
    x, g, b = r(rows, c), r(c), r(c)
    us = timeit(lambda: H.layernorm(x, g, b))
    print(f"  layernorm {rows}x{c}: {us:7.1f} us  {rows * c * 4 / us / 1e3:7.1f} GB/s")
for (bb, hw, c) in [(B, 4096, 320), (B, 4096, 960), (B, 1024, 640), (B, 256, 1280), (B, 64, 2560), (1, 262144, 128), (1, 65536, 256), (1, 16384, 512)]:
    x, g, b = r(bb, hw, c), r(c), r(c)
    us = timeit(lambda: H.groupnorm(x, g, b, 1e-5, True))
    print(f"  groupnorm+silu {bb}x{hw}x{c}: {us:7.1f} us  {bb * hw * c * 6 / us / 1e3:7.1f} GB/s (read x2, write x1)")
h = r(20480, 2560)
us = timeit(lambda: H.geglu(h))
print(f"  geglu 20480x1280: {us:7.1f} us  {20480 * 1280 * 6 / us / 1e3:7.1f} GB/s")
