"""attention_fwd_kernel on the step's shapes (SD-2.1 UNet at B = 5, VAE mid block): us, TFLOP/s.   python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

dev = torch.device("cuda", 0)
#        batch heads  lq    lk   count/step
shapes = [(5, 5, 4096, 4096, 5), (5, 10, 1024, 1024, 5), (5, 20, 256, 256, 5), (5, 20, 64, 64, 1), (5, 5, 4096, 77, 5), (5, 10, 1024, 77, 5), (5, 20, 256, 77, 5),
          (5, 20, 64, 77, 1)]
tot = 0.0
for (B, h, lq, lk, cnt) in shapes:
    lks = (lk + 7) // 8 * 8
    q = torch.randn(B * lq, h * 64, device=dev).half()
    k = torch.randn(B * lks, h * 64, device=dev).half()
    vT = torch.randn(h * 64, B * lks, device=dev).half()
    for _ in range(3):
        o = H.attention(q, k, vT, B, h, lq, lk, lk_stride=lks)
    torch.cuda._sleep(400_000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.attention(q, k, vT, B, h, lq, lk, lk_stride=lks)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * h * lq * lk * 64
    tot += cnt * us
    # fp32 reference on one (batch, head)
    qf, kf, vf = q[:lq, :64].float(), k[:lk, :64].float(), vT[:64, :lk].float().T
    ref = torch.softmax(qf @ kf.T * 0.125, -1) @ vf
    err = float((o[:lq, :64].float() - ref).abs().max())
    print(f"B{B} heads {h:2d} lq {lq:4d} lk {lk:4d}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s  x{cnt}/step  max err {err:.2e}", flush=True)
print(f"attention per step: {tot / 1e3:.2f} ms")
