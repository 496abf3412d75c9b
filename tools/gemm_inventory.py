"""GEMM / implicit-GEMM shape inventory of one ASD step (UNet batch 5 forward + VAE 512^2 forward + input-gradient),
recorded from the live HIP backend, then each unique shape timed in isolation (HIP events) -> time share + TFLOP/s.
  python tools/gemm_inventory.py [--mv]      (GPU box)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scaledreamer_amd.diffusion import hip_ops as H
from scaledreamer_amd.diffusion import weights as W
from scaledreamer_amd.diffusion.engine import HipBackend

dev = torch.device("cuda", 0)
mv = "--mv" in sys.argv
be = HipBackend(dev, unet_cfg=W.UNetConfig(camera_dim=16) if mv else W.UNetConfig())
be.hip_unet.use_graph = False
rec = collections.Counter()
orig = H.gemm


def spy(a, w, bias=None, row_bias=None, rows_per_group=0, residual=None, act=0, out=None, out_f32=False, split_k=None, conv=None, M=None):
    N, K = w.shape
    m = a.shape[0] if conv is None else M
    key = (m, N, K) + ((conv["Hin"], conv["Cin"], conv["stride"], conv["upsample"], conv["Hout"], conv["pad"]) if conv else ())
    rec[key] += 1
    return orig(a, w, bias, row_bias, rows_per_group, residual, act, out, out_f32, split_k, conv, M)


H.gemm = spy
import scaledreamer_amd.diffusion.engine as E
import scaledreamer_amd.diffusion.vae_hip as V
for mod in (E, V):
    if hasattr(mod, "gemm"):
        mod.gemm = spy
g = torch.Generator().manual_seed(0)
if mv:
    x = torch.randn(12, 4, 32, 32, generator=g).to(dev)
    t = torch.full((12,), 500, device=dev)
    ctx = torch.randn(12, 77, 1024, generator=g).to(dev)
    cam = torch.randn(12, 16, generator=g).to(dev)
    be.unet(x, t, ctx, camera=cam, num_frames=4)
    img = torch.rand(4, 3, 256, 256, generator=g).to(dev).requires_grad_(True)
else:
    x = torch.randn(5, 4, 64, 64, generator=g).to(dev)
    t = torch.full((5,), 500, device=dev)
    ctx = torch.randn(5, 77, 1024, generator=g).to(dev)
    be.unet(x, t, ctx)
    img = torch.rand(1, 3, 512, 512, generator=g).to(dev).requires_grad_(True)
n_unet = sum(rec.values())
be.encode(img * 2 - 1).float().sum().backward()
torch.cuda.synchronize()
H.gemm = orig
print(f"# {n_unet} UNet GEMM launches, {sum(rec.values()) - n_unet} VAE (fwd+bwd); {len(rec)} unique shapes")


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
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for key, cnt in rec.items():
    m, N, K = key[:3]
    wgt = torch.randn(N, K, device=dev).half()
    if len(key) > 3:
        hin, cin, stride, ups, hout, pad = key[3:]
        b = m // (hout * hout)
        xin = torch.randn(b, hin, hin, cin, device=dev).half()
        cv = dict(Hin=hin, Win=hin, Cin=cin, Hout=hout, Wout=hout, stride=stride, pad=pad, upsample=ups)
        fn = (lambda xin=xin, wgt=wgt, cv=cv, m=m: H.gemm(xin, wgt, conv=cv, M=m))
    else:
        a = torch.randn(m, K, device=dev).half()
        fn = (lambda a=a, wgt=wgt: H.gemm(a, wgt))
    us = timeit(fn)
    rows.append((us * cnt, us, cnt, key, 2.0 * m * N * K / us / 1e6, H.pick_split_k(m, N, K)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
fl = sum(2.0 * r[3][0] * r[3][1] * r[3][2] * r[2] for r in rows)
print(f"# total {tot / 1e3:.2f} ms  {fl / 1e12:.2f} TFLOP  -> {fl / tot / 1e6:.0f} TFLOP/s average")
print("share%  total_us   each_us  calls  TFLOP/s  split  (M, N, K[, Hin, Cin, stride, upsample, Hout, pad])")
for r in rows[:45]:
    print(f"{100 * r[0] / tot:5.1f}  {r[0]:9.1f}  {r[1]:8.1f}  {r[2]:5d}  {r[4]:7.1f}  {r[5]:5d}  {r[3]}")
