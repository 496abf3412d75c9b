"""GPU time of the diffusion half of one ASD step, by part (HIP events): VAE encode forward, VAE forward + input gradient,
UNet batch-5 forward (HIP-graph replay).   python tools/diffusion_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion.engine import HipBackend

dev = torch.device("cuda", 0)
be = HipBackend(dev)
g = torch.Generator().manual_seed(0)
x = torch.randn(5, 4, 64, 64, generator=g).to(dev)
t = torch.full((5,), 500, device=dev)
ctx = torch.randn(5, 77, 1024, generator=g).to(dev)
img = (torch.rand(1, 3, 512, 512, generator=g).to(dev) * 2 - 1).requires_grad_(True)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def vae_fb():
    img.grad = None
    be.encode(img).float().sum().backward()


with torch.no_grad():
    t_vf = timeit(lambda: be.encode(img.detach()))
t_vfb = timeit(vae_fb)
t_un = timeit(lambda: be.unet(x, t, ctx))
print(f"VAE fwd {t_vf:.2f} ms | VAE fwd+bwd {t_vfb:.2f} ms (bwd {t_vfb - t_vf:.2f}) | UNet x5 fwd {t_un:.2f} ms")
