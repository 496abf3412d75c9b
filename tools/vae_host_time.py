"""Host time of the VAE encoder's C-ABI passes (enqueue only): python tools/vae_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import weights as W
from scaledreamer_amd.diffusion.vae_hip import HipVAEEncoder

cfg = W.VAEConfig()
enc = HipVAEEncoder(W.gen_params(W.vae_encoder_layout(cfg)[0], seed=1), cfg, "cuda")
x = torch.randn(1, 512, 512, 32, device="cuda").half()
for _ in range(3):
    m, saved = enc.forward_nhwc(x)
    enc.backward_nhwc(saved, torch.randn_like(m))
torch.cuda.synchronize()
for name in ("workspace_bytes", "forward", "backward"):
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "workspace_bytes":
            enc.workspace_bytes(1, 512, 512, False)
        elif name == "forward":
            m, saved = enc.forward_nhwc(x)
        else:
            enc.backward_nhwc(saved, m)
        ts.append((time.perf_counter() - t0) * 1e6)
    print(f"{name:16s} host us: median {sorted(ts)[5]:.0f}  min {min(ts):.0f}")
