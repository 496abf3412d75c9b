"""asd_absmax_f32 on the sizes the generator's split passes see: time per call (rotating buffers), by value distribution"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd import ops
for n, name in ((512 * 512 * 27, "weights 512x512x27"), (64 * 64 * 27, "weights 64x64x27"), (128 ** 3 * 64, "volume 128^3 x 64")):
    for dist in ("randn", "ramp"):
        bufs = [(torch.randn(n, device="cuda") if dist == "randn" else torch.linspace(0, 1, n, device="cuda")) for _ in range(4)]
        for b in bufs: ops.absmax(b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): ops.absmax(bufs[i % 4])
        e1.record(); torch.cuda.synchronize()
        print(f"{name:22s} {dist:6s}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per call (incl. the 4-byte memset)")
