"""field_fwd with and without the finite-difference normal (4 vs 1 hash encodes per sample) on the samples of one view."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scaledreamer_amd import ops

cfg, system, data = bench.build_system("hip", seed=10, workload="asd_sd_nerf")
dev = torch.device("cuda", 0)
for _ in range(12):
    system.train_one_step(bench.to_device(data.collate(), dev))
batch = bench.to_device(data.collate(), dev)
ren, geo = system.renderer, system.geometry
with torch.no_grad():
    ri, t0, t1, pts, dirs, off, cnt, _ = ren._sample(batch["rays_o"].reshape(-1, 3).contiguous(), batch["rays_d"].reshape(-1, 3).contiguous())
grid = geo.encoding.encoding.encoding.params.detach()
w = [t.detach() for t in geo._weights()]
print("samples", pts.shape[0])
for normal in (True, False):
    for _ in range(3):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, normal)
    torch.cuda._sleep(600000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, normal)
    e1.record(); torch.cuda.synchronize()
    print("want_normal", normal, e0.elapsed_time(e1) / 10 * 1e3, "us")
# density only (the pruning pass's kernel: same 128 gathers per sample, density MLP only, nothing saved) on the same points
for _ in range(3):
    ops.field_density(geo._meta, geo._fcfg, grid, w[0], w[1], pts)
torch.cuda._sleep(600000)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.field_density(geo._meta, geo._fcfg, grid, w[0], w[1], pts)
e1.record(); torch.cuda.synchronize()
print("density only", e0.elapsed_time(e1) / 10 * 1e3, "us")
