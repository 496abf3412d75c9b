"""Time the background (environment-map) forward / backward kernels on one 64x64 view."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scaledreamer_amd import ops

cfg, system, data = bench.build_system("hip", seed=10, workload="asd_sd_nerf")
bg = system.background
dev = torch.device("cuda", 0)
batch = bench.to_device(data.collate(), dev)
dirs = batch["rays_d"].reshape(-1, 3).contiguous()
meta = bg._meta
print("n", dirs.shape[0], "levels", meta.n_levels, "params", meta.n_params)
w = [bg.network.layers[i].weight.detach() for i in (0, 2, 4)]
grid = bg.encoding.encoding.encoding.params.detach()
dcol = torch.randn(dirs.shape[0], 3, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("fwd us", timeit(lambda: ops.envmap_fwd(meta, grid, w[0], w[1], w[2], dirs)))
print("bwd us", timeit(lambda: ops.envmap_bwd(meta, grid, w[0], w[1], w[2], dirs, dcol)))
print("bwd us (zero d_color)", timeit(lambda: ops.envmap_bwd(meta, grid, w[0], w[1], w[2], dirs, torch.zeros_like(dcol))))
