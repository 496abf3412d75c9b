"""torch.profiler over a few ASD steps: which ATen ops (copies, casts, adds, fills) still launch kernels around the HIP path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

cfg, system, data = bench.build_system("hip", seed=10, workload=(sys.argv[1] if len(sys.argv) > 1 else "asd_sd_nerf"))
dev = torch.device("cuda", 0)
for _ in range(6):
    system.train_one_step(bench.to_device(data.collate(), dev))
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(N):
        system.train_one_step(bench.to_device(data.collate(), dev))
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt / N, e.count / N, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
print("us/step  calls/step  op  shapes")
for r in rows[:45]:
    print(f"{r[0]:8.1f} {r[1]:6.1f}  {r[2]:28s} {r[3]}")
