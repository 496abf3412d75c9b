"""Which ATen operators still launch kernels inside a training step (and from where): torch.profiler over 3 steps of the
headline workload, grouped by operator + Python source line.   python tools/torch_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
import bench

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg, system, data = bench.build_system("hip", seed=10, workload=os.environ.get("ASD_WORKLOAD", "asd_sd_nerf"))
for _ in range(4):
    system.train_one_step(bench.to_device(data.collate(), dev))
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(N):
        system.train_one_step(bench.to_device(data.collate(), dev))
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt <= 0:
        continue
    if not e.key.startswith("aten::") and not e.key.startswith("Memcpy"):
        continue
    stack = [s for s in e.stack if "scaledreamer_amd" in s or "bench.py" in s]
    rows.append((dt / N, e.count / N, e.key, stack[0] if stack else (e.stack[0] if e.stack else "")))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ATen-launched device time per step: {tot / 1e3:.3f} ms")
for dt, cnt, key, where in rows[:60]:
    print(f"{dt:8.1f} us  x{cnt:5.1f}  {key[:40]:40s} {where[-90:]}")
