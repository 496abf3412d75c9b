"""Host side of the headline step: wall time of the calls that ENQUEUE one training step (no synchronisation inside the loop), next to the GPU time per
step.  If the host needs about as long as the GPU, the GPU idles wherever the host has to walk through Python between launches.
    python tools/host_step_time.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

torch.cuda.set_device(0)
torch.set_num_threads(1)
dev = torch.device("cuda", 0)
cfg, system, data = bench.build_system("hip", seed=10, workload=sys.argv[1] if len(sys.argv) > 1 else "asd_sd_nerf")
batch = bench.to_device(data.collate(), dev)
for _ in range(10):
    system.train_one_step(batch)
    batch = bench.to_device(data.collate(), dev)
torch.cuda.synchronize()
sect = {}
def tick(name, t0):
    t1 = time.perf_counter()
    sect[name] = sect.get(name, 0.0) + (t1 - t0)
    return t1
n = 50
w0 = time.perf_counter()
for _ in range(n):
    t = time.perf_counter()
    system.on_train_batch_start(); t = tick("update hooks", t)
    system.optimizer.zero_grad(set_to_none=True); t = tick("zero_grad", t)
    out = system.training_step(batch); t = tick("training_step (render + guidance forward + loss)", t)
    out["loss"].backward(); t = tick("backward", t)
    system.optimizer.step(); system.true_global_step += 1; t = tick("optimizer", t)
    batch = bench.to_device(data.collate(), dev); t = tick("collate + upload", t)
host = time.perf_counter() - w0
torch.cuda.synchronize()
wall = time.perf_counter() - w0
print(f"host enqueue {host / n * 1e3:.3f} ms per step, wall incl. the final drain {wall / n * 1e3:.3f} ms per step")
for k, v in sect.items():
    print(f"  {k:55s} {v / n * 1e3:7.3f} ms")
