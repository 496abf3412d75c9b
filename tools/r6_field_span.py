"""The field-gradient span of the headline step timed on its own (bench.roofline_field_bwd: HIP events around asd_field_bwd on the step's samples):
    python tools/r6_field_span.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
cfg, system, data = bench.build_system("hip", seed=10, workload="asd_sd_nerf")
batch = bench.to_device(data.collate(), dev)
for _ in range(20):
    system.train_one_step(batch)
    batch = bench.to_device(data.collate(), dev)
torch.cuda.synchronize()
r = bench.roofline_field_bwd(system, batch, reps=20)
print({k: r[k] for k in ("avg_launch_ms", "asd_field_bwd_call_ms", "samples_per_launch", "frac")})
