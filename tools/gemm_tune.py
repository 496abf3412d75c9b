"""Populate scaledreamer_amd/diffusion/gemm_plans.json: run the shipped workloads once with the autotuner on (each new GEMM
shape is timed over the valid tile / split-K candidates) and store the winners.   python tools/gemm_tune.py   (GPU box;
copy gpurun_out/gemm_plans.json into scaledreamer_amd/diffusion/)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ASD_GEMM_PLAN_FILE"] = "none"
# time the candidates as the step meets them: weights from HBM (evicted before every timed launch), activations cache-warm.  Against the
# back-to-back (cache-warm weights) timing of the first-use autotuner this re-ranked 73 of the 159 shapes and is worth +1.1 % on the step
os.environ.setdefault("ASD_GEMM_TUNE_COLD", "1")
import torch
import bench
from scaledreamer_amd.diffusion import hip_ops as H

torch.cuda.set_device(0)
torch.set_num_threads(1)
dev = torch.device("cuda", 0)
for wl in ("asd_sd_nerf", "asd_mv_nerf"):
    cfg, system, data = bench.build_system("hip", seed=10, workload=wl)
    for _ in range(3):
        system.train_one_step(bench.to_device(data.collate(), dev))
    torch.cuda.synchronize()
    print(wl, "plans:", len(H.plan_table()))
    del system
    torch.cuda.empty_cache()
out = os.path.join(ROOT, "gpurun_out", "gemm_plans.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
H.save_plans(out)
print("wrote", out, len(H.plan_table()))
