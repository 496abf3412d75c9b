"""Host-side vs device-side timeline of the ASD step: per segment, host wall time WITHOUT added syncs (so a long host
segment = a blocking sync inside it or launch-bound python) and device time from events; plus total wall per step.
  python tools/step_timeline.py [--workload asd_mv_nerf]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

wl = "asd_mv_nerf" if "asd_mv_nerf" in sys.argv else "asd_sd_nerf"
torch.cuda.set_device(0)
torch.set_num_threads(1)
dev = torch.device("cuda", 0)
cfg, system, data = bench.build_system("hip", seed=10, workload=wl)
from scaledreamer_amd import dist as asd_dist


def step(marks=None, evs=None):
    def m(name):
        if marks is not None:
            marks.append((name, time.perf_counter()))
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
    m("start")
    batch = bench.to_device(data.collate(), dev)
    m("collate+h2d")
    system.on_train_batch_start()
    system.optimizer.zero_grad(set_to_none=True)
    m("update_hooks")
    out = system(batch)
    m("render_fwd")
    g_out = system.guidance(out["comp_rgb"], system.prompt_utils, **batch, rgb_as_latents=False)
    m("guidance")
    loss = g_out["loss_asd"] + 30.0 * (out["opacity"] ** 2 + 0.01).sqrt().mean()
    m("loss")
    loss.backward()
    m("backward")
    asd_dist.allreduce_mean_grads(system.optimizer)
    system.optimizer.step()
    system.true_global_step += 1
    m("adamw")


for _ in range(6):
    step()
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
print(f"un-instrumented: {(time.perf_counter() - t0) / N * 1e3:.2f} ms/step")
acc_h, acc_d = {}, {}
t0 = time.perf_counter()
for _ in range(N):
    marks, evs = [], []
    step(marks, evs)
    torch.cuda.synchronize()
    for i in range(1, len(marks)):
        k = marks[i][0]
        acc_h[k] = acc_h.get(k, 0) + (marks[i][1] - marks[i - 1][1]) * 1e3 / N
        acc_d[k] = acc_d.get(k, 0) + evs[i - 1].elapsed_time(evs[i]) / N
print(f"instrumented (sync per step): {(time.perf_counter() - t0) / N * 1e3:.2f} ms/step")
print(f"{'segment':16s} {'host ms':>9s} {'device ms':>10s}")
for k in acc_h:
    print(f"{k:16s} {acc_h[k]:9.2f} {acc_d[k]:10.2f}")
print(f"{'sum':16s} {sum(acc_h.values()):9.2f} {sum(acc_d.values()):10.2f}")
