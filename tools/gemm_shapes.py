"""Where the GEMM / conv time of one training step goes, shape by shape.
   python tools/gemm_shapes.py            (GPU box)
Pass 1 (child process, ASD_GEMM_TRACE=1): one step of the headline workload; every asd_gemm_f16 launch prints its shape and plan.
Pass 2: each distinct shape is timed standalone (HIP events, L2-cold operands rotated through a pool) with the plan the step used,
and set against its own roofline max(flops / 2.5 PF/s, bytes / 6.3 TB/s).  Output: gpurun_out/gemm_shapes.txt, sorted by time lost."""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("ASD_GEMM_TRACE_CHILD"):
    import torch
    import bench
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg, system, data = bench.build_system("hip", seed=10, workload=os.environ.get("ASD_WORKLOAD", "asd_sd_nerf"))
    for _ in range(3):
        system.train_one_step(bench.to_device(data.collate(), dev))
    torch.cuda.synchronize()
    sys.stderr.write("ASD_STEP_BEGIN\n"); sys.stderr.flush()
    system.train_one_step(bench.to_device(data.collate(), dev))
    torch.cuda.synchronize()
    sys.stderr.write("ASD_STEP_END\n"); sys.stderr.flush()
    sys.exit(0)

env = dict(os.environ, ASD_GEMM_TRACE="1", ASD_GEMM_TRACE_CHILD="1", ASD_UNET_GRAPH="0")
err = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
body = err.split("ASD_STEP_BEGIN")[-1].split("ASD_STEP_END")[0]
pat = re.compile(r"ASD_GEMM (\d+) (\d+) (\d+) conv=(\d) (\d+) (\d+) (\d+) (\d+) (\d+) s=(\d+) p=(\d+) u=(\d+) cfg=(\d+) split=(\d+) act=(\d) res=(\d) f32=(\d) gn=(\d)")
order = [tuple(int(v) for v in m.groups()) for m in pat.finditer(body)]
calls = collections.Counter(order)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "gemm_order.txt"), "w") as f:      # launch order of the step (tools/gemm_in_step.py joins it with the step's kernel trace)
    for k in order:
        f.write(" ".join(str(v) for v in k) + "\n")
print(len(calls), "distinct launches,", sum(calls.values()), "per step", file=sys.stderr)

import torch
from scaledreamer_amd.diffusion import hip_ops as H

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rows = []
for key, cnt in calls.items():
    M, N, K, conv, Hin, Win, Cin, Hout, Wout, stride, pad, ups, cfg, split, act, res, f32, gn = key
    pool = 4
    w = [torch.randn(N * (4 if conv and ups == 3 else 1), K, device=dev, dtype=torch.float16) * K ** -0.5 for _ in range(pool)]     # parity form: [4][N][K]
    if conv:
        B = M // (Hout * Wout)
        a = [torch.randn(B, Hin, Win, Cin, device=dev, dtype=torch.float16) for _ in range(pool)]
        cv = dict(Hin=Hin, Win=Win, Cin=Cin, Hout=Hout, Wout=Wout, stride=stride, pad=pad, upsample=ups)
    else:
        a = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(pool)]
        cv = None
    nout = N // 2 if act == 2 else N
    out = torch.empty(M, nout, device=dev, dtype=torch.float32 if f32 else torch.float16)
    bias = torch.zeros(N, device=dev, dtype=torch.float16)
    resid = torch.zeros(M, nout, device=dev, dtype=torch.float16) if res else None

    def run(i):
        H.gemm(a[i % pool], w[i % pool], bias=bias, residual=resid, act=act, out=out, out_f32=bool(f32), conv=cv, M=M, tile_cfg=cfg + 1, split_k=split)

    try:
        for i in range(3):
            run(i)
    except Exception as e:      # a form this tool cannot rebuild from the trace line (parity-form weights): listed, not timed
        print("skipped", key, str(e)[-80:], file=sys.stderr)
        continue
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    flops = 2.0 * M * N * K
    if conv and ups == 2:
        flops *= 4.0 / 9.0 if stride == 1 else 1.0
    abytes = (a[0].numel() + w[0].numel()) * 2 + out.numel() * out.element_size() + (resid.numel() * 2 if res else 0)
    floor = max(flops / 2.5e15, abytes / 6.3e12) * 1e6
    rows.append((cnt * (us - floor), cnt, us, floor, flops / us / 1e9, key))
    del a, w, out
rows.sort(reverse=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.txt"), "w") as f:
    tot = sum(r[1] * r[2] for r in rows)
    f.write(f"# GEMM/conv launches of one step: {sum(r[1] for r in rows)}; standalone time {tot / 1e3:.2f} ms; roofline floor {sum(r[1] * r[3] for r in rows) / 1e3:.2f} ms\n")
    f.write("# lost_us/step  count  us  floor_us  TFLOP/s   M N K conv Hin Win Cin Hout Wout stride pad ups cfg split act res f32 gn\n")
    for lost, cnt, us, floor, tf, key in rows:
        f.write(f"{lost:9.1f} {cnt:4d} {us:8.1f} {floor:8.1f} {tf:8.1f}   " + " ".join(str(v) for v in key) + "\n")
print(open(os.path.join(ROOT, "gpurun_out", "gemm_shapes.txt")).read()[:9000])
