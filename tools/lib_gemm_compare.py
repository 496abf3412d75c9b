"""The step's GEMM / conv shapes (profiles/r06_gemm_shapes_time_lost.txt) timed three ways on the same box, operands rotated through a pool:
this repo's kernel with the step's plan, the vendor library behind torch (hipBLASLt / rocBLAS for F.linear, MIOpen for F.conv2d in channels-last
fp16).  A yardstick only: where a library kernel is much faster, the shape has headroom a different tile could reach.
    python tools/lib_gemm_compare.py [min_lost_us]"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from scaledreamer_amd.diffusion import hip_ops as H

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
min_lost = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rows = []
for line in open(os.path.join(ROOT, "profiles", "r06_gemm_shapes_time_lost.txt")):
    if line.startswith("#"):
        continue
    v = line.split()
    if float(v[0]) < min_lost:
        continue
    rows.append((int(v[1]), float(v[2]), tuple(int(x) for x in v[5:])))


def timed(fn, n=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


out_lines = ["# count  ours_us  lib_us  (committed_us)   M N K conv Hin Win Cin Hout Wout stride pad ups cfg split act res f32 gn"]
gain = 0.0
for cnt, committed, key in rows:
    M, N, K, conv, Hin, Win, Cin, Hout, Wout, stride, pad, ups, cfg, split, act, res, f32, gn = key
    if ups:
        continue
    pool = 4
    w = [torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5 for _ in range(pool)]
    bias = torch.zeros(N, device=dev, dtype=torch.float16)
    nout = N // 2 if act == 2 else N
    out = torch.empty(M, nout, device=dev, dtype=torch.float32 if f32 else torch.float16)
    resid = torch.zeros(M, nout, device=dev, dtype=torch.float16) if res else None
    if conv:
        B = M // (Hout * Wout)
        a = [torch.randn(B, Hin, Win, Cin, device=dev, dtype=torch.float16) for _ in range(pool)]
        cv = dict(Hin=Hin, Win=Win, Cin=Cin, Hout=Hout, Wout=Wout, stride=stride, pad=pad, upsample=ups)
        ks = int(round((K // Cin) ** 0.5))
        a_nchw = [t.permute(0, 3, 1, 2) for t in a]                                              # channels-last views
        w_conv = [t.view(N, ks, ks, Cin).permute(0, 3, 1, 2) for t in w]                         # [N][Cin][kh][kw], channels-last strides
        lib = lambda i: F.conv2d(a_nchw[i % pool], w_conv[i % pool], bias, stride=stride, padding=pad)
    else:
        a = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(pool)]
        cv = None
        lib = lambda i: F.linear(a[i % pool], w[i % pool], bias)
    ours = lambda i: H.gemm(a[i % pool], w[i % pool], bias=bias, residual=resid, act=act, out=out, out_f32=bool(f32), conv=cv, M=M, tile_cfg=cfg + 1, split_k=split)
    try:
        t_ours = timed(ours)
    except Exception as e:
        print("skipped", key, str(e)[-80:], file=sys.stderr)
        continue
    try:
        t_lib = timed(lib)
    except Exception as e:
        t_lib = float("nan")
    if t_lib == t_lib and t_lib < t_ours:
        gain += cnt * (t_ours - t_lib)
    out_lines.append(f"{cnt:4d} {t_ours:8.1f} {t_lib:8.1f} ({committed:6.1f})   " + " ".join(str(x) for x in key))
    del a, w, out
out_lines.append(f"# sum over shapes where the library is faster: {gain:.1f} us per step (an upper bound: the library launches carry no residual / GEGLU / GroupNorm-record epilogues)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "lib_gemm_compare.txt"), "w").write("\n".join(out_lines) + "\n")
print("\n".join(out_lines))
