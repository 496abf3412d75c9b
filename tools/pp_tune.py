"""Re-tune the stride-1 3x3 convolution shapes of the committed plan table (scaledreamer_amd/diffusion/gemm_plans.json) with the
library's autotuner — every tile configuration incl. the ping-pong window kernels of csrc/gemm_pp.hip, weights HBM-cold — and write
the merged table to gpurun_out/gemm_plans.json.    python tools/pp_tune.py   (GPU box)"""
import ast, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ASD_GEMM_TUNE_COLD", "1")
import ctypes as C
import torch
from scaledreamer_amd._lib import GemmArgs, check, lib, stream
from scaledreamer_amd.diffusion import hip_ops as H

dev = torch.device("cuda", 0)
table = json.load(open(H.PLAN_FILE))
changed = 0
for key, old in sorted(table.items()):
    M, N, K, tail = ast.literal_eval(key)
    if not isinstance(tail, tuple):
        continue
    hin, cin, stride, up, pad = tail
    if stride != 1 or up != 0 or pad != 1 or hin % 16 or cin % 64 or N % 64 or M % (hin * hin):
        continue
    B = M // (hin * hin)
    x = torch.randn(B, hin, hin, cin, device=dev).half()
    w = (torch.randn(N, 9 * cin, device=dev) * (9 * cin) ** -0.5).half()
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    g = GemmArgs()
    g.A, g.W, g.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldw, g.ldc = 0, 9 * cin, N
    g.rows_per_group = 1
    g.conv, g.Hin, g.Win, g.Cin, g.Hout, g.Wout, g.stride, g.pad, g.upsample = 1, hin, hin, cin, hin, hin, 1, 1, 0
    g.zero_page = H.zero_page(dev).data_ptr()
    g.split_k = 0
    sc = H.tune_scratch(dev)
    check(lib().asd_gemm_tune(C.byref(g), C.c_void_p(sc.data_ptr()), C.c_int64(sc.numel()), stream()))
    new = H.plan_of(g)
    if list(new) != list(old):
        changed += 1
    print(key, old, "->", list(new), flush=True)
out = os.path.join(ROOT, "gpurun_out", "gemm_plans.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
H.save_plans(out)
print("wrote", out, len(H.plan_table()), "plans;", changed, "changed")
