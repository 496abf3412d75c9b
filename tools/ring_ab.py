"""Few-block GEMM variants (tile configurations 15-19: 4-stage operand ring, intra-block split-K with 2 / 4 k-groups) against the tuned two-stage plans on the launches of the step that have FEW
blocks: numerics against the two-stage result, then L2-cold timings (operands rotated through a pool) over tile x split-K.
   python tools/ring_ab.py            (GPU box)   -> gpurun_out/ring_ab.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
NAMES = {1: "128x64", 2: "128x128", 3: "256x64", 4: "256x128", 6: "256x256", 13: "64x64", 16: "64x64r4", 17: "64x64k2", 18: "64x64k4", 19: "128x64k2", 20: "128x128k2"}
# (M, N, K, conv dict | None, launches per step)
SHAPES = [
    (1280, 1280, 1280, None, 30), (1280, 1280, 5120, None, 5), (1280, 2560, 1280, None, 5), (1280, 1280, 2560, None, 2),
    (5120, 640, 640, None, 35), (5120, 640, 2560, None, 5), (5120, 1280, 640, None, 5), (320, 1280, 1280, None, 4), (320, 1280, 2560, None, 3),
    (320, 2560, 1280, None, 1), (4096, 512, 512, None, 5), (4096, 512, 4096, None, 4), (640, 5120, 640, None, 5),
    (320, 1280, 11520, dict(Hin=8, Win=8, Cin=1280, Hout=8, Wout=8, stride=1, pad=1, upsample=0), 11),
    (320, 1280, 23040, dict(Hin=8, Win=8, Cin=2560, Hout=8, Wout=8, stride=1, pad=1, upsample=0), 3),
    (1280, 1280, 11520, dict(Hin=16, Win=16, Cin=1280, Hout=16, Wout=16, stride=1, pad=1, upsample=0), 6),
    (1280, 1280, 23040, dict(Hin=16, Win=16, Cin=2560, Hout=16, Wout=16, stride=1, pad=1, upsample=0), 2),
    (4096, 512, 4608, dict(Hin=64, Win=64, Cin=512, Hout=64, Wout=64, stride=1, pad=1, upsample=0), 16),
    (5120, 640, 5760, dict(Hin=32, Win=32, Cin=640, Hout=32, Wout=32, stride=1, pad=1, upsample=0), 6),
    (20480, 320, 320, None, 25), (20480, 320, 1280, None, 5),
]
POOL = 6
lines = []


def emit(s):
    print(s, flush=True)
    lines.append(s)


for (M, N, K, cv, cnt) in SHAPES:
    w = [torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5 for _ in range(POOL)]
    if cv:
        B = M // (cv["Hout"] * cv["Wout"])
        a = [torch.randn(B, cv["Hin"], cv["Win"], cv["Cin"], device=dev, dtype=torch.float16) for _ in range(POOL)]
    else:
        a = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(POOL)]
    out = torch.empty(M, N, device=dev, dtype=torch.float16)

    def run(i, tile, split):
        return H.gemm(a[i % POOL], w[i % POOL], out=out, conv=cv, M=M, tile_cfg=tile, split_k=split)

    def timeit(tile, split, n=24):
        try:
            for i in range(3):
                run(i, tile, split)
        except Exception as e:
            return None
        torch.cuda._sleep(int(3e6))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            run(i, tile, split)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    plan_us = timeit(0, None)     # the tuned plan of the shape (or the cost model)
    ref = run(0, 0, None).float().clone()
    res = []
    for tile in (13, 16, 17, 18, 19, 20, 1, 2):
        for split in (1, 2, 3, 4, 6, 8):
            if split > 1 and K // split < 256:
                continue
            if tile in (20, 2) and N % 128:
                continue
            us = timeit(tile, split, 16)
            if us is None:
                continue
            err = float((run(0, tile, split).float() - ref).abs().max() / ref.abs().max())
            res.append((us, tile, split, err))
    res.sort()
    best2 = min((r for r in res if r[1] in (13, 1, 2)), default=None)
    bestr = min((r for r in res if r[1] in (16, 17, 18, 19, 20)), default=None)
    emit(f"{M}x{N}x{K}{' conv' if cv else ''} x{cnt}: plan {plan_us:6.1f} us | best two-stage {NAMES[best2[1]]}/s{best2[2]} {best2[0]:6.1f} | best ring "
         f"{NAMES[bestr[1]]}/s{bestr[2]} {bestr[0]:6.1f} (err {bestr[3]:.1e}) | saved/step {(plan_us - min(plan_us, bestr[0])) * cnt:6.1f} us")
    emit("      " + "  ".join(f"{NAMES[t]}/s{s}:{u:.1f}" for u, t, s, e in res[:8]) + f"   max err {max(r[3] for r in res):.1e}")
    del a, w
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/ring_ab.txt", "w").write("\n".join(lines) + "\n")
