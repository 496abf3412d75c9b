"""HBM-cold timing of the 8x8-level convolutions (M = 320, N = 1280, Cin = 1280 / 2560) on the tuned implicit-GEMM plans against the
weight-streaming kernel (tile configuration 25, csrc/gemm_ws.hip): every launch of a timed sequence reads another copy of the weights
(>= 700 MB of distinct weights per shape: more than the 256 MB Infinity Cache), activations stay warm as inside the step.
    python tools/r6_ws_time.py [out.txt]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from scaledreamer_amd.diffusion import hip_ops as H  # noqa: E402


def bench(cin, tile_cfg, sk, fused, copies, x, bias, temb, res, gamma, beta, reps=3):
    kw = dict(bias=bias, residual=res, row_bias=temb, rows_per_group=64, split_k=sk, tile_cfg=tile_cfg)
    if fused:
        kw.update(gn_rows=64, gn_apply=dict(gamma=gamma, beta=beta, eps=1e-5, silu=True))
    else:
        kw.update(gn_rows=64)
    def seq():
        for w in copies:
            out = H.conv3x3(x, w, **kw)
            if not fused:      # what the step runs behind a records launch: the apply kernel
                H.groupnorm_apply(out[0].view(5, 64, -1), gamma, beta, 1e-5, True, out[1])

    seq()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):          # GPU-paced: the launches of one pass over the copies replay back to back
        seq()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(copies) * 1e3)
    return best


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    torch.manual_seed(0)
    for cin in (1280, 2560):
        n = 1280
        x = (torch.randn(5, 8, 8, cin, device="cuda") * 0.5).half()
        ncopy = max(8, int(720e6 / (n * 9 * cin * 2)))
        copies = [(torch.randn(n, 9 * cin, device="cuda") * (9 * cin) ** -0.5).half() for _ in range(ncopy)]
        bias, temb = torch.randn(n, device="cuda").half(), torch.randn(5, n, device="cuda").half()
        res = torch.randn(320, n, device="cuda").half()
        gamma, beta = torch.ones(n, device="cuda").half(), torch.zeros(n, device="cuda").half()
        rows = []
        plans = [(17, 4), (2, 16), (13, 4), (16, 4)] + [(H.WS_TILE + 1, s) for s in (4, 5, 8, 10, 20) if (cin // 32) % s == 0]
        if os.environ.get("WS_ONLY"):
            plans = [(H.WS_TILE + 1, int(v)) for v in os.environ["WS_ONLY"].split(",")]
        for tile_cfg, sk in plans:
            for fused in ((True,) if os.environ.get("WS_ONLY") else (False, True)):
                try:
                    us = bench(cin, tile_cfg, sk, fused, copies, x, bias, temb, res, gamma, beta)
                except Exception as e:  # noqa: BLE001
                    us = float("nan")
                    print("failed", tile_cfg, sk, e)
                rows.append((us, tile_cfg, sk, fused))
        for us, tile_cfg, sk, fused in rows:
            line = (f"[{os.environ.get('ASD_HIP_LIB', 'product').split('/')[-1]}] conv3x3 8x8 x5 images, Cin {cin} -> {n} (K = {9 * cin}, {n * 9 * cin * 2 / 1e6:.1f} MB of weights, {ncopy} copies rotated): tile_cfg {tile_cfg:2d} split {sk:2d} "
                    f"{'reduction + GroupNorm in one launch' if fused else 'reduction with records + apply launch   '}: {us:6.1f} us per layer (graph replay)")
            print(line)
            if out:
                out.write(line + "\n")


if __name__ == "__main__":
    main()
