"""The weight-streaming convolutions of the UNet's 8x8 / 16x16 levels (M = 320 / 1280 rows, 29.5 MB of weights per layer): time of every
tile configuration x split-K with COLD weights (a pool larger than the 256 MB Infinity Cache is rotated), against the time a plain read
of the same weights takes.    python tools/small_m_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scaledreamer_amd.diffusion import hip_ops as H


def timeit(fn, reps=24):
    for i in range(3):
        fn(i)
    torch.cuda._sleep(200000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    torch.manual_seed(0)
    for B, hw, cin, cout in ([(5, 8, 1280, 1280), (5, 8, 2560, 1280)] if len(sys.argv) > 1 else [(5, 8, 1280, 1280), (5, 16, 1280, 1280), (5, 8, 2560, 1280), (1, 64, 512, 512)]):
        pool = max(2, int(400e6 // (cout * 9 * cin * 2)) + 1)
        x = torch.randn(B, hw, hw, cin, device="cuda").half()
        ws = [H.pack_conv3x3_weight((torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5).half()) for _ in range(pool)]
        res = torch.randn(B * hw * hw, cout, device="cuda").half()
        sink = torch.empty(cout * 9 * cin // 2, device="cuda", dtype=torch.float32)
        rd = timeit(lambda i: torch.sum(ws[i % pool].view(torch.int16), dtype=torch.int32))
        plan = timeit(lambda i: H.conv3x3(x, ws[i % pool], residual=res))
        print(f"{(B, hw, cin, cout)}: weights {cout * 9 * cin * 2 / 1e6:.1f} MB, pool {pool}; reduction read {rd:.1f} us; plan {plan:.1f} us", flush=True)
        rows = []
        for t in range(len(H.TILE_BN)):
            bn, bm = H.TILE_BN[t], H.TILE_BM[t]
            if (cout % bn and bn != 64) or (t in H.PP_TILES and hw % (bm // 16)) or (t in H.WINDOW_TILES and t not in getattr(H, 'PP8_TILES', ()) and hw % 16):
                continue
            if t in getattr(H, 'PP8_TILES', ()) and (hw != 8 or (B * 64) % bm):
                continue
            for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
                if t in H.WINDOW_TILES and sk > cin // (64 if t in getattr(H, 'PP8_TILES', ()) else 128):
                    continue
                try:
                    us = timeit(lambda i: H.conv3x3(x, ws[i % pool], residual=res, tile_cfg=t + 1, split_k=sk), reps=12)
                except Exception as e:
                    continue
                rows.append((us, t, sk))
        rows.sort()
        print("   best: " + " | ".join(f"cfg{t} {H.TILE_BM[t]}x{H.TILE_BN[t]} s{sk}: {us:.1f}" for us, t, sk in rows[:10]), flush=True)


if __name__ == "__main__":
    main()
