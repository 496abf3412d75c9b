"""repeatability of the fused tri-plane field's gradients: the same fwd + bwd N times in one process, deviation of every gradient from the first run
(atomics reorder sums: ~1e-6; anything larger is a race or a missed hazard)   python tools/tri_stress.py [n] [iters] [keys: fd|nofd]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scaledreamer_amd import ops
from tri_mfma_check import cfg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3001
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fd = (sys.argv[3] if len(sys.argv) > 3 else "fd") == "fd"
g = torch.Generator().manual_seed(11)
planes = (torch.randn(3, 64, 64, 32, generator=g) * 0.5).cuda()
ws = [(torch.randn(o, i, generator=g) * (1.0 / i) ** 0.5).cuda() for o, i in ((64, 96), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))]
w6 = (ws[0].t().contiguous(), ws[1], ws[2], ws[3].t().contiguous(), ws[4], ws[5])
pts = (torch.rand(n, 3, generator=g) * 4.4 - 2.2).cuda()
gs = [torch.randn(n, d, generator=g).cuda() for d in (1, 3, 3, 3)]
c = cfg()
ref = None
worst = 0.0
for it in range(iters):
    sdf, feats, normal, fdg = ops.trifield_fwd(planes, c, w6, pts, True, True)
    dpl = torch.zeros_like(planes)
    dws = ops.trifield_bwd(planes, c, w6, pts, sdf, gs[0][:, 0].contiguous(), gs[1], gs[2] if fd else None, gs[3] if fd else None, dpl)
    junk = torch.randn(1 << 20, device="cuda").sin().sum()          # unrelated work between iterations
    torch.cuda.synchronize()
    cur = [sdf, feats, normal, fdg, dpl] + list(dws)
    if ref is None:
        ref = [t.clone() for t in cur]
        continue
    dev = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(cur, ref)]
    worst = max(worst, max(dev))
    if max(dev) > 1e-4:
        print("iteration", it, "deviations:", " ".join(f"{d:.1e}" for d in dev))
print(f"n={n} fd={fd}: worst deviation over {iters} iterations {worst:.2e}  (order: sdf features normal sdf_grad planes dW1s dW2s dW3s dW1f dW2f dW3f)")
