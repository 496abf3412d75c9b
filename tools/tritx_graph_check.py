"""Replays of the tri-plane transformer's captured passes against the uncaptured path, gradient by gradient, micro-batch by micro-batch (round 6: a captured
hipMemsetAsync node left NaNs from its second replay on; the passes now clear their cells with a fill kernel).  python tools/tritx_graph_check.py"""
import os, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from scaledreamer_amd.generators import TriplaneTransformer
from test_gpu_tritx import TRI_HD48, _seeded, rel
def make():
    t = TriplaneTransformer(**TRI_HD48)
    with torch.no_grad():
        for k, p in t.named_parameters():
            p.copy_(_seeded(f"acc.{k}", tuple(p.shape), 7, 1.0 if "norm" in k and k.endswith("weight") else 0.2))
    return t.cuda()
tes = [_seeded(f"acc.text{i}", (2, 77, 128), 7).cuda() for i in range(4)]
gps = [_seeded(f"acc.g{i}", (2, 3, 32, 16, 16), 7).cuda() for i in range(4)]
os.environ["ASD_TRITX_GRAPH"] = "0"
ref = make()
os.environ["ASD_TRITX_GRAPH"] = "1"
tt = make()
for mode in ("separate", "accumulate"):
    for i in range(4):
        if mode == "separate" or i == 0:
            for m in (ref, tt):
                for p in m.parameters():
                    p.grad = None
        for m in (ref, tt):
            (m(tes[i]) * gps[i]).sum().backward()
        torch.cuda.synchronize()
        errs = sorted(((rel(p.grad, q.grad.double()), k) for (k, p), (_, q) in zip(tt.named_parameters(), ref.named_parameters())), reverse=True)
        b = next(iter(tt._tritx_bufs.values()))
        print(mode, i, "fwd graph", b.fwd_graph is not None, "bwd graph", b.bwd_graph is not None, "worst", errs[:3], "n bad", sum(e > 1e-4 for e, _ in errs))

# the order of tests/test_gpu_tritx.py::test_graph_replays_and_fused_accumulation_match_the_uncaptured_path: reference first, capture in the middle of an accumulation
os.environ["ASD_TRITX_GRAPH"] = "0"
ref2 = make()
per = []
for i in range(4):
    before = {k: (p.grad.clone() if p.grad is not None else None) for k, p in ref2.named_parameters()}
    (ref2(tes[i]) * gps[i]).sum().backward()
    per.append({k: (p.grad - before[k]) if before[k] is not None else p.grad.clone() for k, p in ref2.named_parameters()})
for variant in ("none", "nested", "alloc", "nested_graph_off"):
    os.environ["ASD_TRITX_GRAPH"] = "0" if variant == "nested_graph_off" else "1"
    t2 = make()
    for i in range(4):
        before = {k: (p.grad.clone() if p.grad is not None else None) for k, p in t2.named_parameters()}
        pl = t2(tes[i])
        if variant.startswith("nested") and i == 2:
            with torch.no_grad():
                t2(tes[0])
        if variant == "alloc" and i == 2:
            junk = [torch.randn(1 << 22, device="cuda") for _ in range(8)]
            del junk
        (pl * gps[i]).sum().backward()
        torch.cuda.synchronize()
        bad = []
        for k, p in t2.named_parameters():
            d = p.grad - before[k] if before[k] is not None else p.grad
            e = rel(d, per[i][k].double())
            if e > 1e-4:
                bad.append((round(e, 3), k))
        print(variant, "micro-batch", i, "bad", len(bad), sorted(bad, reverse=True)[:4])
