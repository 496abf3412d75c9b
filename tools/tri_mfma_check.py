"""ops.trifield_fwd / trifield_bwd against a float64 restatement, straight at the C ABI (no module around it), plus kernel timings.
   python tools/tri_mfma_check.py [n] [--bwd]      (ASD_TRI_MFMA=0: the one-thread-per-sample kernels)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as F
from scaledreamer_amd import ops, _lib


def cfg():
    f = _lib.FieldCfg()
    for d in range(3):
        f.bbox_min[d], f.bbox_max[d] = -2.0, 2.0
    f.radius, f.bias_mode, f.bias_value = 2.0, _lib.ASD_BIAS_SPHERE, 0.8
    f.blob_scale, f.blob_std, f.activation = 0.0, 1.0, _lib.ASD_ACT_NONE
    f.fd_eps, f.n_hidden, f.n_feature_dims, f.field_mode = 0.01, 64, 3, _lib.ASD_FIELD_SDF
    return f


def ref64(planes_cl, ws, pts, gs=None):
    """planes_cl [3,H,W,32]; ws = W1s [64,96], W2s, W3s [1,64], W1f, W2f, W3f [3,64] (native layouts)"""
    c = planes_cl.double().permute(0, 3, 1, 2)[None].clone().requires_grad_(gs is not None)      # [1,3,32,H,W]
    w = [x.double().clone().requires_grad_(gs is not None) for x in ws]

    def enc(p):
        u = p.double()[None] / 2.0
        proj = [u[..., [0, 1]], u[..., [0, 2]], u[..., [2, 1]]]
        return torch.cat([F.grid_sample(c[:, k], proj[k][:, None], mode="bilinear", padding_mode="zeros", align_corners=False)[0, :, 0].t() for k in range(3)], -1)

    def sdf_of(p):
        return torch.relu(torch.relu(enc(p) @ w[0].t()) @ w[1].t()) @ w[2].t() + (p.double().pow(2).sum(-1, keepdim=True).sqrt() - 0.8)

    s = sdf_of(pts)
    f = torch.relu(torch.relu(enc(pts) @ w[3].t()) @ w[4].t()) @ w[5].t()
    sg = torch.cat([(sdf_of((pts + 0.01 * torch.eye(3, device=pts.device)[k]).clamp(-2.0, 2.0)) - s) / 0.01 for k in range(3)], -1)
    out = {"sdf": s[:, 0], "features": f, "sdf_grad": sg, "normal": F.normalize(sg, dim=-1)}
    if gs is None:
        return out
    sum((out[k] * gs[k].double()).sum() for k in gs).backward()
    return out, c.grad[0].permute(0, 2, 3, 1), [x.grad for x in w]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50_001
    bwd = "--bwd" in sys.argv
    g = torch.Generator().manual_seed(3)
    planes = (torch.randn(3, 64, 64, 32, generator=g) * 0.5).cuda()
    ws = [(torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).cuda() for o, i in ((64, 96), (64, 64), (1, 64), (64, 96), (64, 64), (3, 64))]
    w6 = (ws[0].t().contiguous(), ws[1], ws[2], ws[3].t().contiguous(), ws[4], ws[5])
    pts = (torch.rand(n, 3, generator=g) * 4.4 - 2.2).cuda()
    c = cfg()
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-30))
    for want_normal in (True, False):
        sdf, feats, normal, fdg = ops.trifield_fwd(planes, c, w6, pts, want_normal, True)
        torch.cuda.synchronize()
        r = ref64(planes, ws, pts)
        line = f"n={n} normal={want_normal}: sdf {l2(sdf, r['sdf']):.2e} features {l2(feats, r['features']):.2e}"
        if want_normal:
            line += f" sdf_grad {l2(fdg, r['sdf_grad']):.2e} normal {l2(normal, r['normal']):.2e}"
        print(line)
    sdf_only = ops.trifield_fwd(planes, c, w6, pts, False, False)[0]
    print("sdf only:", l2(sdf_only, ref64(planes, ws, pts)["sdf"]))
    # timing of the forward (events around 5 calls)
    big = (torch.rand(2_000_000, 3, generator=g) * 4.0 - 2.0).cuda()
    for want_normal in (True, False):
        ops.trifield_fwd(planes, c, w6, big, want_normal, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.trifield_fwd(planes, c, w6, big, want_normal, True)
        e1.record(); torch.cuda.synchronize()
        print(f"forward, 2 M samples, normal={want_normal}: {e0.elapsed_time(e1) / 5:.3f} ms")
    if bwd:
        keys = ("sdf", "features", "normal", "sdf_grad")
        gs = {k: torch.randn(n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}
        gs["sdf"] = gs["sdf"][:, 0]
        sdf, feats, normal, fdg = ops.trifield_fwd(planes, c, w6, pts, True, True)
        dpl = torch.zeros_like(planes)
        dws = ops.trifield_bwd(planes, c, w6, pts, sdf, gs["sdf"].contiguous(), gs["features"], gs["normal"], gs["sdf_grad"], dpl)
        torch.cuda.synchronize()
        _, cref, wref = ref64(planes, ws, pts, gs)
        print(f"bwd: planes {l2(dpl, cref):.2e} " + " ".join(f"dW{i} {l2(a, b):.2e}" for i, (a, b) in enumerate(zip(dws, wref))))
        # without a normal / sdf_grad gradient: one row per sample
        gs2 = {"sdf": gs["sdf"], "features": gs["features"]}
        dpl = torch.zeros_like(planes)
        dws = ops.trifield_bwd(planes, c, w6, pts, sdf, gs2["sdf"].contiguous(), gs2["features"], None, None, dpl)
        torch.cuda.synchronize()
        _, cref, wref = ref64(planes, ws, pts, gs2)
        print(f"bwd (no normal): planes {l2(dpl, cref):.2e} " + " ".join(f"dW{i} {l2(a, b):.2e}" for i, (a, b) in enumerate(zip(dws, wref))))
        # sdf gradient only
        dpl = torch.zeros_like(planes)
        dws = ops.trifield_bwd(planes, c, w6, pts, sdf, gs2["sdf"].contiguous(), None, None, None, dpl)
        torch.cuda.synchronize()
        _, cref, wref = ref64(planes, ws, pts, {"sdf": gs["sdf"]})
        print(f"bwd (sdf only): planes {l2(dpl, cref):.2e} " + " ".join(f"dW{i} {l2(a, b):.2e}" for i, (a, b) in enumerate(zip(dws[:3], wref[:3]))) + f" feature-head grads all zero: {all(float(d.abs().max()) == 0 for d in dws[3:])}")
        dpl = torch.zeros(3, 64, 64, 32, device="cuda")
        sdfb = ops.trifield_fwd(planes, c, w6, big, True, True)[0]
        gb = [torch.randn(big.shape[0], d, device="cuda") for d in (1, 3, 3, 3)]
        ops.trifield_bwd(planes, c, w6, big, sdfb, gb[0][:, 0].contiguous(), gb[1], gb[2], gb[3], dpl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.trifield_bwd(planes, c, w6, big, sdfb, gb[0][:, 0].contiguous(), gb[1], gb[2], gb[3], dpl)
        torch.cuda.synchronize()
        print(f"backward, 2 M samples: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
