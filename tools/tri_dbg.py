"""debug / evidence tool: the fused tri-plane field (asd_trifield_*) and the composed path (HIP sampler + library heads) against a float64
torch restatement (F.grid_sample + double MLP) at sizes where fp32 summation order matters"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scaledreamer_amd.plugins
from scaledreamer_amd.registry import find
common = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere", "sdf_bias_params": 0.8}
g = torch.Generator().manual_seed(5)
geo = find("Triplane-transformer-sdf")(dict(common, space_generator_config=dict(inner_dim=64, condition_dim=128, triplane_low_res=32, triplane_high_res=64, triplane_dim=32, num_layers=1, num_heads=4, local_text=True, mlp_ratio=4))).cuda()
geo.do_update_step(0, 0)
cache = torch.randn(2, 3, 32, 64, 64, generator=g) * 0.5

def ref64(pts, c, gs, keys):
    c = c.double().requires_grad_(True)
    ws = [p.detach().double().requires_grad_(True) for p in geo._heads_weights()]
    B, n = pts.shape[:2]
    def enc(p):                          # [B, m, 3] world -> [B, m, 96]
        u = p.double() / 2.0
        proj = [u[..., [0, 1]], u[..., [0, 2]], u[..., [2, 1]]]
        feats = [F.grid_sample(c[:, k], proj[k][:, None], mode="bilinear", padding_mode="zeros", align_corners=False)[:, :, 0].permute(0, 2, 1) for k in range(3)]
        return torch.cat(feats, -1)
    def sdf_of(p):
        h = torch.relu(torch.relu(enc(p) @ ws[0].t()) @ ws[1].t()) @ ws[2].t()
        return h + (p.double().pow(2).sum(-1, keepdim=True).sqrt() - 0.8)
    s = sdf_of(pts)
    f = torch.relu(torch.relu(enc(pts) @ ws[3].t()) @ ws[4].t()) @ ws[5].t()
    eps = 0.01
    probes = [(pts + eps * torch.eye(3, device=pts.device)[k]).clamp(-2.0, 2.0) for k in range(3)]
    sg = torch.cat([(sdf_of(q) - s) / eps for q in probes], -1)
    out = {"sdf": s.reshape(B * n, 1), "features": f.reshape(B * n, 3), "sdf_grad": sg.reshape(B * n, 3), "normal": F.normalize(sg, dim=-1).reshape(B * n, 3)}
    sum((out[k] * gs[k].double()).sum() for k in keys).backward()
    names = [k for k, p in geo.named_parameters() if "network" in k]
    return c.grad, dict(zip(names, [w.grad for w in ws]))

for n in [int(a) for a in sys.argv[1:]] or (3001, 100001):
    pts = (torch.rand(2, n, 3, generator=g) * 4.4 - 2.2).cuda()
    for keys in (("sdf", "features"), ("sdf", "features", "normal", "sdf_grad")):
        gs = {k: torch.randn(2 * n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}
        def run(fused):
            os.environ["ASD_TRIFIELD"] = "1" if fused else "0"
            for p in geo.parameters(): p.grad = None
            c = cache.clone().cuda().requires_grad_(True)
            out = geo(pts, c, output_normal=True)
            sum((out[k] * gs[k]).sum() for k in keys).backward()
            return c.grad, {k: p.grad.clone() for k, p in geo.named_parameters() if p.grad is not None and "network" in k}
        c1, h1 = run(True); c0, h0 = run(False)
        cr, hr = ref64(pts, cache.cuda(), gs, keys)
        mx = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
        l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
        for tag, c, h in (("fused   ", c1, h1), ("composed", c0, h0)):
            print(n, "+".join(keys), tag, "vs float64: planes max %.1e l2 %.1e |" % (mx(c, cr), l2(c, cr)), " ".join("%s %.0e/%.0e" % (k.split(".")[0][:3] + k.split(".")[2], mx(h[k], hr[k]), l2(h[k], hr[k])) for k in hr))
