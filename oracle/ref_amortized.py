"""CPU restatement of the amortized (multi-prompt) render path composed from the oracle primitives.

TEST INFRASTRUCTURE ONLY (see asd_oracle.c).  Follows
  custom/amortized/models/geometry/hyper_iNGP.py:18-111 (LinearHyperNetwork), :229-349 (Hypernet_Sdf.forward / forward_sdf)
  custom/amortized/models/background/multiprompt_neural_environment_hashgrid_map_background.py:60-116
  threestudio/models/estimators.py:23-118 (ImportanceEstimator.sampling, _transform_stot)
  custom/amortized/models/renderers/generative_space_volsdf_volume_renderer.py:172-446 (_forward, importance estimator,
  use_volsdf=True, training mode) and threestudio/models/renderers/neus_volume_renderer.py:19-23,93-96.
torch (CPU, fp32) carries the glue and autograd; the hash grid, the fused SDF field, the importance resampling and the
alpha compositing are the oracle's C functions.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as O


def hypernet_out_dims(n_input: int, out_dims: dict) -> dict:
    return {k: [n_input] + (list(v) if isinstance(v, (list, tuple)) else [v]) for k, v in out_dims.items()}


def linear_hypernetwork(x: torch.Tensor, p: dict, out_dims: dict) -> dict:
    """x [B, c_dim]; p: 'layers.0.weight' (no bias), 'layers.1.{weight,bias}' (LayerNorm), 'layers.3.{weight,bias}'
    (n_hidden_layers = 1).  Returns {name: [W_1 [B, in, out], W_2 ...]}  (hyper_iNGP.py:82-103)."""
    h = F.linear(x, p["layers.0.weight"])
    h = F.silu(F.layer_norm(h, (h.shape[-1],), p["layers.1.weight"], p["layers.1.bias"]))
    out = F.linear(h, p["layers.3.weight"], p["layers.3.bias"])
    res, start = {}, 0
    for name, ch in out_dims.items():
        ws = []
        for cin, cout in zip(ch[:-1], ch[1:]):
            ws.append(out[:, start:start + cin * cout].reshape(x.shape[0], cin, cout))
            start += cin * cout
        res[name] = ws
    return res


class _SdfField(torch.autograd.Function):
    """fused Hypernet_Sdf.forward for ONE prompt: (sdf, features, normal, sdf_grad) from points + per-prompt weights."""

    @staticmethod
    def forward(ctx, grid, w1d, w2d, w1f, w2f, points, m, fc, want_normal):
        a = [t.detach().numpy() for t in (grid, w1d, w2d, w1f, w2f, points)]
        if want_normal:
            sdf, feats, normal, fdg, _ = O.field_fwd(m, fc, *a, want_normal=True, want_fd_grad=True)
        else:
            (sdf, feats, normal, _), fdg = O.field_fwd(m, fc, *a, want_normal=False), None
        ctx.a, ctx.m, ctx.fc, ctx.want_normal = a, m, fc, want_normal
        z3 = torch.zeros(points.shape[0], 3)
        return (torch.from_numpy(sdf)[:, None], torch.from_numpy(feats), torch.from_numpy(normal) if want_normal else z3,
                torch.from_numpy(fdg) if want_normal else z3)

    @staticmethod
    def backward(ctx, d_sdf, d_feat, d_normal, d_fdg):
        n = lambda t: None if t is None else t.contiguous().numpy()
        dn, dg = (n(d_normal), n(d_fdg)) if ctx.want_normal else (None, None)
        dgrid, dw1d, dw2d, dw1f, dw2f = O.field_bwd(ctx.m, ctx.fc, *ctx.a, n(d_sdf.reshape(-1)), n(d_feat), dn, dg)
        return (*(torch.from_numpy(g) for g in (dgrid, dw1d, dw2d, dw1f, dw2f)), None, None, None, None)


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, meta):
        ctx.meta, ctx.x = meta, x.detach().numpy()
        return torch.from_numpy(O.hashgrid_fwd(meta, params.detach().numpy(), ctx.x))

    @staticmethod
    def backward(ctx, dout):
        return None, torch.from_numpy(O.hashgrid_bwd(ctx.meta, ctx.x, dout.contiguous().numpy())), None


def sdf_field_cfg(radius=2.0, sphere_radius=0.5, fd_eps=0.01):
    return O.field_cfg(radius=radius, bias_mode=3, bias_value=sphere_radius, activation=3, fd_eps=fd_eps, field_mode=1)


def hyper_geometry(points, space_cache, grid, m, fc, output_normal):
    """points [B, Np, 3]; space_cache = linear_hypernetwork(...) -> dict of [B*Np, .] tensors (hyper_iNGP.py:263-330)."""
    outs = []
    for b in range(points.shape[0]):
        w1d, w2d = (w[b].t().contiguous() for w in space_cache["sdf_weights"])          # [in,out] -> [out,in]
        w1f, w2f = (w[b].t().contiguous() for w in space_cache["feature_weights"])
        outs.append(_SdfField.apply(grid, w1d, w2d, w1f, w2f, points[b].contiguous(), m, fc, output_normal))
    sdf, feats, normal, sdf_grad = (torch.cat([o[i] for o in outs], 0) for i in range(4))
    out = {"sdf": sdf, "features": feats}
    if output_normal:
        out.update(normal=normal, shading_normal=normal, sdf_grad=sdf_grad)
    return out


def hyper_background(dirs, bg_cache, bgrid, mb):
    """dirs [B,H,W,3] -> sigmoid(bmm(relu(bmm(enc, W1)), W2))   (multiprompt_..._background.py:87-104)."""
    B, Hh, Ww, _ = dirs.shape
    enc = _HashGrid.apply(((dirs + 1.0) / 2.0).reshape(-1, 3).contiguous(), bgrid, mb).view(B, Hh * Ww, -1)
    w1, w2 = bg_cache["bg_weights"]
    return torch.sigmoid(torch.bmm(torch.relu(torch.bmm(enc, w1)), w2)).view(B, Hh, Ww, 3)


def volsdf_density(sdf, inv_std):
    a = torch.clamp(torch.as_tensor(inv_std, dtype=torch.float32), 0.0, 80.0)
    return a * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() * a))


def importance_sampling(sigma_fn, n_rays, n_prop, n_fine, near, far, jitter0=None, jitter1=None):
    """estimators.py:62-101 with one proposal level and sampling_type "uniform" -> (t_starts, t_ends) [n_rays, n_prop+n_fine+1]."""
    s_in = np.tile(np.array([[0.0, 1.0]], np.float32), (n_rays, 1))
    s_prop = O.importance_resample(s_in, s_in, n_prop, jitter0)
    t_prop = (s_prop * np.float32(far) + (np.float32(1.0) - s_prop) * np.float32(near)).astype(np.float32)
    sig = sigma_fn(torch.from_numpy(t_prop[:, :-1]), torch.from_numpy(t_prop[:, 1:]))
    cdf = O.transmittance_cdf(t_prop, sig.detach().numpy())
    s_fine = O.importance_resample(s_prop, cdf, n_fine, jitter1)
    t_fine = (s_fine * np.float32(far) + (np.float32(1.0) - s_fine) * np.float32(near)).astype(np.float32)
    t_all = O.merge_sorted(t_prop, t_fine)
    return torch.from_numpy(t_all[:, :-1].copy()), torch.from_numpy(t_all[:, 1:].copy()), dict(s_prop=s_prop, cdf=cdf, s_fine=s_fine)


class _AlphaWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, n_rays, per_ray):
        offset = (np.arange(n_rays) * per_ray).astype(np.int32)
        count = np.full(n_rays, per_ray, np.int32)
        n = alphas.shape[0]
        z, z3, bg = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((n_rays, 3), np.float32)
        a = alphas.detach().numpy()
        out = O.composite_fwd(a, z, z, z3, offset, count, bg, mode=1)
        ctx.stuff = (a, z, z3, offset, count, bg, out)
        return torch.from_numpy(out["weights"])

    @staticmethod
    def backward(ctx, dw):
        a, z, z3, offset, count, bg, out = ctx.stuff
        d_alpha, _, _ = O.composite_bwd(a, z, z, z3, offset, count, bg, out, d_weights=dw.contiguous().numpy(), mode=1)
        return torch.from_numpy(d_alpha), None, None


def render(P: dict) -> dict:
    """P: rays_o, rays_d [B,H,W,3], text_embed [B,1024], grid, bgrid (leaf tensors), geo_hyper / bg_hyper parameter dicts,
    jitter0, jitter1, n_prop, n_fine, near, far, radius, inv_std.  Returns the renderer's training output dictionary."""
    rays_o, rays_d = P["rays_o"], P["rays_d"]
    B, Hh, Ww, _ = rays_o.shape
    n_rays = B * Hh * Ww
    ro, rd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    m, mb = O.grid_meta(), O.grid_meta(16, 2, 19, 16, 1.0)
    fc = sdf_field_cfg(P.get("radius", 2.0), 0.5, 0.01)
    od = hypernet_out_dims(32, {"sdf_weights": [64, 1], "feature_weights": [64, 3]})
    cache = linear_hypernetwork(P["text_embed"], P["geo_hyper"], od)
    inv_std = float(P["inv_std"])

    def prop_sigma(t0, t1):
        with torch.no_grad():
            pos = ro[:, None, :] + rd[:, None, :] * ((t0 + t1)[..., None] / 2.0)
            g = hyper_geometry(pos.reshape(B, -1, 3), cache, P["grid"], m, fc, False)
            return volsdf_density(g["sdf"], min(max(inv_std, 1e-6), 1e6)).reshape(pos.shape[:2])
    t0, t1, dbg = importance_sampling(prop_sigma, n_rays, P["n_prop"], P["n_fine"], P["near"], P["far"], P.get("jitter0"), P.get("jitter1"))
    per_ray = t0.shape[1]
    ray_idx = torch.arange(n_rays)[:, None].expand(-1, per_ray).reshape(-1)
    t_starts, t_ends = t0.reshape(-1, 1), t1.reshape(-1, 1)
    t_dirs = rd[ray_idx]
    t_pos = (t_starts + t_ends) / 2.0
    positions = ro[ray_idx] + t_dirs * t_pos
    t_int = t_ends - t_starts
    geo = hyper_geometry(positions.reshape(B, -1, 3), cache, P["grid"], m, fc, True)
    rgb = torch.sigmoid(geo["features"])
    bg_cache = linear_hypernetwork(P["text_embed"], P["bg_hyper"], hypernet_out_dims(32, {"bg_weights": [64, 3]}))
    comp_bg = hyper_background(rays_d, bg_cache, P["bgrid"], mb)
    alpha = torch.abs(t_int.detach()) * volsdf_density(geo["sdf"], min(max(inv_std, 1e-6), 1e6))
    w = _AlphaWeights.apply(alpha[:, 0].contiguous(), n_rays, per_ray)[:, None]
    acc = lambda v: torch.zeros(n_rays, v.shape[-1]).index_add_(0, ray_idx, w * v)
    opacity = acc(torch.ones_like(w))
    depth = acc(t_pos)
    fg = acc(rgb)
    z_var = acc((t_pos - depth[ray_idx]) ** 2)
    comp = fg + comp_bg.reshape(n_rays, 3) * (1.0 - opacity)
    cn = F.normalize(acc(geo["normal"]), dim=-1)
    cn = torch.lerp(torch.zeros_like(cn), (cn.detach() + 1.0) / 2.0, opacity)
    v = lambda t, c: t.view(B, Hh, Ww, c)
    return dict(comp_rgb=v(comp, 3), comp_rgb_fg=v(fg, 3), comp_rgb_bg=comp_bg, opacity=v(opacity, 1), depth=v(depth, 1),
                z_variance=v(z_var, 1), comp_normal=v(cn, 3), weights=w, t_points=t_pos, t_intervals=t_int, t_dirs=t_dirs,
                ray_indices=ray_idx, points=positions, inv_std=torch.tensor(np.exp(np.float32(P["variance_param"]) * 10.0)) if "variance_param" in P else torch.tensor(inv_std),
                _debug=dbg, **geo)


# ---- generator-backed geometries (3DConv-net / Triplane-transformer-sdf) -----------------------------------------------
class _VoxelSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxel_cl, points):
        ctx.shape, ctx.pts = tuple(voxel_cl.shape), points.detach().numpy()
        return torch.from_numpy(O.voxel_sample_fwd(voxel_cl.detach().numpy(), ctx.pts))

    @staticmethod
    def backward(ctx, d_out):
        return torch.from_numpy(O.voxel_sample_bwd(d_out.contiguous().numpy(), ctx.pts, ctx.shape)), None


class _TriplaneSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes_cl, points):
        ctx.shape, ctx.pts = tuple(planes_cl.shape), points.detach().numpy()
        return torch.from_numpy(O.triplane_sample_fwd(planes_cl.detach().numpy(), ctx.pts, 1.0))

    @staticmethod
    def backward(ctx, d_out):
        return torch.from_numpy(O.triplane_sample_bwd(d_out.contiguous().numpy(), ctx.pts, ctx.shape, 1.0)), None


def sampled_sdf_geometry(points, cache, kind, sdf_w, feat_w, radius=2.0, sphere_r=0.8, eps=0.01):
    """stylegan_3dconv_net.py:259-346 / triplane_transformer.py:156-240: points [B,Np,3], cache channel-first as the generators
    emit it ([B,C,D,H,W] voxel | [B,3,C,H,W] planes), sdf_w / feat_w = lists of VanillaMLP weight matrices (no bias, ReLU)."""
    B = points.shape[0]

    def mlp(x, ws):
        for w in ws[:-1]:
            x = torch.relu(x @ w.t())
        return x @ ws[-1].t()

    def encode(pw):   # pw: world points [B, M, 3]
        p = pw / radius                                       # contract_to_unisphere_custom: bbox [-r, r] -> [-1, 1]
        if kind == "voxel":
            return _VoxelSample.apply(cache.permute(0, 2, 3, 4, 1).contiguous(), p)
        return _TriplaneSample.apply(cache.permute(0, 1, 3, 4, 2).contiguous(), p)

    def sdf_of(pw):
        return mlp(encode(pw), sdf_w) + pw.norm(dim=-1, keepdim=True) - sphere_r

    enc = encode(points)
    sdf = mlp(enc, sdf_w) + points.norm(dim=-1, keepdim=True) - sphere_r
    feats = mlp(enc, feat_w)
    offs = torch.eye(3) * eps
    po = (points[..., None, :] + offs).clamp(-radius, radius)                     # [B, Np, 3, 3]
    sdf_off = sdf_of(po.reshape(B, -1, 3)).reshape(B, -1, 3)
    sdf_grad = (sdf_off - sdf) / eps
    normal = F.normalize(sdf_grad, dim=-1)
    return {"sdf": sdf.reshape(-1, 1), "features": feats.reshape(-1, feats.shape[-1]), "normal": normal.reshape(-1, 3),
            "sdf_grad": sdf_grad.reshape(-1, 3)}
