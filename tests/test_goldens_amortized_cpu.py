"""Oracle (CPU restatement) of the amortized render path vs goldens produced by the reference's own code
(tests/golden/make_goldens_amortized.py): voxel / tri-plane samplers and volsdf_density (in-tree torch code of the reference:
a true pin), ImportanceEstimator.sampling and the Hyper-iNGP VolSDF renderer (reference glue over injected primitives)."""
import os
import zlib

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)


def seeded(name, shape, seed, scale=1.0):
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g) * scale


def test_voxel_and_triplane_samplers_match_reference_grid_sample():
    from oracle import oracle as O

    g = _load("amortized_samplers")
    vox_cl = np.ascontiguousarray(g["voxel"].transpose(0, 2, 3, 4, 1))           # [B,C,D,H,W] -> [B,D,H,W,C]
    out = O.voxel_sample_fwd(vox_cl, g["points"][:1])
    np.testing.assert_allclose(out, g["tri_out"], rtol=1e-5, atol=2e-6)
    dv = O.voxel_sample_bwd(g["tri_g"], g["points"][:1], vox_cl.shape)
    np.testing.assert_allclose(dv.transpose(0, 4, 1, 2, 3), g["tri_dvoxel"], rtol=1e-5, atol=2e-6)
    pl_cl = np.ascontiguousarray(g["planes"].transpose(0, 1, 3, 4, 2))           # [B,3,C,H,W] -> [B,3,H,W,C]
    outp = O.triplane_sample_fwd(pl_cl, g["points"], 1.0)                          # box_warp = 2
    np.testing.assert_allclose(outp, g["plane_out"], rtol=1e-5, atol=2e-6)
    dp = O.triplane_sample_bwd(g["plane_g"], g["points"], pl_cl.shape, 1.0)
    np.testing.assert_allclose(dp.transpose(0, 1, 4, 2, 3), g["plane_dplanes"], rtol=1e-5, atol=2e-6)


def test_volsdf_density_matches_reference():
    from oracle import oracle as O
    from oracle import ref_amortized as RA

    g = _load("amortized_samplers")
    for key, inv_std in (("volsdf_30", 30.0), ("volsdf_200", 200.0)):
        np.testing.assert_allclose(O.volsdf_density(g["sdf"], inv_std), g[key], rtol=2e-6, atol=1e-5)
        np.testing.assert_allclose(RA.volsdf_density(torch.from_numpy(g["sdf"]), inv_std).numpy(), g[key], rtol=2e-6, atol=1e-5)


@pytest.mark.parametrize("tag", ["strat", "det"])
def test_importance_sampling_chain_matches_reference_estimator(tag):
    from oracle import ref_amortized as RA

    g = _load("amortized_importance_" + tag)
    centre, width, amp = (torch.from_numpy(g[k]) for k in ("centre", "width", "amp"))
    sigma_fn = lambda t0, t1: amp * torch.exp(-0.5 * (((t0 + t1) / 2 - centre) / width) ** 2)
    strat = tag == "strat"
    t0, t1, dbg = RA.importance_sampling(sigma_fn, centre.shape[0], int(g["n_prop"]), int(g["n_fine"]), float(g["near"]), float(g["far"]),
                                         g["jitter0"] if strat else None, g["jitter1"] if strat else None)
    np.testing.assert_allclose(dbg["s_prop"], g["s_prop"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(dbg["cdf"], g["cdfs"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(dbg["s_fine"], g["s_fine"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(t0.numpy(), g["t_starts"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(t1.numpy(), g["t_ends"], rtol=0, atol=1e-4)
    assert (t1.numpy() >= t0.numpy()).all() and t0.shape[1] == int(g["n_prop"]) + int(g["n_fine"]) + 1


def amortized_problem(g):
    """inputs of the Hyper-iNGP renderer golden, regenerated from the seeds (grid tables, hypernetwork weights)."""
    from golden_util import grid_params

    seed = int(g["seed"])
    P = dict(rays_o=torch.from_numpy(g["rays_o"]), rays_d=torch.from_numpy(g["rays_d"]), text_embed=torch.from_numpy(g["text_embed"]),
             jitter0=g["jitter0"], jitter1=g["jitter1"], n_prop=int(g["n_prop"]), n_fine=int(g["n_fine"]), near=0.1, far=4.0, radius=2.0,
             inv_std=float(np.exp(np.float32(0.340119) * 10.0)))
    P["grid"] = torch.from_numpy(grid_params(seed, 12_599_920, 0.004)).requires_grad_(True)
    from oracle import oracle as O
    nbg = O.grid_meta(16, 2, 19, 16, 1.0).n_params
    P["bgrid"] = torch.from_numpy(grid_params(seed + 1, nbg, 0.5)).requires_grad_(True)
    shapes = {"layers.0.weight": (64, 1024), "layers.1.weight": (64,), "layers.1.bias": (64,)}
    for tag, n_out in (("geo_hyper", 32 * 64 + 64 + 32 * 64 + 64 * 3), ("bg_hyper", 32 * 64 + 64 * 3)):
        d = {}
        for k, shp in {**shapes, "layers.3.weight": (n_out, 64), "layers.3.bias": (n_out,)}.items():
            if len(shp) == 2:
                t = seeded(f"{tag}.{k}", shp, seed, (2.0 / (shp[0] + shp[1])) ** 0.5)
            elif k.endswith("bias"):
                t = seeded(f"{tag}.{k}", shp, seed, 0.02)
            else:
                t = 1.0 + seeded(f"{tag}.{k}", shp, seed, 0.05)
            d[k] = t.requires_grad_(True)
        P[tag] = d
    return P


def amortized_loss(out, g):
    loss_eik = ((torch.linalg.norm(out["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
    return (out["comp_rgb"] * torch.from_numpy(g["g_rgb"]).to(out["comp_rgb"].device)).sum() + 20.0 * loss_sparsity + 100.0 * loss_eik, loss_eik


def check_amortized_against_golden(out, P, g, loss, loss_eik, tol=1.0):
    f = lambda t: t.detach().float().cpu().numpy()
    for k, atol in (("t_points", 2e-4), ("points", 3e-4), ("sdf", 5e-4), ("features", 2e-4), ("weights", 2e-4), ("opacity", 3e-4),
                    ("depth", 1e-3), ("comp_rgb", 5e-4), ("comp_rgb_bg", 1e-5), ("comp_normal", 2e-3), ("z_variance", 2e-3)):
        np.testing.assert_allclose(f(out[k]), g["out_" + k], rtol=0, atol=atol * tol, err_msg=k)
    # the finite-difference gradient divides by eps = 0.01: compare relative to its magnitude
    sg, ref = f(out["sdf_grad"]), g["out_sdf_grad"]
    assert np.abs(sg - ref).max() <= 3e-3 * tol * max(1.0, np.abs(ref).max())
    assert abs(loss.item() / float(g["loss"]) - 1) < 2e-4 * tol and abs(loss_eik.item() / float(g["loss_eikonal"]) - 1) < 2e-4 * tol
    for tag in ("geo_hyper", "bg_hyper"):
        for k, p in P[tag].items():
            got = f(p.grad).reshape(-1)
            ref_l2 = float(g[f"gl2_{tag}.{k}"])
            assert abs(np.linalg.norm(got.astype(np.float64)) / ref_l2 - 1) < 2e-3 * tol, (tag, k)
            sub = g[f"g_{tag}.{k}"]
            assert np.abs(got[::7] - sub).max() <= 2e-3 * tol * np.abs(sub).max() + 1e-7, (tag, k)
    gg = f(P["grid"].grad)
    assert abs(np.linalg.norm(gg.astype(np.float64)) / float(g["g_grid_l2"]) - 1) < 2e-3 * tol
    ref = g["g_grid_val"]
    assert np.abs(gg[g["g_grid_idx"]] - ref).max() <= 3e-3 * tol * np.abs(ref).max()
    gb = f(P["bgrid"].grad)
    assert abs(np.linalg.norm(gb.astype(np.float64)) / float(g["g_bgrid_l2"]) - 1) < 2e-3 * tol
    assert np.abs(gb[g["g_bgrid_idx"]] - g["g_bgrid_val"]).max() <= 3e-3 * tol * np.abs(g["g_bgrid_val"]).max()


def test_oracle_hyper_ingp_volsdf_renderer_matches_reference():
    from oracle import ref_amortized as RA

    g = _load("amortized_hyper_ingp_2x4x4")
    P = amortized_problem(g)
    out = RA.render(P)
    loss, loss_eik = amortized_loss(out, g)
    loss.backward()
    check_amortized_against_golden(out, P, g, loss, loss_eik)


def _seed_params(module, tag, seed, scale=1.0):
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(seeded(f"{tag}.{k}", tuple(p.shape), seed, scale))


def check_generator3d_golden(device="cpu", backend="library", tol=1.0):
    """scaledreamer_amd.generators.Generator3D vs the reference's in-tree `Generator` (same state-dict keys, forward, gradients); shared with
    the GPU test of the HIP backend (tests/test_gpu_conv3d.py)"""
    from scaledreamer_amd.generators import Generator3D

    g = _load("amortized_generator3d_16")
    seed = int(g["seed"])
    gen = Generator3D(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16, img_channels=32, channel_multiplier=1, backend=backend)
    assert list(gen.state_dict().keys()) == g["keys"].tolist()
    _seed_params(gen, "gen3d", seed, 0.3)
    with torch.no_grad():
        for m in gen.modules():
            if hasattr(m, "noise_const"):
                m.noise_const.copy_(seeded("gen3d.noise." + str(tuple(m.noise_const.shape)), tuple(m.noise_const.shape), seed))
    gen = gen.to(device)
    img = gen(seeded("gen3d.z", (2, 64), seed).to(device), seeded("gen3d.c", (2, 1024), seed).to(device), noise_mode="const")["image"]
    np.testing.assert_allclose(img.detach()[:, ::4, ::2, ::2, ::2].cpu().numpy(), g["image_sub"], rtol=1e-4 * tol, atol=1e-4 * tol * np.abs(g["image_sub"]).max())
    assert abs(img.double().norm().item() / float(g["image_l2"]) - 1) < 1e-5 * tol
    (img * seeded("gen3d.g", tuple(img.shape), seed).to(device)).sum().backward()
    gr = {k: p.grad for k, p in gen.named_parameters()}
    ref = g["g_affine"]
    np.testing.assert_allclose(gr["synthesis.blocks.1.conv0.affine.weight"].cpu().numpy(), ref, rtol=1e-3 * tol, atol=1e-4 * tol * np.abs(ref).max())
    for key, name in (("g_conv_l2", "synthesis.blocks.0.conv1.weight"), ("g_embed_l2", "mapping.embed.weight"), ("g_const_l2", "synthesis.first_block.const")):
        assert abs(gr[name].double().norm().item() / float(g[key]) - 1) < 1e-4 * tol, name
    return gen, gr


def test_generator3d_matches_reference_stylegan3d():
    """the torch-op restatement (backend="library", explicit) against the reference golden; the HIP backend refuses CPU tensors"""
    check_generator3d_golden("cpu", "library")
    from scaledreamer_amd.generators import Generator3D

    gen = Generator3D(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16, img_channels=32)
    assert gen.synthesis.backend == "hip"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gen(torch.zeros(1, 64), torch.zeros(1, 1024), noise_mode="const")


@pytest.mark.parametrize("local", [1, 0])
def test_triplane_transformer_matches_reference(local):
    from scaledreamer_amd.generators import TriplaneTransformer

    g = _load(f"amortized_triplane_transformer_local{local}")
    seed = int(g["seed"])
    tt = TriplaneTransformer(inner_dim=64, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2,
                             num_heads=4, local_text=bool(local), mlp_ratio=4)
    assert list(tt.state_dict().keys()) == g["keys"].tolist()
    _seed_params(tt, f"tri{local}", seed, 0.2)
    planes = tt(seeded("tri.text", (2, 77, 128) if local else (2, 128), seed))
    np.testing.assert_allclose(planes.detach().numpy(), g["planes"], rtol=1e-4, atol=1e-5 * np.abs(g["planes"]).max())
    (planes * seeded("tri.g", tuple(planes.shape), seed)).sum().backward()
    tg = {k: p.grad for k, p in tt.named_parameters()}
    for key, name in (("g_pos_embed", "pos_embed"), ("g_deconv", "deconv.weight")):
        np.testing.assert_allclose(tg[name].numpy(), g[key], rtol=1e-3, atol=1e-4 * np.abs(g[key]).max())
    assert abs(tg["layers.0.self_attn.to_q.weight"].double().norm().item() / float(g["g_q_l2"]) - 1) < 1e-4


def sampled_geometry_problem(name, g):
    seed = int(g["seed"])
    nh = 1 if name == "voxel" else 2
    din = 32 if name == "voxel" else 96
    shapes = [(64, din)] + [(64, 64)] * (nh - 1)
    heads = {}
    for tag, dout in (("sdf_network", 1), ("feature_network", 3)):
        ws = [seeded(f"{name}.{tag}.layers.{2 * i}.weight", s, seed, 0.25) for i, s in enumerate(shapes)]
        ws.append(seeded(f"{name}.{tag}.layers.{2 * len(shapes)}.weight", (dout, 64), seed, 0.25))
        heads[tag] = [w.requires_grad_(True) for w in ws]
    cache_shape = (1, 32, 16, 16, 16) if name == "voxel" else (2, 3, 32, 16, 16)
    cache = seeded(f"{name}.cache", cache_shape, seed).requires_grad_(True)
    return heads, cache, torch.from_numpy(g["points"])


def check_sampled_geometry(out, heads_grads, cache_grad, name, g, tol=1.0):
    seed = int(g["seed"])
    f = lambda t: t.detach().float().cpu().numpy()
    np.testing.assert_allclose(f(out["sdf"]), g["out_sdf"], rtol=0, atol=2e-5 * tol)
    np.testing.assert_allclose(f(out["features"]), g["out_features"], rtol=0, atol=2e-5 * tol)
    ref = g["out_sdf_grad"]
    assert np.abs(f(out["sdf_grad"]) - ref).max() <= 2e-3 * tol * max(1.0, np.abs(ref).max())   # differences / eps = 0.01
    assert np.abs(f(out["normal"]) - g["out_normal"]).max() <= 5e-3 * tol
    assert abs(np.linalg.norm(f(cache_grad).astype(np.float64)) / float(g["d_cache_l2"]) - 1) < 2e-3 * tol
    sub = g["d_cache_sub"]
    assert np.abs(f(cache_grad).reshape(-1)[::5] - sub).max() <= 3e-3 * tol * np.abs(sub).max()
    for key, got in heads_grads.items():
        ref = g["g_" + key]
        assert np.abs(f(got) - ref).max() <= 3e-3 * tol * np.abs(ref).max(), key


def sampled_geometry_loss(out, name, seed, device="cpu"):
    return sum((out[k] * seeded(f"{name}.g_{k}", tuple(out[k].shape), seed).to(device)).sum() for k in ("sdf", "features", "normal", "sdf_grad"))


@pytest.mark.parametrize("name", ["voxel", "triplane"])
def test_oracle_sampled_geometry_matches_reference(name):
    from oracle import ref_amortized as RA

    g = _load("amortized_geometry_" + name)
    heads, cache, pts = sampled_geometry_problem(name, g)
    out = RA.sampled_sdf_geometry(pts, cache, name, heads["sdf_network"], heads["feature_network"])
    sampled_geometry_loss(out, name, int(g["seed"])).backward()
    hg = {f"{tag}.layers.{2 * i}.weight": w.grad for tag, ws in heads.items() for i, w in enumerate(ws)}
    check_sampled_geometry(out, hg, cache.grad, name, g)
