"""GPU parity of the amortized (multi-prompt) render path, through the C ABI: kernels vs the oracle on seeded inputs,
drop-in sampler functions and the Hyper-iNGP VolSDF renderer vs the goldens produced by the reference's own code."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("stratified", [True, False])
def test_importance_resample_cdf_merge_match_oracle(stratified):
    from oracle import oracle as O
    from scaledreamer_amd import ops

    rng = np.random.default_rng(5)
    n_rays, e_in, n_out = 777, 129, 64
    vals = np.sort(rng.uniform(0.1, 4.0, (n_rays, e_in)).astype(np.float32), axis=1)
    sig = (rng.uniform(0, 1, (n_rays, e_in - 1)) ** 6 * 60).astype(np.float32)
    sig[:5] = 0.0
    cdf_o = O.transmittance_cdf(vals, sig)
    cdf = ops.transmittance_cdf(_dev(vals), _dev(sig)).cpu().numpy()
    np.testing.assert_allclose(cdf, cdf_o, rtol=0, atol=3e-7)
    jit = rng.uniform(0, 1, n_rays).astype(np.float32) if stratified else None
    out_o = O.importance_resample(vals, cdf_o, n_out, jit)
    out = ops.importance_resample(_dev(vals), _dev(cdf_o), n_out, None if jit is None else _dev(jit)).cpu().numpy()
    np.testing.assert_array_equal(out, out_o)                           # same fp32 arithmetic, same search result: bit-exact
    assert (np.diff(out, axis=1) >= 0).all() and out.min() >= vals.min() and out.max() <= vals.max()
    merged = ops.merge_sorted(_dev(vals), _dev(out_o)).cpu().numpy()
    np.testing.assert_array_equal(merged, O.merge_sorted(vals, out_o))
    np.testing.assert_array_equal(merged, np.sort(np.concatenate([vals, out_o], 1), axis=1))


@pytest.mark.parametrize("C_", [32, 8])
def test_voxel_and_triplane_kernels_match_oracle(C_):
    from oracle import oracle as O
    from scaledreamer_amd import ops

    rng = np.random.default_rng(6)
    B, D, H, W, M = 2, 9, 12, 10, 3000
    vox = rng.normal(size=(B, D, H, W, C_)).astype(np.float32)
    pts = rng.uniform(-1.15, 1.15, (B, M, 3)).astype(np.float32)
    out = ops.voxel_sample_fwd(_dev(vox), _dev(pts)).cpu().numpy()
    np.testing.assert_array_equal(out, O.voxel_sample_fwd(vox, pts))
    g = rng.normal(size=out.shape).astype(np.float32)
    dv = ops.voxel_sample_bwd(_dev(g), _dev(pts), vox.shape).cpu().numpy()
    np.testing.assert_allclose(dv, O.voxel_sample_bwd(g, pts, vox.shape), rtol=1e-4, atol=1e-4)
    pl = rng.normal(size=(B, 3, H, W, C_)).astype(np.float32)
    outp = ops.triplane_sample_fwd(_dev(pl), _dev(pts), 1.0).cpu().numpy()
    np.testing.assert_array_equal(outp, O.triplane_sample_fwd(pl, pts, 1.0))
    gp = rng.normal(size=outp.shape).astype(np.float32)
    dp = ops.triplane_sample_bwd(_dev(gp), _dev(pts), pl.shape, 1.0).cpu().numpy()
    np.testing.assert_allclose(dp, O.triplane_sample_bwd(gp, pts, pl.shape, 1.0), rtol=1e-4, atol=1e-4)
    x = rng.normal(size=(3, 37, 1000)).astype(np.float32)
    np.testing.assert_array_equal(ops.relayout(_dev(x)).cpu().numpy(), x.transpose(0, 2, 1))


def test_dropin_samplers_match_reference_goldens():
    """get_trilinear_feature / sample_from_planes with the reference's argument layout (channel-first) and autograd."""
    from scaledreamer_amd import samplers as S

    g = _load("amortized_samplers")
    vox = _dev(g["voxel"]).requires_grad_(True)
    pts = _dev(g["points"])
    f = S.get_trilinear_feature(pts[:1], vox)
    np.testing.assert_allclose(f.detach().cpu().numpy(), g["tri_out"], rtol=1e-5, atol=2e-6)
    (f * _dev(g["tri_g"])).sum().backward()
    np.testing.assert_allclose(vox.grad.cpu().numpy(), g["tri_dvoxel"], rtol=1e-4, atol=1e-5)
    planes = _dev(g["planes"]).requires_grad_(True)
    fp = S.sample_from_planes(planes, pts)
    np.testing.assert_allclose(fp.detach().cpu().numpy(), g["plane_out"], rtol=1e-5, atol=2e-6)
    (fp * _dev(g["plane_g"])).sum().backward()
    np.testing.assert_allclose(planes.grad.cpu().numpy(), g["plane_dplanes"], rtol=1e-4, atol=1e-5)
    bbox = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]]).cuda()
    np.testing.assert_allclose(S.contract_to_unisphere_custom(pts * 2, bbox, False).cpu().numpy(), g["contract"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("with_fd", [True, False])
def test_sdf_field_mode_matches_oracle(with_fd):
    from oracle import oracle as O
    from oracle import ref_amortized as RA
    from scaledreamer_amd import _lib, ops

    rng = np.random.default_rng(8)
    n = 5000
    om = O.grid_meta()
    oc = RA.sdf_field_cfg(2.0, 0.5, 0.01)
    hm = _lib.make_grid_meta(16, 2, 19, 16, 1.447269237440378)
    hc = _lib.FieldCfg()
    for d in range(3):
        hc.bbox_min[d], hc.bbox_max[d] = -2.0, 2.0
    hc.radius, hc.bias_mode, hc.bias_value, hc.blob_scale, hc.blob_std = 2.0, _lib.ASD_BIAS_SPHERE, 0.5, 0.0, 1.0
    hc.activation, hc.fd_eps, hc.n_hidden, hc.n_feature_dims, hc.field_mode = _lib.ASD_ACT_NONE, 0.01, 64, 3, _lib.ASD_FIELD_SDF
    grid = rng.uniform(-0.05, 0.05, om.n_params).astype(np.float32)
    w = [rng.normal(size=s).astype(np.float32) * 0.2 for s in ((64, 32), (1, 64), (64, 32), (3, 64))]
    pts = rng.uniform(-2.3, 2.3, (n, 3)).astype(np.float32)
    dg, dw, dp = _dev(grid), [_dev(a) for a in w], _dev(pts)
    sdf, feat, nrm, fdg, enc = ops.field_fwd(hm, hc, dg, *dw, dp, True, want_fd_grad=True)
    s2, f2, n2, g2, _ = O.field_fwd(om, oc, grid, *w, pts, want_normal=True, want_fd_grad=True)
    np.testing.assert_allclose(sdf.cpu().numpy(), s2, rtol=0, atol=2e-6)
    np.testing.assert_allclose(feat.cpu().numpy(), f2, rtol=0, atol=2e-6)
    assert np.abs(fdg.cpu().numpy() - g2).max() <= 5e-4 * max(1.0, np.abs(g2).max())      # differences of ~1e-6 divided by eps = 0.01
    ds, df = rng.normal(size=n).astype(np.float32), rng.normal(size=(n, 3)).astype(np.float32)
    dn = rng.normal(size=(n, 3)).astype(np.float32) if with_fd else None
    dgd = rng.normal(size=(n, 3)).astype(np.float32) * 0.1 if with_fd else None
    d_grid = torch.zeros_like(dg)
    got = ops.field_bwd(hm, hc, dg, *dw, dp, enc, sdf, _dev(ds), _dev(df), None if dn is None else _dev(dn), d_grid,
                        d_fd_grad=None if dgd is None else _dev(dgd))
    want = O.field_bwd(om, oc, grid, *w, pts, ds, df, dn, dgd)
    rel = lambda a, b: np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30)
    assert rel(d_grid.cpu().numpy(), want[0]) < 2e-3
    for a, b in zip(got, want[1:]):
        assert rel(a.cpu().numpy(), b) < 2e-3


def test_hyper_ingp_volsdf_renderer_matches_reference_golden():
    from test_goldens_amortized_cpu import amortized_loss, amortized_problem, check_amortized_against_golden

    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    g = _load("amortized_hyper_ingp_2x4x4")
    P = amortized_problem(g)
    enc = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
           "per_level_scale": 1.447269237440378}
    hyper = {"c_dim": 1024, "out_dims": {"sdf_weights": [64, 1], "feature_weights": [64, 3]}, "spectral_norm": False, "n_neurons": 64,
             "n_hidden_layers": 1}
    geo = find("Hyper-iNGP")({"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere",
                              "sdf_bias_params": 0.5, "shape_init": "sphere", "shape_init_params": 0.5, "hypernet_config": hyper,
                              "pos_encoding_config": enc}).cuda()
    mat = find("no-material")({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True}).cuda()
    bg = find("multiprompt-neural-hashgrid-environment-map-background")(
        {"color_activation": "sigmoid", "random_aug": True, "random_aug_prob": 0.2, "pos_encoding_config": dict(enc, per_level_scale=1.0)}).cuda()
    ren = find("generative-space-volsdf-volume-renderer")(
        {"radius": 2.0, "use_volsdf": True, "trainable_variance": False, "learned_variance_init": 0.340119, "estimator": "importance",
         "num_samples_per_ray": int(g["n_fine"]), "num_samples_per_ray_importance": int(g["n_prop"]), "near_plane": 0.1, "far_plane": 4.0,
         "train_chunk_size": 0}, geometry=geo, material=mat, background=bg).cuda()
    geo.do_update_step(0, 0)
    assert geo._fcfg is not None, "the Hyper-iNGP config of asd_sd_hyper_iNGP_50k.yaml must take the fused SDF kernels"
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.copy_(P["grid"].detach())
        bg.encoding.encoding.encoding.params.copy_(P["bgrid"].detach())
        for tag, net in (("geo_hyper", geo.hypernet), ("bg_hyper", bg.hypernet)):
            for k, p in net.named_parameters():
                p.copy_(P[tag][k].detach())
    jit = [_dev(g["jitter0"]), _dev(g["jitter1"])]
    ren.estimator.jitter_fn = lambda n, device: jit.pop(0)
    real = random.random
    random.random = lambda: 0.9
    try:
        ren.train(); geo.train(); bg.train(); mat.train()
        out = ren(rays_o=_dev(g["rays_o"]), rays_d=_dev(g["rays_d"]), light_positions=_dev(g["light_positions"]), text_embed=_dev(g["text_embed"]))
    finally:
        random.random = real
    loss, loss_eik = amortized_loss(out, g)
    loss.backward()
    Pg = {"grid": geo.encoding.encoding.encoding.params, "bgrid": bg.encoding.encoding.encoding.params,
          "geo_hyper": dict(geo.hypernet.named_parameters()), "bg_hyper": dict(bg.hypernet.named_parameters())}
    check_amortized_against_golden(out, Pg, g, loss, loss_eik, tol=2.0)
    assert abs(float(out["inv_std"]) - 30.0) < 1e-2


def test_hyper_ingp_amortized_asd_step_runs():
    """asd_sd_hyper_iNGP preset end to end: prompt draw -> hypernetworks -> importance-sampled VolSDF render -> SD guidance
    (reduced-width HIP UNet, full VAE) -> eikonal + sparsity -> backward into hash grids and hypernetworks -> Adam."""
    from scaledreamer_amd import presets
    from scaledreamer_amd.diffusion import weights as W
    from scaledreamer_amd.diffusion.engine import HipBackend
    from scaledreamer_amd.multiprompt import MultipromptRandomCameraIterableDataset, SyntheticMultiPromptProcessor
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    torch.manual_seed(0)
    random.seed(0)
    dev = torch.device("cuda", 0)
    cfg = presets.asd_sd_hyper_ingp()
    backend = HipBackend(dev, unet_cfg=W.UNetConfig(model_channels=128, context_dim=128), vae_cfg=W.VAEConfig(), seed=3)
    proc = SyntheticMultiPromptProcessor(cfg["data"]["prompt_library"]["train"], seed=2, device=dev, ctx_dim=128, global_dim=1024,
                                         front_threshold=30.0, back_threshold=30.0)
    system = find(cfg["system_type"])(cfg["system"], guidance_backend=backend, prompt_processor=proc)
    system.train()
    data = MultipromptRandomCameraIterableDataset(cfg["data"], rank=0, n_ranks=1)
    w_before = system.geometry.hypernet.layers[3].weight.detach().clone()
    g_before = system.geometry.encoding.encoding.encoding.params.detach().clone()
    for _ in range(2):
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.collate().items()}
        loss = system.train_one_step(batch)
    assert torch.isfinite(loss).item()
    assert system.geometry._fcfg is not None
    for k in ("train/loss_asd", "train/loss_eikonal", "train/loss_sparsity"):
        assert torch.isfinite(system.logged[k]).item(), k
    assert (system.geometry.hypernet.layers[3].weight.detach() != w_before).any().item()
    assert (system.geometry.encoding.encoding.encoding.params.detach() != g_before).any().item()


@pytest.mark.parametrize("name", ["voxel", "triplane"])
def test_sampled_geometry_classes_match_reference_golden(name):
    """`3DConv-net` / `Triplane-transformer-sdf` forward(points, space_cache, output_normal=True) and its gradients w.r.t. the
    space cache and the MLP heads, against the reference's own classes (tests/golden/make_goldens_amortized.py)."""
    from test_goldens_amortized_cpu import check_sampled_geometry, sampled_geometry_loss, sampled_geometry_problem

    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    g = _load("amortized_geometry_" + name)
    heads, cache, pts = sampled_geometry_problem(name, g)
    common = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere", "sdf_bias_params": 0.8}
    if name == "voxel":
        geo = find("3DConv-net")(dict(common, space_generator_config=dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16,
                                                                         img_channels=32, channel_multiplier=1)))
    else:
        geo = find("Triplane-transformer-sdf")(dict(common, space_generator_config=dict(
            backend="library",      # head dim 16: outside the HIP generator's family, the reference fixtures meet the torch-op restatement
            inner_dim=64, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2, num_heads=4,
            local_text=True, mlp_ratio=4)))
    geo = geo.cuda()
    geo.do_update_step(0, 0)
    with torch.no_grad():
        for tag, net in (("sdf_network", geo.sdf_network), ("feature_network", geo.feature_network)):
            for (k, p), w in zip(net.named_parameters(), heads[tag]):
                p.copy_(w.detach())
    cache_d = cache.detach().cuda().requires_grad_(True)
    out = geo(pts.cuda(), cache_d, output_normal=True)
    sampled_geometry_loss(out, name, int(g["seed"]), device="cuda").backward()
    hg = {f"{tag}.{k}": p.grad for tag, net in (("sdf_network", geo.sdf_network), ("feature_network", geo.feature_network))
          for k, p in net.named_parameters()}
    check_sampled_geometry(out, hg, cache_d.grad, name, g, tol=2.0)
    # the chunked / checkpointed evaluation used for very large sample counts (CHECKPOINT_ABOVE) is the same function
    for p_ in geo.parameters():
        p_.grad = None
    geo.CHECKPOINT_ABOVE, geo.CHECKPOINT_CHUNK = 0, 7 * pts.shape[0]        # ragged chunks of 7 points per batch element
    cache_c = cache.detach().cuda().requires_grad_(True)
    out_c = geo(pts.cuda(), cache_c, output_normal=True)
    # (fp32 library GEMMs of another row count sum in another order: 1e-7 on the SDF, x 1 / eps = 100 on the finite differences)
    for k in out:
        assert float((out_c[k] - out[k]).abs().max()) <= 2e-3 * float(out[k].abs().max()) + 1e-5, k
    sampled_geometry_loss(out_c, name, int(g["seed"]), device="cuda").backward()
    scale = float(cache_d.grad.abs().max())
    assert float((cache_c.grad - cache_d.grad).abs().max()) <= 2e-3 * scale
    for tag, net in (("sdf_network", geo.sdf_network), ("feature_network", geo.feature_network)):
        for k, p_ in net.named_parameters():
            ref = hg[f"{tag}.{k}"]
            assert float((p_.grad - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("kind", ["3dconv", "triplane"])
def test_generator_backed_amortized_step_runs(kind):
    """asd_sd_3dconv_net / asd_mv_triplane_transformer presets (generator resolution reduced for test time): generator ->
    feature volume / planes -> HIP samplers -> importance-sampled VolSDF render -> guidance -> backward into the generator."""
    from scaledreamer_amd import presets
    from scaledreamer_amd.diffusion import weights as W
    from scaledreamer_amd.diffusion.engine import HipBackend
    from scaledreamer_amd.multiprompt import MultipromptRandomCameraIterableDataset, SyntheticMultiPromptProcessor
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    torch.manual_seed(0)
    random.seed(0)
    dev = torch.device("cuda", 0)
    if kind == "3dconv":
        cfg = presets.asd_sd_3dconv_net()
        cfg["system"]["geometry"]["space_generator_config"]["img_resolution"] = 32
        backend = HipBackend(dev, unet_cfg=W.UNetConfig(model_channels=128, context_dim=128), vae_cfg=W.VAEConfig(), seed=3)
        local = False
    else:
        cfg = presets.asd_mv_triplane_transformer()
        # head dimension 48 (192 / 4) as in the shipped YAML: the generator must run on csrc/tritx.hip, not on the library restatement
        cfg["system"]["geometry"]["space_generator_config"].update(num_layers=2, inner_dim=192, num_heads=4, condition_dim=128)
        backend = HipBackend(dev, unet_cfg=W.UNetConfig(model_channels=128, context_dim=128, camera_dim=16), vae_cfg=W.VAEConfig(), seed=3)
        local = True
    proc = SyntheticMultiPromptProcessor(cfg["data"]["prompt_library"]["train"], seed=2, device=dev, ctx_dim=128, global_dim=1024,
                                         front_threshold=30.0, back_threshold=30.0, use_local_text_embeddings=local,
                                         use_perp_neg=kind == "3dconv")
    system = find(cfg["system_type"])(cfg["system"], guidance_backend=backend, prompt_processor=proc)
    system.train()
    data = find(cfg["data_type"])(cfg["data"], rank=0, n_ranks=1)
    gen_w = next(p for n, p in system.geometry.space_generator.named_parameters() if p.ndim >= 2)
    before = gen_w.detach().clone()
    for _ in range(2):         # (a configuration outside the HIP generator's family raises: no silent drop to library ops)
        batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in data.collate().items()}
        loss = system.train_one_step(batch)
    assert torch.isfinite(loss).item()
    assert (gen_w.detach() != before).any().item(), "no gradient reached the generator"


def _triplane_field_float64(geo, pts, cache, gs, keys):
    """float64 restatement of Triplane-transformer-sdf.forward (F.grid_sample lookups + double heads + sphere bias + finite differences) and its
    gradients w.r.t. the planes and the six head weights"""
    import torch.nn.functional as F

    c = cache.double().requires_grad_(True)
    ws = [p.detach().double().requires_grad_(True) for p in geo._heads_weights()]
    B, n = pts.shape[:2]

    def enc(p):
        u = p.double() / 2.0
        proj = [u[..., [0, 1]], u[..., [0, 2]], u[..., [2, 1]]]
        return torch.cat([F.grid_sample(c[:, k], proj[k][:, None], mode="bilinear", padding_mode="zeros", align_corners=False)[:, :, 0].permute(0, 2, 1)
                          for k in range(3)], -1)

    def sdf_of(p):
        return torch.relu(torch.relu(enc(p) @ ws[0].t()) @ ws[1].t()) @ ws[2].t() + (p.double().pow(2).sum(-1, keepdim=True).sqrt() - 0.8)

    s = sdf_of(pts)
    f = torch.relu(torch.relu(enc(pts) @ ws[3].t()) @ ws[4].t()) @ ws[5].t()
    sg = torch.cat([(sdf_of((pts + 0.01 * torch.eye(3, device=pts.device)[k]).clamp(-2.0, 2.0)) - s) / 0.01 for k in range(3)], -1)
    out = {"sdf": s.reshape(B * n, 1), "features": f.reshape(B * n, 3), "sdf_grad": sg.reshape(B * n, 3), "normal": F.normalize(sg, dim=-1).reshape(B * n, 3)}
    sum((out[k] * gs[k].double()).sum() for k in keys).backward()
    return {k: v.detach() for k, v in out.items()}, c.grad, [w.grad for w in ws]


_SAMPLED_COMMON = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere", "sdf_bias_params": 0.8}
_TRI_GEN = dict(backend="library", inner_dim=64, condition_dim=128, triplane_low_res=32, triplane_high_res=64, triplane_dim=32, num_layers=1, num_heads=4, local_text=True, mlp_ratio=4)


@pytest.mark.parametrize("kind", ["voxel", "triplane"])
def test_fused_sampled_field_matches_the_composed_path(kind, monkeypatch):
    """`3DConv-net` / `Triplane-transformer-sdf` with lookup + MLP heads + bias + finite differences as ONE kernel each way (asd_voxfield_* /
    asd_trifield_*) against the composed path of the same module (HIP sampler + library heads, itself pinned by the reference goldens above):
    outputs and every gradient - feature volume / planes, all head weights."""
    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    g = torch.Generator().manual_seed(5)
    torch.manual_seed(5)          # (the heads' default initialisation comes from the global generator: a ReLU pre-activation within fp32 rounding of zero
    #                                 would switch a row of the 3001 on in one path and off in the other — 2e-2 of a gradient; this draw has none)
    if kind == "voxel":
        geo = find("3DConv-net")(dict(_SAMPLED_COMMON, space_generator_config=dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16,
                                                                                  img_channels=32, channel_multiplier=1)))
        cache = (torch.randn(2, 32, 16, 16, 16, generator=g) * 0.5)
    else:
        geo = find("Triplane-transformer-sdf")(dict(_SAMPLED_COMMON, space_generator_config=dict(_TRI_GEN)))
        cache = torch.randn(2, 3, 32, 64, 64, generator=g) * 0.5
    n = 3001
    geo = geo.cuda()
    geo.do_update_step(0, 0)
    assert geo._fcfg is not None
    pts = (torch.rand(2, n, 3, generator=g) * 4.4 - 2.2).cuda()          # some points outside the box: zero padding of the lookup
    gs = {k: torch.randn(2 * n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}

    def run(fused):
        monkeypatch.setenv("ASD_VOXFIELD", "1" if fused else "0")
        monkeypatch.setenv("ASD_TRIFIELD", "1" if fused else "0")
        for p in geo.parameters():
            p.grad = None
        c = cache.clone().cuda().requires_grad_(True)
        out = geo(pts, c, output_normal=True)
        sum((out[k] * gs[k]).sum() for k in gs).backward()
        heads = {k: p.grad.clone() for k, p in geo.named_parameters() if p.grad is not None and ("sdf_network" in k or "feature_network" in k)}
        return {k: out[k].detach() for k in gs}, c.grad, heads

    o1, c1, h1 = run(True)
    o0, c0, h0 = run(False)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    for k in gs:
        tol = 2e-3 if k in ("normal", "sdf_grad") else 2e-5          # (finite differences divide 1e-7 rounding differences by eps = 0.01)
        assert rel(o1[k], o0[k]) < tol, k
    assert rel(c1, c0) < 2e-3
    assert set(h1) == set(h0) and len(h1) == (4 if kind == "voxel" else 6)
    for k in h0:
        assert rel(h1[k], h0[k]) < 2e-3, k


def test_fused_triplane_field_across_backward_chunks_against_float64(monkeypatch):
    """270 001 samples per batch entry, the backward pass walking 100 000-sample chunks (ASD_TRI_CHUNK; 1 M by default), with random upstream gradients: at this size the fp32 sums
    (600 k contributions into 12 k plane cells through atomics, weight gradients over 2.2 M rows) differ between ANY two summation orders by
    1e-4 .. 1e-2 of the largest entry, so the fused path and the composed path are each compared with a float64 restatement and the fused one
    must not be further from it than the composed one (tools/tri_dbg.py prints both)."""
    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    monkeypatch.setenv("ASD_TRI_CHUNK", "100000")
    g = torch.Generator().manual_seed(6)
    torch.manual_seed(6)
    geo = find("Triplane-transformer-sdf")(dict(_SAMPLED_COMMON, space_generator_config=dict(_TRI_GEN))).cuda()
    geo.do_update_step(0, 0)
    cache = (torch.randn(2, 3, 32, 64, 64, generator=g) * 0.5).cuda()
    n = 270_001
    pts = (torch.rand(2, n, 3, generator=g) * 4.4 - 2.2).cuda()
    keys = ("sdf", "features", "normal", "sdf_grad")
    gs = {k: torch.randn(2 * n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}

    def run(fused):
        monkeypatch.setenv("ASD_TRIFIELD", "1" if fused else "0")
        for p in geo.parameters():
            p.grad = None
        c = cache.clone().requires_grad_(True)
        out = geo(pts, c, output_normal=True)
        sum((out[k] * gs[k]).sum() for k in keys).backward()
        return {k: out[k].detach() for k in keys}, c.grad, [p.grad.clone() for p in geo._heads_weights()]

    o1, c1, h1 = run(True)
    o0, c0, h0 = run(False)
    oref, cref, href = _triplane_field_float64(geo, pts, cache, gs, keys)
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
    for k in ("sdf", "features"):
        assert l2(o1[k], oref[k]) < 1e-5, k
    assert l2(o1["sdf_grad"], oref["sdf_grad"]) < max(2 * l2(o0["sdf_grad"], oref["sdf_grad"]), 1e-4)
    # gradients: besides the order of the sums, every ReLU pre-activation within fp32 rounding of zero switches one row's contribution against
    # float64 — in either path, at different rows (tests/test_gpu_trifield.py) — so the bound is what a handful of such rows cost (a chunk dropped
    # or doubled would be 1e-1)
    e1, e0 = l2(c1, cref), l2(c0, cref)
    print(f"planes gradient vs float64: fused {e1:.2e}, composed {e0:.2e}")
    assert e1 < max(4 * e0, 5e-3)
    for a, b, r in zip(h1, h0, href):
        assert l2(a, r) < max(4 * l2(b, r), 5e-3)


@pytest.mark.parametrize("kind", ["voxel", "triplane"])
def test_fused_field_several_evaluations_of_one_cache_share_one_gradient_buffer(kind, monkeypatch):
    """The VolSDF renderer evaluates the field several times per step (proposal sdf, shading samples, optionally chunk by chunk): every fused
    evaluation is its own autograd node, all of them accumulate into ONE gradient buffer per cache and backward pass (sampled_geometry._GradSlot).
    Three forward() chunks + one forward_sdf() of the same two-entry cache in one graph must give the cache gradient of ONE evaluation of all the
    points (the kernels are linear in the upstream gradients), and a second backward pass must not see the first one's buffer."""
    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    g = torch.Generator().manual_seed(9)
    torch.manual_seed(9)
    if kind == "voxel":
        geo = find("3DConv-net")(dict(_SAMPLED_COMMON, space_generator_config=dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16,
                                                                                  img_channels=32, channel_multiplier=1)))
        cache0 = torch.randn(2, 32, 16, 16, 16, generator=g) * 0.5
    else:
        geo = find("Triplane-transformer-sdf")(dict(_SAMPLED_COMMON, space_generator_config=dict(_TRI_GEN)))
        cache0 = torch.randn(2, 3, 32, 64, 64, generator=g) * 0.5
    geo = geo.cuda()
    geo.do_update_step(0, 0)
    n = 900
    pts = (torch.rand(2, n, 3, generator=g) * 3.6 - 1.8).cuda()
    pts_sdf = (torch.rand(2, 300, 3, generator=g) * 3.6 - 1.8).cuda()
    gs = {k: torch.randn(2 * n, d, generator=g).cuda() for k, d in (("sdf", 1), ("features", 3), ("normal", 3), ("sdf_grad", 3))}
    g_sdf = torch.randn(2, 300, 1, generator=g).cuda()

    def loss_of(c, chunks):
        total = 0.0
        bounds = [0, n] if chunks == 1 else [0, 250, 600, n]
        for a, b in zip(bounds[:-1], bounds[1:]):
            out = geo(pts[:, a:b].contiguous(), c, output_normal=True)
            for k in gs:
                gk = gs[k].view(2, n, -1)[:, a:b].reshape(2 * (b - a), -1)
                total = total + (out[k] * gk).sum()
        # two consumers of another kind on the same cache — one created before the field nodes (runs last in the backward pass), one after
        # (runs first): their gradients must meet the shared buffer at the cache's own edge, after the last scatter (round-5 advisor finding)
        return (c * c).sum() * 0.05 + total + (geo.forward_sdf(pts_sdf, c) * g_sdf).sum() + (c.roll(1, -1) - c).abs().sum() * 0.02

    def grads(chunks):
        for p in geo.parameters():
            p.grad = None
        c = cache0.clone().cuda().requires_grad_(True)
        loss_of(c, chunks).backward()
        return c.grad.clone(), {k: p.grad.clone() for k, p in geo.named_parameters() if p.grad is not None}

    c1, h1 = grads(1)
    c3, h3 = grads(3)
    c3b, _ = grads(3)           # a fresh graph and backward pass: nothing of the previous pass's buffer may leak in
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    assert rel(c3, c1) < 1e-5 and rel(c3b, c1) < 1e-5, (rel(c3, c1), rel(c3b, c1))
    for k in h1:
        assert rel(h3[k], h1[k]) < 1e-4, k
