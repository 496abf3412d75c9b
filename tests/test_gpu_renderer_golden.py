"""GPU parity of the drop-in renderer plugins against the goldens produced by the reference's own glue.

The HIP `nerf-volume-renderer` + `implicit-volume` + `no-material` + `neural-environment-map-background`
(looked up through the registry exactly as BaseLift3DSystem.configure does, systems/base.py:292-303) must
reproduce every key of the reference's output dictionary (north_star: RGB / sigma within 1e-3 abs; we hold
1e-5) and the parameter gradients of the reference's loss.
"""
import numpy as np
import pytest
import torch

from golden_util import RENDERER_GOLDENS, load_renderer_golden, reference_loss_torch

pytestmark = pytest.mark.gpu

BG_ENC = {"otype": "HashGrid", "n_features_per_level": 2, "log2_hashmap_size": 19, "n_levels": 4, "base_resolution": 4,
          "per_level_scale": 4.0}


def build_system(g):
    import scaledreamer_amd.plugins  # noqa: F401  (registers the plugin classes)
    from scaledreamer_amd.registry import find

    geo = find("implicit-volume")({"radius": 1.0, "normal_type": "finite_difference"})
    mat = find("no-material")({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True})
    bg = find("neural-environment-map-background")({"color_activation": "sigmoid", "random_aug": True,
                                                    "random_aug_prob": 0.5, "dir_encoding_config": BG_ENC})
    ren = find("nerf-volume-renderer")({"radius": 1.0, "num_samples_per_ray": g["spp"]}, geometry=geo, material=mat,
                                       background=bg)
    sd_geo = {"encoding.encoding.encoding.params": g["grid"], "density_network.layers.0.weight": g["w1d"],
              "density_network.layers.2.weight": g["w2d"], "feature_network.layers.0.weight": g["w1f"],
              "feature_network.layers.2.weight": g["w2f"]}
    sd_bg = {"encoding.encoding.encoding.params": g["bgrid"], "network.layers.0.weight": g["bw0"],
             "network.layers.2.weight": g["bw1"], "network.layers.4.weight": g["bw2"]}
    geo.load_state_dict({k: torch.from_numpy(v) for k, v in sd_geo.items()}, strict=False)
    bg.load_state_dict({k: torch.from_numpy(v) for k, v in sd_bg.items()}, strict=False)
    for m in (geo, mat, bg, ren):
        m.cuda().train()
    # reference checkpoint keys for the occupancy grid
    ren.load_state_dict({"estimator.occs": torch.from_numpy(g["occs"]), "estimator.binaries": torch.from_numpy(g["binaries"])},
                        strict=False)
    jit = torch.from_numpy(g["jitter"]).cuda()
    ren.jitter_fn = lambda n, device: jit
    bg.rand_fn = lambda: 0.9
    return geo, mat, bg, ren


@pytest.mark.parametrize("name", RENDERER_GOLDENS)
def test_hip_renderer_matches_reference_outputs_and_grads(name):
    g = load_renderer_golden(name)
    geo, mat, bg, ren = build_system(g)
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    out = ren(rays_o=dev("rays_o"), rays_d=dev("rays_d"), light_positions=dev("light_positions"),
              elevation=None, azimuth=None)  # extra batch keys arrive via **kwargs and are ignored
    expected = {k[4:] for k in g if k.startswith("out_")}
    assert set(out.keys()) == expected
    n = g["out_weights"].shape[0]
    assert out["weights"].shape == (n, 1) and out["ray_indices"].dtype == torch.int64
    cpu = {k: v.detach().cpu().numpy() for k, v in out.items()}
    np.testing.assert_array_equal(cpu["ray_indices"], g["out_ray_indices"])
    for k in ["t_points", "t_intervals", "points", "t_dirs"]:
        np.testing.assert_array_equal(cpu[k], g["out_" + k], err_msg=k)
    for k in ["density", "features", "weights", "comp_rgb", "comp_rgb_fg", "comp_rgb_bg", "opacity", "depth", "z_variance"]:
        assert cpu[k].shape == g["out_" + k].shape, k
        np.testing.assert_allclose(cpu[k], g["out_" + k], rtol=2e-5, atol=1e-5, err_msg=k)
    np.testing.assert_allclose(cpu["normal"], g["out_normal"], rtol=0, atol=2e-3)
    assert out["shading_normal"] is out["normal"] or torch.equal(out["shading_normal"], out["normal"])

    loss = reference_loss_torch(out, g)
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    got = {"w1d": geo.density_network.layers[0].weight, "w2d": geo.density_network.layers[2].weight,
           "w1f": geo.feature_network.layers[0].weight, "w2f": geo.feature_network.layers[2].weight,
           "bw0": bg.network.layers[0].weight, "bw1": bg.network.layers[2].weight, "bw2": bg.network.layers[4].weight}
    for k, p in got.items():
        ref = g["g_" + k]
        scale = max(float(np.abs(ref).max()), 1e-6)
        np.testing.assert_allclose(p.grad.cpu().numpy() / scale, ref / scale, rtol=0, atol=3e-3, err_msg=k)
    gg = geo.encoding.encoding.encoding.params.grad.cpu().numpy()
    idx, val = g["g_grid_idx"], g["g_grid_val"]
    np.testing.assert_allclose(gg[idx] / np.abs(val).max(), val / np.abs(val).max(), rtol=0, atol=3e-3)
    assert abs(np.linalg.norm(gg.astype(np.float64)) / float(g["g_grid_l2"]) - 1) < 3e-3
    gb = bg.encoding.encoding.encoding.params.grad.cpu().numpy()
    idx, val = g["g_bgrid_idx"], g["g_bgrid_val"]
    np.testing.assert_allclose(gb[idx] / np.abs(val).max(), val / np.abs(val).max(), rtol=0, atol=3e-3)


def test_unread_normals_are_deferred_and_equal_the_eager_ones():
    """C2 sets `requires_normal` on a material that never reads normals and lambda_orient = 0: the finite-difference normal (three
    more encodes per kept sample) must not be evaluated unless somebody indexes it — and when somebody does, it is the same tensor the
    eager call gives, gradients included."""
    g = load_renderer_golden(RENDERER_GOLDENS[0])
    geo, mat, bg, ren = build_system(g)
    assert mat.requires_normal and not mat.reads_normal
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    out = ren(rays_o=dev("rays_o"), rays_d=dev("rays_d"), light_positions=dev("light_positions"))
    assert {"normal", "shading_normal"} <= out.pending() and "normal" in out
    (out["comp_rgb"].sum() + out["opacity"].sum()).backward()
    assert {"normal", "shading_normal"} <= out.pending()          # a step that never reads them never pays for them
    lazy = out["normal"]
    assert not ({"normal", "shading_normal"} & out.pending())
    eager = geo(out["points"], output_normal=True)["normal"]
    assert torch.equal(lazy, eager) and lazy.requires_grad
    p = geo.encoding.encoding.encoding.params
    p.grad = None
    (lazy * out["t_dirs"]).sum().backward()
    g_lazy = p.grad.clone()
    p.grad = None
    (eager * out["t_dirs"]).sum().backward()
    scale = float(p.grad.abs().max())
    assert scale > 0 and float((g_lazy - p.grad).abs().max()) <= 1e-4 * scale      # atomics: summation order only


def test_eval_mode_and_empty_rays():
    g = load_renderer_golden(RENDERER_GOLDENS[0])
    geo, mat, bg, ren = build_system(g)
    ren.eval(); geo.eval(); bg.eval()
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    with torch.no_grad():
        out = ren(rays_o=dev("rays_o"), rays_d=dev("rays_d"), light_positions=dev("light_positions"))
    assert "comp_normal" in out and "weights" not in out
    assert out["comp_rgb"].shape == (1, int(g["h"]), int(g["w"]), 3)
    # rays that miss the box entirely: one dummy sample, background only
    o = torch.full((1, 4, 4, 3), 5.0, device="cuda")
    d = torch.nn.functional.normalize(torch.ones(1, 4, 4, 3, device="cuda"), dim=-1)
    ren.train()
    out = ren(rays_o=o, rays_d=d, light_positions=o[:, 0, 0])
    assert out["weights"].shape == (1, 1) and float(out["opacity"].abs().max()) == 0.0
    torch.testing.assert_close(out["comp_rgb"], out["comp_rgb_bg"])


def test_sync_free_training_pass_equals_the_pass_that_reads_the_count(monkeypatch):
    """renderer.forward keeps the kept-sample count on the device in training (capacity-sized sample tensors, `n_dev` into the field
    kernels, per-sample dictionary entries cut to length on first access): same outputs, same gradients as the pass that reads the
    count back (ASD_SYNC_FREE=0), and no per-sample entry is materialised by a step that only reads per-ray entries."""
    g = load_renderer_golden(RENDERER_GOLDENS[0])
    dev = lambda k: torch.from_numpy(g[k]).cuda()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ASD_SYNC_FREE", mode)
        geo, mat, bg, ren = build_system(g)
        out = ren(rays_o=dev("rays_o"), rays_d=dev("rays_d"), light_positions=dev("light_positions"))
        if mode == "1":
            assert {"weights", "ray_indices", "points", "density", "features"} <= out.pending()
        loss = (out["comp_rgb"] * torch.linspace(0.5, 1.5, 3, device="cuda")).sum() + (out["opacity"] ** 2).sum() + out["z_variance"].sum()
        loss.backward()
        if mode == "1":
            assert {"weights", "ray_indices", "points", "density", "features"} <= out.pending()
        grads = {n: p.grad.detach().clone() for n, p in list(geo.named_parameters()) + [("bg." + n, p) for n, p in bg.named_parameters()]}
        res[mode] = ({k: out[k].detach().clone() for k in out.keys()}, grads, ren.last_n_samples)
    (o0, g0, n0), (o1, g1, n1) = res["0"], res["1"]
    assert n0 == n1 == o0["weights"].shape[0] and set(o0) == set(o1)
    for k in o0:
        assert o0[k].shape == o1[k].shape, k
        if o0[k].dtype == torch.int64:
            assert torch.equal(o0[k], o1[k]), k
        else:
            torch.testing.assert_close(o1[k], o0[k], rtol=1e-6, atol=1e-6, msg=k)
    for k in g0:
        scale = float(g0[k].abs().max())
        assert float((g1[k] - g0[k]).abs().max()) <= 1e-4 * max(scale, 1e-12), k      # atomics: summation order only
