"""Host-side mirror of the reference's interfaces: registry, structured configs, schedules, camera sampler,
module / state-dict layout.  No GPU."""
import math

import numpy as np

import pytest
import torch


def test_registry_semantics():
    from scaledreamer_amd.registry import __modules__, find, register
    import scaledreamer_amd.plugins  # noqa: F401

    for name in ["implicit-volume", "no-material", "neural-environment-map-background", "nerf-volume-renderer",
                 "stable-diffusion-asynchronous-score-distillation-guidance", "random-camera-datamodule", "scaledreamer-system"]:
        assert name in __modules__
    with pytest.raises(ValueError):
        register("implicit-volume")(object)
    mixed = find("no-material:implicit-volume")
    assert mixed.__mro__[1] is __modules__["implicit-volume"]


def test_structured_config_rejects_unknown_keys_and_schedules():
    from scaledreamer_amd.config import C, parse_structured
    from scaledreamer_amd.renderer import NeRFVolumeRenderer

    with pytest.raises(KeyError):
        parse_structured(NeRFVolumeRenderer.Config, {"radius": 1.0, "not_a_field": 3})
    cfg = parse_structured(NeRFVolumeRenderer.Config, {"radius": 2.0})
    assert cfg.radius == 2.0 and cfg.num_samples_per_ray == 512 and cfg.estimator == "occgrid"
    assert C([0, 0.5, 0.02, 25000], 0, 12500) == pytest.approx(0.26)
    assert C([10000, 0.0, 100.0, 10001], 0, 9999) == 0.0 and C([10000, 0.0, 100.0, 10001], 0, 10001) == 100.0
    assert C(3.5, 0, 0) == 3.5


def test_yaml_loader_resolves_interpolations():
    from scaledreamer_amd.config import load_config

    y = """
name: x
tag: "${rmspace:${system.prompt},_}"
system:
  prompt: "a b c"
  geometry: {radius: 1.5}
  renderer: {radius: "${system.geometry.radius}"}
trainer: {max_steps: 100}
checkpoint: {every: "${trainer.max_steps}"}
"""
    cfg = load_config(y, from_string=True, cli_args=["system.geometry.radius=2.0"])
    assert cfg.tag == "a_b_c" and cfg.system.renderer.radius == 2.0 and cfg.checkpoint.every == 100
    with pytest.raises(ValueError):
        load_config("a: ???", from_string=True)


def test_camera_batch_matches_reference_layout():
    from scaledreamer_amd import presets
    from scaledreamer_amd.data import RandomCameraIterableDataset

    from oracle import oracle as O

    def host_batch(ds):
        """stages 1-2 are host logic; the rays (a HIP kernel in the product) come from the oracle's restatement here"""
        b = ds.cameras()
        ro, rd = O.generate_rays(b["c2w"].numpy(), b["focal_length"].numpy(), ds.height, ds.width, True)
        b["rays_o"], b["rays_d"] = torch.from_numpy(ro), torch.from_numpy(rd)
        return b

    torch.manual_seed(0)
    ds = RandomCameraIterableDataset(presets.asd_sd_nerf()["data"])
    b = host_batch(ds)
    assert b["rays_o"].shape == (1, 64, 64, 3) and b["rays_d"].shape == (1, 64, 64, 3) and b["c2w"].shape == (1, 4, 4)
    torch.testing.assert_close(b["rays_d"].norm(dim=-1), torch.ones(1, 64, 64))
    assert -10 <= float(b["elevation"]) <= 45 and 1.0 <= float(b["camera_distances"]) <= 1.5
    torch.testing.assert_close(b["rays_o"][0, 0, 0], b["camera_positions"][0])
    # the camera looks at the origin: the central ray passes within half a pixel of it
    mid = b["rays_d"][0, 31:33, 31:33].mean(dim=(0, 1))
    d = torch.nn.functional.normalize(-b["camera_positions"][0], dim=0)
    assert float((torch.nn.functional.normalize(mid, dim=0) * d).sum()) > 0.999
    ds.update_step(0, 10000)
    assert host_batch(ds)["rays_o"].shape == (1, 256, 256, 3)              # resolution milestone (asd_sd_nerf.yaml:11-13)
    if not torch.cuda.is_available():
        from scaledreamer_amd._lib import AsdError
        with pytest.raises((AsdError, RuntimeError, AssertionError)):      # the rays are a device kernel: no CPU fallback
            ds.collate()


def test_module_and_state_dict_layout_is_the_references():
    from scaledreamer_amd import presets
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    sysc = presets.asd_sd_nerf()["system"]
    geo = find("implicit-volume")(sysc["geometry"])
    bg = find("neural-environment-map-background")(sysc["background"])
    mat = find("no-material")(sysc["material"])
    ren = find("nerf-volume-renderer")(sysc["renderer"], geometry=geo, material=mat, background=bg)
    assert set(geo.state_dict()) == {"bbox", "encoding.encoding.encoding.params", "density_network.layers.0.weight",
                                     "density_network.layers.2.weight", "feature_network.layers.0.weight",
                                     "feature_network.layers.2.weight"}
    assert geo.encoding.encoding.encoding.params.numel() == 12_599_920
    assert bg.encoding.encoding.encoding.params.numel() == 1_581_184
    assert set(bg.state_dict()) == {"encoding.encoding.encoding.params", "network.layers.0.weight", "network.layers.2.weight",
                                    "network.layers.4.weight"}
    assert {"bbox", "estimator.occs", "estimator.binaries", "estimator.aabbs", "estimator.resolution"} <= set(ren.state_dict())
    assert sum(p.numel() for p in geo.parameters()) + sum(p.numel() for p in bg.parameters()) == 14_185_888  # 56.7 MB all-reduce payload
    assert ren.render_step_size == pytest.approx(1.732 * 2 / 512)
    with pytest.raises(NotImplementedError):
        find("nerf-volume-renderer")({"estimator": "proposal"}, geometry=geo, material=mat, background=bg)


def test_hip_path_fails_loudly_without_a_gpu():
    """No CPU fallback: asking the HIP ops for CPU tensors raises instead of silently computing elsewhere."""
    from scaledreamer_amd import _lib, ops

    m = _lib.make_grid_meta(16, 2, 19, 16, 1.447269237440378)
    with pytest.raises(_lib.AsdError):
        ops.hashgrid_fwd(m, torch.zeros(m.n_params), torch.zeros(4, 3))


@pytest.mark.parametrize("golden,name,extra", [
    ("camera_sv", "random-camera-datamodule", {}),
    ("camera_mv", "mvdream-random-multiview-camera-datamodule", {}),
    ("camera_mv_magic3d", "mvdream-random-multiview-camera-datamodule",
     dict(light_sample_strategy="magic3d", camera_perturb=0.1, center_perturb=0.2, up_perturb=0.02, zoom_range=[0.8, 1.0])),
])
def test_camera_batches_match_reference_collate(golden, name, extra):
    """a1: the product datamodules draw from torch / random in the reference's order (tests/golden/make_goldens_camera.py); the rays
    are checked through the oracle's orc_generate_rays here and through the HIP kernel in tests/test_gpu_renderer_kernels.py."""
    import os
    import random

    import scaledreamer_amd.data  # noqa: F401
    from scaledreamer_amd.registry import find

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", golden + ".npz"))
    sv = dict(batch_size=[2, 1], width=[16, 32], height=[16, 32], resolution_milestones=[10000], camera_distance_range=[1.0, 1.5],
              fovy_range=[40, 70], elevation_range=[-10, 45], camera_perturb=0.0, center_perturb=0.0, up_perturb=0.0,
              eval_camera_distance=1.2, eval_fovy_deg=70.0, n_val_views=30)
    mv = dict(batch_size=[8, 4], n_view=4, width=[16, 32], height=[16, 32], resolution_milestones=[10000],
              camera_distance_range=[0.8, 1.0], fovy_range=[15, 60], elevation_range=[0, 30], camera_perturb=0.0,
              center_perturb=0.0, up_perturb=0.0, eval_camera_distance=3.0, eval_fovy_deg=40.0, n_val_views=30)
    cfg = dict(sv if golden == "camera_sv" else mv)
    cfg.update(extra)
    for s in g["seeds"].tolist():
        ds = find(name)(cfg)
        torch.manual_seed(s)
        random.seed(s)
        b = ds.cameras()
        from oracle import oracle as O
        ro, rd = O.generate_rays(b["c2w"].numpy(), b["focal_length"].numpy(), ds.height, ds.width, True)
        b["rays_o"], b["rays_d"] = torch.from_numpy(ro), torch.from_numpy(rd)
        for k in ["rays_o", "rays_d", "mvp_mtx", "camera_positions", "c2w", "light_positions", "elevation", "azimuth",
                  "camera_distances", "fovy"]:
            np.testing.assert_allclose(b[k].numpy(), g[f"s{s}.{k}"], rtol=2e-6, atol=2e-6, err_msg=f"{golden} seed {s} key {k}")


def test_multiprompt_utils_reduce_to_single_prompt_utils():
    """MultiPromptProcessorOutput (custom/.../prompt_processors/base.py:434-560) with the same view-dependent embeddings for
    every batch element must assemble exactly what the single-prompt PromptProcessorOutput (pinned by its golden) does."""
    from scaledreamer_amd.guidance import PromptUtils
    from scaledreamer_amd.multiprompt import MultiPromptUtils, SyntheticMultiPromptProcessor

    g = torch.Generator().manual_seed(3)
    vd, un = torch.randn(4, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g).expand(4, -1, -1).contiguous()
    single = PromptUtils(vd, un, vd[0], un[0], front_threshold=30.0, back_threshold=30.0)
    B = 5
    multi = MultiPromptUtils([vd[0, 0]] * B, [vd[0]] * B, un[0], [vd] * B, un, front_threshold=30.0, back_threshold=30.0)
    ele = torch.tensor([5.0, 70.0, 20.0, -5.0, 30.0])
    azi = torch.tensor([10.0, 50.0, 100.0, -170.0, -60.0])
    dist = torch.ones(B)
    t1, w1 = single.get_text_embeddings_perp_neg(ele, azi, dist, True)
    t2, w2 = multi.get_text_embeddings_perp_neg(ele, azi, dist, True)
    assert torch.equal(t1, t2) and torch.allclose(w1, w2)
    assert torch.equal(single.get_text_embeddings(ele, azi, dist, True), multi.get_text_embeddings(ele, azi, dist, True))
    proc = SyntheticMultiPromptProcessor(["a", "b", "c"], seed=1, ctx_dim=64)
    pu = proc(prompt=["c", "a"])
    assert pu.get_global_text_embeddings().shape == (2, 64) and torch.equal(pu.text_embeddings_vd[1], proc.table["a"][2])
    with pytest.raises(ValueError):
        proc(prompt="zzz")


def test_multiprompt_datamodule_shards_library_by_rank():
    from scaledreamer_amd.multiprompt import MultipromptRandomCameraIterableDataset as D

    lib = {"train": [f"p{i}" for i in range(10)]}
    cfg = dict(batch_size=2, width=8, height=8, dim_gaussian=4, prompt_library=lib)
    d0, d1 = D(cfg, rank=0, n_ranks=4), D(cfg, rank=1, n_ranks=4)
    assert d0.prompt_library == ["p0", "p4", "p8"] and d1.prompt_library == ["p1", "p5", "p9"]
    b = d0.cameras()         # draws + cameras + noise + prompts (the rays are added on the device by collate())
    assert b["noise"].shape == (2, 4) and len(b["prompt"]) == 2 and set(b["prompt"]) <= set(d0.prompt_library)
    assert b["c2w"].shape == (2, 4, 4) and b["focal_length"].shape == (2,)


def adan_cases():
    """the seeded parameters / gradients of tests/golden/adan_steps.npz (two groups with their own lr, four steps, three settings)"""
    import os
    import zlib

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "adan_steps.npz"))
    seed = int(g["seed"])

    def seeded(name, shape):
        gen = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        return torch.randn(shape, generator=gen)
    cases = (("plain", {}), ("clip_wd", dict(max_grad_norm=0.5, weight_decay=0.02)), ("noprox", dict(weight_decay=0.02, no_prox=True)))
    return g, seeded, cases


def test_oracle_adan_matches_reference_optimizer():
    """oracle/adan_ref.py vs the reference's own Adan class (golden); the product's fused kernel is compared on the GPU
    (tests/test_gpu_optimizers.py)"""
    from oracle.adan_ref import adan_step

    g, seeded, cases = adan_cases()
    for tag, kw in cases:
        p1, p2 = seeded("adan.p1", (7, 5)), seeded("adan.p2", (11,))
        groups = [dict(params=[p1], state=[{}], lr=0.01), dict(params=[p2], state=[{}], lr=0.003)]
        for step in range(4):
            groups[0]["grads"], groups[1]["grads"] = [seeded(f"adan.g1.{step}", (7, 5))], [seeded(f"adan.g2.{step}", (11,))]
            adan_step(groups, step + 1, betas=(0.98, 0.92, 0.99), eps=1e-15, **kw)
        np.testing.assert_allclose(p1.numpy(), g[f"{tag}.p1"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(p2.numpy(), g[f"{tag}.p2"], rtol=1e-6, atol=1e-7)


def test_prompt_processor_reads_the_reference_cache_format(tmp_path):
    """md5(f"{model}-{prompt}").pt files of [77, 1024] tensors (prompt_processors/base.py:19-23, 411-420), view-dependent prompt
    strings (base.py:266-297), FileNotFoundError for a missing entry."""
    import hashlib

    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    model, prompt, neg = "pretrained/stable-diffusion-2-1-base", "a DSLR photo of a hamburger", "ugly, blurry"
    strings = [prompt, neg] + [f"{prompt}, {d} view" for d in ("side", "front", "back", "overhead")]
    g = torch.Generator().manual_seed(0)
    table = {s: torch.randn(77, 16, generator=g) for s in strings}
    for s, e in table.items():
        torch.save(e, tmp_path / (hashlib.md5(f"{model}-{s}".encode()).hexdigest() + ".pt"))
    cfg = {"pretrained_model_name_or_path": model, "prompt": prompt, "negative_prompt": neg, "use_perp_neg": True,
           "front_threshold": 30.0, "back_threshold": 30.0}
    pp = find("stable-diffusion-prompt-processor")(cfg, cache_dir=str(tmp_path))
    pu = pp()
    assert pp.prompts_vd == strings[2:] and pu.use_perp_neg and pu.front_threshold == 30.0
    assert torch.equal(pu.text_embeddings_vd.cpu(), torch.stack([table[s] for s in strings[2:]]))
    assert torch.equal(pu.uncond_text_embeddings_vd.cpu(), torch.stack([table[neg]] * 4))
    assert torch.equal(pu.text_embeddings.cpu(), table[prompt][None])
    with pytest.raises(FileNotFoundError):
        find("stable-diffusion-prompt-processor")(dict(cfg, prompt="something never cached"), cache_dir=str(tmp_path))
    # an encoder hook fills the cache in the reference's format
    calls = []
    def enc(prompts):
        calls.append(list(prompts))
        return torch.zeros(len(prompts), 77, 16)
    pp2 = find("stable-diffusion-prompt-processor")(dict(cfg, prompt="a new prompt"), cache_dir=str(tmp_path), encode_fn=enc)
    assert calls and "a new prompt" in calls[0] and neg not in calls[0] and pp2().text_embeddings_vd.shape == (4, 77, 16)


def test_state_dict_keys_match_the_reference_modules():
    """checkpoint compatibility: same keys and shapes as the reference's own modules
    (tests/golden/make_goldens_state_dicts.py)"""
    import json
    import os

    import scaledreamer_amd.plugins  # noqa: F401
    from scaledreamer_amd.registry import find

    with open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")) as f:
        ref = json.load(f)
    enc = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
           "per_level_scale": 1.447269237440378}
    bg4 = {"otype": "HashGrid", "n_features_per_level": 2, "log2_hashmap_size": 19, "n_levels": 4, "base_resolution": 4, "per_level_scale": 4.0}
    hyper = {"c_dim": 1024, "out_dims": {"sdf_weights": [64, 1], "feature_weights": [64, 3]}, "spectral_norm": False, "n_neurons": 64,
             "n_hidden_layers": 1}
    gen3d = dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16, img_channels=32, channel_multiplier=1)
    tri = dict(inner_dim=64, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2, num_heads=4,
               local_text=True, mlp_ratio=4)
    hg = find("Hyper-iNGP")({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.5, "hypernet_config": hyper, "pos_encoding_config": enc})
    hb = find("multiprompt-neural-hashgrid-environment-map-background")({"color_activation": "sigmoid", "pos_encoding_config": dict(enc, per_level_scale=1.0)})
    mods = {
        "implicit-volume": find("implicit-volume")({"radius": 1.0, "normal_type": "finite_difference", "pos_encoding_config": enc}),
        "neural-environment-map-background": find("neural-environment-map-background")({"color_activation": "sigmoid", "random_aug": True, "dir_encoding_config": bg4}),
        "no-material": find("no-material")({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True}),
        "Hyper-iNGP": hg,
        "multiprompt-neural-hashgrid-environment-map-background": hb,
        "generative-space-volsdf-volume-renderer": find("generative-space-volsdf-volume-renderer")(
            {"radius": 2.0, "use_volsdf": True, "trainable_variance": False, "learned_variance_init": 0.340119, "estimator": "importance",
             "num_samples_per_ray": 64, "num_samples_per_ray_importance": 128}, geometry=hg, material=None, background=hb),
        "3DConv-net": find("3DConv-net")({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.8, "space_generator_config": gen3d}),
        "Triplane-transformer-sdf": find("Triplane-transformer-sdf")({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.8, "space_generator_config": tri}),
    }
    for name, m in mods.items():
        got = {k: list(v.shape) for k, v in m.state_dict().items() if not k.startswith("estimator.")}
        assert got == ref[name], (name, sorted(set(got) ^ set(ref[name])))


# ---- a10: the product's own training_step loss assembly vs the reference's (tests/golden/make_goldens_system.py) ----------------
def _a10_out(seed, requires_grad=True):
    import zlib

    def rnd(name, shape):
        g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        return torch.randn(shape, generator=g)
    B, Hh, Ww, n = 1, 8, 8, 300
    o = {
        "comp_rgb": torch.sigmoid(rnd("comp_rgb", (B, Hh, Ww, 3))),
        "comp_normal": rnd("comp_normal", (B, Hh, Ww, 3)) * 0.5,
        "opacity": torch.sigmoid(2.5 * rnd("opacity", (B, Hh, Ww, 1))),
        "z_variance": rnd("z_variance", (B, Hh, Ww, 1)).abs() * 0.05,
        "weights": torch.sigmoid(rnd("weights", (n, 1))) * 0.1,
        "normal": torch.nn.functional.normalize(rnd("normal", (n, 3)), dim=-1),
        "t_dirs": torch.nn.functional.normalize(rnd("t_dirs", (n, 3)), dim=-1),
        "sdf_grad": rnd("sdf_grad", (n, 3)) * 0.7,
        "inv_std": torch.tensor(30.0),
    }
    o["opacity"].view(-1)[:5] = torch.tensor([0.0, 1.0, 0.0004, 0.9997, 0.5])
    o["comp_normal"].view(-1)[3] = float("nan")
    return {k: (v.clone().requires_grad_(True) if requires_grad and k not in ("t_dirs", "inv_std") else v) for k, v in o.items()}


A10_CASES = {
    "asd_sd_nerf_step0": ("coarse", dict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=30, lambda_opaque=[10000, 0.0, 100.0, 10001],
                                         lambda_z_variance=0.0)),
    "asd_sd_nerf_step10001": ("coarse", dict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=30,
                                             lambda_opaque=[10000, 0.0, 100.0, 10001], lambda_z_variance=0.0)),
    "all_terms": ("coarse", dict(lambda_asd=0.5, lambda_orient=[0, 10.0, 1000.0, 1000], lambda_sparsity=3.0, lambda_opaque=2.0,
                                 lambda_z_variance=4.0, lambda_eikonal=7.0)),
    "coarse_geometry": ("coarse+geometry", dict(lambda_asd=1.0, lambda_orient=1.0, lambda_sparsity=1.0, lambda_opaque=0.0,
                                                lambda_z_variance=0.0)),
}


@pytest.mark.parametrize("case", sorted(A10_CASES))
def test_training_step_loss_terms_match_the_reference(case):
    """scaledreamer_amd.system.StableDreamer.training_step on the tensors the reference's StableDreamer.training_step
    (scaledreamer.py:48-170) was run on: scalar loss, every logged value, gradient w.r.t. every renderer output."""
    import os
    from golden_util import GOLDEN_DIR
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    g = dict(np.load(os.path.join(GOLDEN_DIR, "system_training_step.npz")))
    stage, loss_cfg = A10_CASES[case]
    out = _a10_out(int(g["seed"]))

    def guidance(rgb, prompt_utils, rgb_as_latents=False, **batch):
        probe = torch.linspace(-1.0, 2.0, rgb.numel(), dtype=rgb.dtype).view_as(rgb)
        return {"loss_asd": (rgb * probe).sum() + 0.5 * (rgb ** 2).sum(), "grad_norm": rgb.detach().norm(), "min_step": 20, "max_step": 980}

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.cfg = ConfigDict(stage=stage, loss=ConfigDict(loss_cfg))
    s.current_epoch, s.true_global_step = 0, int(g[case + ".step"])
    s.logged = {}
    s.renderer = lambda **batch: dict(out)
    s.guidance, s.prompt_utils = guidance, None
    loss = s.training_step({"elevation": torch.zeros(1)})["loss"]
    loss.backward()
    assert float(loss) == pytest.approx(float(g[case + ".loss"]), rel=1e-6)
    want_logged = {k[len(case) + 5:]: float(v) for k, v in g.items() if k.startswith(case + ".log.")}
    assert sorted(s.logged) == sorted(want_logged)
    for k, v in want_logged.items():
        assert float(s.logged[k]) == pytest.approx(v, rel=1e-6), k
    for k, t in out.items():
        if t.requires_grad:
            want = torch.from_numpy(g[f"{case}.grad.{k}"])
            got = t.grad if t.grad is not None else torch.zeros_like(t)
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-7, msg=k)


def test_training_step_rejects_mesh_stages_and_missing_outputs():
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.current_epoch, s.true_global_step, s.logged = 0, 0, {}
    out = _a10_out(3, requires_grad=False)
    del out["sdf_grad"]
    s.renderer = lambda **batch: dict(out)
    s.guidance, s.prompt_utils = (lambda rgb, pu, **kw: {"loss_asd": rgb.sum()}), None
    s.cfg = ConfigDict(stage="geometry", loss=ConfigDict(lambda_asd=1.0))
    with pytest.raises(ValueError):
        s.training_step({})
    s.cfg = ConfigDict(stage="coarse", loss=ConfigDict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=0.0, lambda_opaque=0.0,
                                                        lambda_z_variance=0.0, lambda_eikonal=1.0))
    with pytest.raises(ValueError, match="sdf is required"):
        s.training_step({})


# ---- frozen-prior checkpoints: diffusers / LDM layouts -> the engine's name-keyed layout, never a silent random prior -----------
def _diffusers_unet_names(cfg):
    """parameter names of diffusers' UNet2DConditionModel for an SD-2.x topology (CrossAttnDownBlock2D x3 + DownBlock2D,
    UNetMidBlock2DCrossAttn, UpBlock2D + CrossAttnUpBlock2D x3), written out from the architecture — not from our mapping."""
    res = ["norm1", "conv1", "time_emb_proj", "norm2", "conv2"]
    tf = ["norm", "proj_in", "proj_out"] + [f"transformer_blocks.0.{k}" for k in
         ["norm1", "norm2", "norm3", "attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v",
          "attn2.to_out.0", "ff.net.0.proj", "ff.net.2"]]
    nobias = ("to_q", "to_k", "to_v")
    names = []

    def add(prefix, leaves, shortcut=False):
        for l in leaves + (["conv_shortcut"] if shortcut else []):
            names.append(f"{prefix}.{l}.weight")
            if not l.endswith(nobias):
                names.append(f"{prefix}.{l}.bias")
    add("time_embedding", ["linear_1", "linear_2"])
    add("", [])
    names += ["conv_in.weight", "conv_in.bias", "conv_norm_out.weight", "conv_norm_out.bias", "conv_out.weight", "conv_out.bias"]
    mult = cfg.channel_mult
    ch = cfg.model_channels
    for lvl in range(4):
        for j in range(2):
            cout = mult[lvl] * cfg.model_channels
            add(f"down_blocks.{lvl}.resnets.{j}", res, shortcut=ch != cout)
            ch = cout
            if lvl < 3:
                add(f"down_blocks.{lvl}.attentions.{j}", tf)
        if lvl < 3:
            add(f"down_blocks.{lvl}.downsamplers.0", ["conv"])
    add("mid_block.resnets.0", res)
    add("mid_block.attentions.0", tf)
    add("mid_block.resnets.1", res)
    for lvl in range(4):
        for j in range(3):
            add(f"up_blocks.{lvl}.resnets.{j}", res, shortcut=True)     # every up ResBlock sees a concatenated input
            if lvl > 0:
                add(f"up_blocks.{lvl}.attentions.{j}", tf)
        if lvl < 3:
            add(f"up_blocks.{lvl}.upsamplers.0", ["conv"])
    return names


def test_diffusers_and_ldm_checkpoints_map_onto_the_engine_layout(tmp_path):
    from safetensors.torch import save_file
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.diffusion import checkpoint as CK, weights as W

    full = W.UNetConfig()
    shapes = W.unet_layout(full)[0]
    mapped = [CK.diffusers_unet_key_to_ldm(k, full) for k in _diffusers_unet_names(full)]
    assert len(set(mapped)) == len(mapped) and set(mapped) == set(shapes)          # a bijection onto the 686-tensor layout
    assert CK.diffusers_unet_key_to_ldm("up_blocks.0.upsamplers.0.conv.weight", full) == "output_blocks.2.1.conv.weight"
    assert CK.diffusers_unet_key_to_ldm("up_blocks.1.upsamplers.0.conv.weight", full) == "output_blocks.5.2.conv.weight"
    assert CK.diffusers_unet_key_to_ldm("down_blocks.2.downsamplers.0.conv.bias", full) == "input_blocks.9.0.op.bias"
    assert CK.diffusers_unet_key_to_ldm("mid_block.resnets.1.time_emb_proj.weight", full) == "middle_block.2.emb_layers.1.weight"
    # file round trip on a narrow model: an LDM checkpoint and a diffusers directory give back the very tensors
    ucfg, vcfg = W.UNetConfig(model_channels=32, num_head_channels=32, context_dim=16), W.VAEConfig(ch=32)
    up, vp = W.gen_params(W.unet_layout(ucfg)[0], 5), W.gen_params(W.vae_encoder_layout(vcfg)[0], 6)
    ldm = {"model.diffusion_model." + k: v for k, v in up.items()}
    ldm.update({"first_stage_model." + k: v for k, v in vp.items()})
    ldm["first_stage_model.decoder.conv_in.weight"] = torch.zeros(1)                # ignored
    torch.save({"state_dict": ldm}, tmp_path / "mv.pt")
    gu, gv, what = CK.resolve_params(ConfigDict(ckpt_path=str(tmp_path / "mv.pt")), ucfg, vcfg)
    assert all(torch.equal(gu[k], up[k]) for k in up) and all(torch.equal(gv[k], vp[k]) for k in vp) and "LDM" in what
    inv = {CK.diffusers_unet_key_to_ldm(k, ucfg): k for k in _diffusers_unet_names(ucfg)}
    (tmp_path / "sd" / "unet").mkdir(parents=True)
    (tmp_path / "sd" / "vae").mkdir()
    save_file({inv[k]: v.contiguous() for k, v in up.items()}, str(tmp_path / "sd" / "unet" / "diffusion_pytorch_model.safetensors"))
    vae_d = {}
    for k, v in vp.items():
        d = k
        if k.startswith("encoder."):
            d = (k.replace("encoder.down.", "encoder.down_blocks.").replace(".block.", ".resnets.").replace(".downsample.conv", ".downsamplers.0.conv")
                  .replace("mid.block_1", "mid_block.resnets.0").replace("mid.block_2", "mid_block.resnets.1").replace("nin_shortcut", "conv_shortcut")
                  .replace("mid.attn_1.norm", "mid_block.attentions.0.group_norm").replace("mid.attn_1.q", "mid_block.attentions.0.to_q")
                  .replace("mid.attn_1.k", "mid_block.attentions.0.to_k").replace("mid.attn_1.v", "mid_block.attentions.0.to_v")
                  .replace("mid.attn_1.proj_out", "mid_block.attentions.0.to_out.0").replace("encoder.norm_out", "encoder.conv_norm_out"))
            if "attentions.0.to_" in d and d.endswith("weight"):
                v = v.reshape(v.shape[0], v.shape[1])                               # diffusers stores these as Linear
        vae_d[d] = v.contiguous()
    vae_d["decoder.conv_in.weight"] = torch.zeros(1)
    save_file(vae_d, str(tmp_path / "sd" / "vae" / "diffusion_pytorch_model.safetensors"))
    gu, gv, what = CK.resolve_params(ConfigDict(pretrained_model_name_or_path=str(tmp_path / "sd")), ucfg, vcfg)
    assert all(torch.equal(gu[k], up[k]) for k in up) and all(torch.equal(gv[k], vp[k]) for k in vp) and "diffusers" in what
    # a hub id is resolved offline against the local Hugging Face cache layout
    import os
    import shutil

    snap = tmp_path / "hub" / "models--stabilityai--stable-diffusion-2-1-base" / "snapshots" / "abc123"
    shutil.copytree(tmp_path / "sd", snap)
    old = os.environ.get("HF_HUB_CACHE")
    os.environ["HF_HUB_CACHE"] = str(tmp_path / "hub")
    try:
        hu, hv, what = CK.resolve_params(ConfigDict(pretrained_model_name_or_path="stabilityai/stable-diffusion-2-1-base"), ucfg, vcfg)
        assert all(torch.equal(hu[k], up[k]) for k in up) and "local hub cache" in what
    finally:
        if old is None:
            del os.environ["HF_HUB_CACHE"]
        else:
            os.environ["HF_HUB_CACHE"] = old
    shutil.rmtree(tmp_path / "hub")
    # nothing on disk: an error unless random weights were asked for explicitly
    with pytest.raises(CK.MissingWeightsError):
        CK.resolve_params(ConfigDict(pretrained_model_name_or_path="stabilityai/stable-diffusion-2-1-base"), ucfg, vcfg)
    assert CK.resolve_params(ConfigDict(pretrained_model_name_or_path="stabilityai/stable-diffusion-2-1-base", allow_random_weights=True),
                             ucfg, vcfg)[:2] == (None, None)
    del ldm["model.diffusion_model.out.2.bias"]
    torch.save(ldm, tmp_path / "broken.pt")
    with pytest.raises(KeyError):
        CK.resolve_params(ConfigDict(ckpt_path=str(tmp_path / "broken.pt")), ucfg, vcfg)


def test_converted_arguments_outlive_the_launch():
    """ops._Keep: every converted temporary stays referenced until the call returns, so two conversions can never
    be handed the same allocator block."""
    from scaledreamer_amd import ops

    k = ops._Keep()
    a, b = torch.arange(12.0).view(3, 4).t(), torch.arange(12.0).view(3, 4).t() * 2     # both need a contiguous copy
    pa, pb = k(a), k(b)
    assert len(k.held) == 2 and pa.value != pb.value and k.held[0].data_ptr() == pa.value
    assert k(None).value in (None, 0) and len(k.held) == 2


def test_lazy_outputs_defer_until_first_access_and_survive_unpacking():
    """renderer.LazyOutputs: the `normal` / `shading_normal` entries of the training output dictionary exist (`in`, keys, len) but are
    produced on first access only — by indexing, .get, .items() or `**out` — and exactly once."""
    from scaledreamer_amd.renderer import LazyOutputs

    calls = []
    out = LazyOutputs({"comp_rgb": 1})
    out.update({"weights": 2})
    out.defer(("normal", "shading_normal"), lambda: (calls.append(1), {"normal": 10, "shading_normal": 11})[1])
    assert "normal" in out and len(out) == 4 and out.pending() == {"normal", "shading_normal"} and not calls
    assert set(out.keys()) == {"comp_rgb", "weights", "normal", "shading_normal"}
    assert out["comp_rgb"] == 1 and out.get("missing") is None and not calls
    assert {**out} == {"comp_rgb": 1, "weights": 2, "normal": 10, "shading_normal": 11} and len(calls) == 1
    assert out["shading_normal"] == 11 and len(calls) == 1 and not out.pending()
    out2 = LazyOutputs({"a": 1})
    out2.defer(("n",), lambda: {"n": 5})
    assert (lambda a, **kw: kw)(**out2) == {"n": 5}


def test_fused_adamw_host_side_accepts_torch_state_and_rejects_typos():
    """optimizers.AdamW without a device: construction, unknown / unsupported keywords, and load_state_dict of a torch.optim state
    (whose param_groups have no `adam_l2`) — the step itself is HIP-only (tests/test_gpu_optimizers.py)."""
    import pytest

    from scaledreamer_amd.optimizers import AdamW

    p = [torch.nn.Parameter(torch.randn(5))]
    with pytest.raises(TypeError):
        AdamW(p, lr=1e-3, weight_decya=0.1)
    with pytest.raises(NotImplementedError):
        AdamW(p, amsgrad=True)
    AdamW(p, foreach=None, fused=None, amsgrad=False)           # torch keywords that change nothing are accepted
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.randn(5))], lr=2e-3)
    ref.param_groups[0]["params"][0].grad = torch.ones(5)
    ref.step()
    ours = AdamW(p, lr=1e-3, adam_l2=True)
    ours.load_state_dict(ref.state_dict())
    assert ours.param_groups[0]["adam_l2"] is True and ours.param_groups[0]["lr"] == 2e-3
    assert int(ours.state[p[0]]["step"]) == 1
    bad = torch.optim.AdamW([torch.nn.Parameter(torch.randn(5))], amsgrad=True)
    with pytest.raises(NotImplementedError):
        AdamW([torch.nn.Parameter(torch.randn(5))]).load_state_dict(bad.state_dict())


def test_stride2_input_gradient_parity_packing_matches_autograd():
    """weights._pack_stride2_dgrad_conv3x3: the Downsample convolution's input gradient as four 2x2 convolutions over the low-resolution
    gradient image (the layout csrc/net.hip feeds to the upsample == 3 kernel), evaluated here in torch against autograd of the
    reference form (model.py:80-85: pad (0,1,0,1), 3x3 / stride 2 / pad 0)"""
    import torch
    import torch.nn.functional as F

    from scaledreamer_amd.diffusion.weights import _pack_stride2_dgrad_conv3x3

    torch.manual_seed(3)
    cin, cout, Hh = 32, 64, 12
    w = torch.randn(cout, cin, 3, 3, dtype=torch.float64)
    x = torch.randn(2, cin, Hh, Hh, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, stride=2)
    dy = torch.randn_like(y)
    (want,) = torch.autograd.grad(y, x, dy)
    w4 = _pack_stride2_dgrad_conv3x3(w.float()).double().view(2, 2, cin, 2, 2, cout)      # [a][b][cin][ty][tx][cout]
    hl = Hh // 2
    dyp = F.pad(dy, (1, 1, 1, 1))                                                          # low-resolution rows -1 .. hl
    got = torch.zeros_like(want)
    for a in (0, 1):
        for b in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    src = dyp[:, :, a + ty:a + ty + hl, b + tx:b + tx + hl]                # rows y - 1 + a + ty of the unpadded image
                    got[:, :, a::2, b::2] += torch.einsum("bohw,io->bihw", src, w4[a, b, :, ty, tx, :])
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)


def test_scheduled_scalars_match_the_reference_values():
    """config.C / config.Schedule (knot table) against values of the reference's C() (threestudio/utils/misc.py:66-101) on 3-, 4-, 6- and
    8-element schedules, int (global step) and float (epoch) clocks, both interpolations — tests/golden/schedule_values.json was written
    by evaluating the reference function in the build container"""
    import json
    import os

    from scaledreamer_amd.config import C, Schedule

    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "schedule_values.json")))
    assert len(rows) > 300
    for spec, epoch, step, interp, want in rows:
        got = C(spec, epoch, step, interp)
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want)), (spec, epoch, step, interp, got, want)
    assert C(0.25, 3, 7) == 0.25 and Schedule([0, 1.0, 0.5, 100, 0.1, 200]).segment(150) == 1
    for bad, err in (([1, 2], AssertionError), ([1, 2, 3, 4, 5], AssertionError), ("x", TypeError)):
        try:
            C(bad, 0, 0)
        except err:
            continue
        raise AssertionError(f"{bad!r} must raise {err.__name__}")
    try:
        C([0, 1.0, 2.0, 10], 0, 5, "cubic")
    except ValueError:
        pass
    else:
        raise AssertionError("unknown interpolation must raise ValueError")


def test_presets_do_not_allow_a_random_prior_unless_asked():
    """ADVICE (round 2): a preset-driven run whose checkpoint path does not resolve must fail, not train against random weights;
    bench / smoke / this test suite switch the synthetic prior on explicitly (presets.ALLOW_RANDOM_WEIGHTS)"""
    from scaledreamer_amd import presets

    saved = presets.ALLOW_RANDOM_WEIGHTS
    try:
        presets.ALLOW_RANDOM_WEIGHTS = False
        for make in (presets.asd_sd_nerf, presets.asd_mv_nerf, presets.asd_sd_hyper_ingp, presets.asd_mv_triplane_transformer):
            assert make()["system"]["guidance"]["allow_random_weights"] is False
        presets.ALLOW_RANDOM_WEIGHTS = True
        assert presets.asd_sd_nerf()["system"]["guidance"]["allow_random_weights"] is True
    finally:
        presets.ALLOW_RANDOM_WEIGHTS = saved


def test_grad_slot_shares_one_buffer_per_backward_pass_cpu():
    """sampled_geometry._GradSlot / _CacheGate on CPU tensors with a stand-in for the fused field nodes: several autograd nodes read one cache
    through ONE gated alias and ACCUMULATE into one gradient buffer (first node of a backward pass hands it to autograd, the others add in place
    and return None; the gate runs after all of them).  The cache gradient must equal the plain sum, for two backward passes over the SAME
    graph and over fresh graphs, with consumers of another kind on the cache created before AND after the field nodes (the engine runs the
    later-created one first: its gradient must not make autograd sum the buffer away before the last scatter), and a foreign consumer of the
    gated alias itself must raise instead of losing gradient silently."""
    import pytest
    import torch

    from scaledreamer_amd.sampled_geometry import _gate_of

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, cache_view, b, weight, slot):
            ctx.save_for_backward(cache_view, weight)
            ctx.b, ctx.slot = b, slot
            return (cache_view[b] * weight).sum(0)

        @staticmethod
        def backward(ctx, g):
            cache_view, weight = ctx.saved_tensors
            buf, first = ctx.slot.acquire(cache_view)
            buf[ctx.b] += weight * g                  # "the scatter kernel accumulates"
            return (buf if first else None), None, None, None

    torch.manual_seed(0)
    base = torch.randn(2, 3, 4, 5)
    ws = [torch.randn(4, 5) for _ in range(5)]

    def reference():
        ref = base.clone().requires_grad_(True)
        t2 = (ref * 0.5).sum()
        for i, w in enumerate(ws):
            t2 = t2 + (ref.permute(0, 2, 3, 1)[i % 2] * w.unsqueeze(-1)).sum()
        (t2 + (ref * ref).sum() * 0.25).backward()
        return ref.grad

    for _ in range(2):                                   # second round: a new graph must not see the first one's buffer
        cache = base.clone().requires_grad_(True)
        total = (cache * 0.5).sum()                      # a consumer created before the field nodes
        for i, w in enumerate(ws):
            gated, slot = _gate_of(cache, cache.permute(0, 2, 3, 1))
            total = total + Node.apply(gated, i % 2, w.unsqueeze(-1).expand(4, 5, 3), slot).sum()
        total = total + (cache * cache).sum() * 0.25     # and one created after them: the engine runs it first
        total.backward(retain_graph=True)
        torch.testing.assert_close(cache.grad, reference())
        assert slot.buf is None                           # released by the gate
        cache.grad = None
        total.backward()                                 # the same graph once more: a fresh buffer, the same gradient
        torch.testing.assert_close(cache.grad, reference())

    cache = base.clone().requires_grad_(True)
    gated, slot = _gate_of(cache, cache.permute(0, 2, 3, 1))
    out = Node.apply(gated, 0, ws[0].unsqueeze(-1).expand(4, 5, 3), slot).sum() + Node.apply(gated, 1, ws[1].unsqueeze(-1).expand(4, 5, 3), slot).sum()
    with pytest.raises(RuntimeError, match="private"):
        (out + (gated * 2.0).sum()).backward()


def test_trainer_keys_of_the_presets_reach_the_system():
    """accumulate_grad_batches of the reference's YAMLs (asd_mv_triplane_transformer_10k.yaml:129: 2 on the 8-GPU node, _1GPU.yaml:131: 8)"""
    from scaledreamer_amd import presets

    with presets.random_weights_allowed():
        assert presets.asd_mv_triplane_transformer(n_gpus=8)["trainer"]["accumulate_grad_batches"] == 2
        assert presets.asd_mv_triplane_transformer(n_gpus=1)["trainer"]["accumulate_grad_batches"] == 8
        assert "accumulate_grad_batches" not in presets.asd_sd_nerf()["trainer"]

    class S:
        pass

    s = presets.apply_trainer(S(), {"trainer": {"accumulate_grad_batches": 4}})
    assert s.accumulate_grad_batches == 4 and presets.apply_trainer(S(), {}).accumulate_grad_batches == 1


def test_multiprompt_perp_neg_selection_form_equals_the_branching_form():
    """MultiPromptUtils._perp_neg_on_device (what device tensors take: no read-back of the angles) evaluated on CPU tensors against the
    branching form of the reference (prompt_processors/base.py:470-533) — every direction class, its boundaries, the 90-degree switch"""
    from scaledreamer_amd.multiprompt import SyntheticMultiPromptProcessor, _perp_neg_on_device

    prompts = [f"p{i}" for i in range(12)]
    pu = SyntheticMultiPromptProcessor(prompts, device="cpu")(prompts)
    az = torch.tensor([0.0, 30.0, 45.0, -45.0, 89.99, 90.0, 120.0, 135.0, -135.0, 179.0, -100.0, 10.0])
    el = torch.tensor([0.0, 10.0, 59.0, 60.0, 60.01, 75.0, -5.0, 20.0, 30.0, 61.0, 5.0, 89.0])
    for gs in (-1, -3.0):
        want, ww = pu.get_text_embeddings_perp_neg(el, az, None, True, guidance_scale_neg=None if gs == -1 else gs)
        got, w = _perp_neg_on_device(pu, el, az, gs)
        assert torch.equal(got, want) and torch.equal(w, ww)


def test_fused_optimizer_table_matches_the_c_struct():
    """the numpy record the fused optimizers fill per step is include/asd_hip.h's asd_opt_tensor field for field"""
    import ctypes as C

    from scaledreamer_amd._lib import OptTensor
    from scaledreamer_amd.optimizers import _OPT_DTYPE

    assert _OPT_DTYPE.itemsize == C.sizeof(OptTensor)
    for name, _ in OptTensor._fields_:
        assert _OPT_DTYPE.fields[name][1] == getattr(OptTensor, name).offset, name
