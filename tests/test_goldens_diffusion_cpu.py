"""Pins the diffusion oracle (oracle/diffusion_ref.py) and the product's ASD glue against goldens that the
reference's own classes produced in the build container (tests/golden/make_goldens_diffusion.py)."""
import os
import zlib

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR


def rnd(name, shape, seed=0):
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def _unet_cfg(g):
    from scaledreamer_amd.diffusion import weights as W

    mc, hc, cd, cam = (int(v) for v in g["cfg"])
    return W.UNetConfig(model_channels=mc, num_head_channels=hc, context_dim=cd, camera_dim=cam or None,
                        channel_mult=tuple(int(v) for v in g["channel_mult"]),
                        attention_resolutions=tuple(int(v) for v in g["attention_resolutions"]))


def _run_unet(g):
    from oracle import diffusion_ref as D
    from scaledreamer_amd.diffusion import weights as W

    cfg = _unet_cfg(g)
    seed, B, hw, n_ctx = int(g["seed"]), int(g["batch"]), int(g["hw"]), int(g["n_ctx"])
    layout = W.unet_layout(cfg)
    p = W.gen_params(layout[0], seed)
    x = rnd("in.x", (B, cfg.in_channels, hw, hw), seed)
    ctx = rnd("in.context", (B, n_ctx, cfg.context_dim), seed)
    kw = {}
    if cfg.camera_dim is not None:
        kw = dict(camera=rnd("in.camera", (B, cfg.camera_dim), seed), num_frames=int(g["num_frames"]))
    with torch.no_grad():
        return D.unet_forward(p, layout, cfg, x, torch.from_numpy(g["t"]), ctx, **kw).numpy()


@pytest.mark.parametrize("name", ["diffusion_unet_small", "diffusion_mvunet_small"])
def test_oracle_unet_matches_reference_small(name):
    g = _load(name)
    np.testing.assert_allclose(_run_unet(g), g["eps"], rtol=1e-4, atol=2e-5)


def test_oracle_unet_matches_reference_full_sd21_shape():
    # 865 910 724 parameters, 1 x 4 x 64 x 64 latents, 77 x 1024 context: ~25 s on 8 cores
    torch.set_num_threads(os.cpu_count() or 1)
    g = _load("diffusion_unet_sd21_full")
    eps = _run_unet(g)
    assert eps.shape == (1, 4, 64, 64)
    np.testing.assert_allclose(eps, g["eps"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("name", ["diffusion_vae_small", "diffusion_vae_full_256"])
def test_oracle_vae_encoder_forward_and_input_gradient(name):
    from oracle import diffusion_ref as D
    from scaledreamer_amd.diffusion import weights as W

    torch.set_num_threads(os.cpu_count() or 1)
    g = _load(name)
    ch, nrb, zc = (int(v) for v in g["cfg"])
    cfg = W.VAEConfig(ch=ch, num_res_blocks=nrb, z_channels=zc, ch_mult=tuple(int(v) for v in g["ch_mult"]))
    shapes, plan = W.vae_encoder_layout(cfg)
    seed, B, res, st = int(g["seed"]), int(g["batch"]), int(g["res"]), int(g["grad_stride"])
    p = W.gen_params(shapes, seed)
    x = torch.tanh(rnd("in.img", (B, 3, res, res), seed)).requires_grad_(True)
    m = D.vae_encode_moments(p, plan, x)
    np.testing.assert_allclose(m.detach().numpy(), g["moments"], rtol=1e-3, atol=1e-4)
    (m * rnd("in.gmoments", tuple(m.shape), seed)).sum().backward()
    gx = x.grad[:, :, ::st, ::st].numpy()
    scale = float(np.abs(g["grad_x_sub"]).max())
    np.testing.assert_allclose(gx / scale, g["grad_x_sub"] / scale, rtol=0, atol=1e-4)


def test_schedule_matches_reference():
    from oracle import diffusion_ref as D
    from scaledreamer_amd.guidance import ddpm_alphas_cumprod

    ac = _load("diffusion_schedule")["alphas_cumprod"]
    np.testing.assert_allclose(D.alphas_cumprod().numpy(), ac, rtol=2e-6)
    np.testing.assert_allclose(ddpm_alphas_cumprod().numpy(), ac, rtol=2e-6)


def test_prompt_side_perp_neg_matches_reference():
    """PromptUtils.get_text_embeddings_perp_neg (prompt_processors/base.py:82-167) vs the reference's own output."""
    from scaledreamer_amd.guidance import PromptUtils

    g = _load("diffusion_asd_glue")
    seed = int(g["seed"])
    pu = PromptUtils(rnd("prompt.vd", (4, 77, 1024), seed), rnd("prompt.uncond", (1, 77, 1024), seed).expand(4, -1, -1).contiguous(),
                     front_threshold=30.0, back_threshold=30.0)
    elevation, azimuth, dist = (torch.from_numpy(g[k]) for k in ("elevation", "azimuth", "camera_distances"))
    temb, w = pu.get_text_embeddings_perp_neg(elevation, azimuth, dist, True)
    np.testing.assert_allclose(w.numpy(), g["perp_neg_weights"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(temb.mean(dim=2).numpy(), g["text_embeddings_mean"], rtol=1e-5, atol=1e-6)


def test_timestep_window_and_shifted_timestep_match_reference():
    """set_min_max_steps / update_step annealing and get_t_plus (stable_diffusion_asd_guidance.py:131-134,294-316,430-440) are host
    logic: they run without the HIP library (the recorded t / rand of the reference call give its recorded t+)."""
    from scaledreamer_amd.guidance import SDTimestepShiftedScoreDistillationGuidance as G

    g = _load("diffusion_asd_glue")
    guid = G({"guidance_scale": 7.5, "plus_ratio": 0.1, "plus_random": True, "guidance_perp_neg": -0.5, "min_step_percent": 0.5,
              "max_step_percent": 0.98}, backend=object())
    assert (guid.min_step, guid.max_step) == (int(g["min_step"]), int(g["max_step"]))
    guid.rand_fn = lambda shape, device: torch.from_numpy(g["rand"])
    t_plus = guid.get_t_plus(torch.from_numpy(g["t"]))
    np.testing.assert_array_equal(t_plus.numpy().astype(np.float32), g["unet_in_t"][-4:])
    guid.cfg.min_step_percent, guid.cfg.max_step_percent = [0, 0.5, 0.02, 25000], [0, 0.98, 0.5, 25000]   # asd_sd_nerf.yaml:93-94
    guid.update_step(0, 12500)
    assert (guid.min_step, guid.max_step) == (260, 740)
    with pytest.raises(Exception, match="HIP path only"):
        guid(torch.rand(1, 64, 64, 3), type("PU", (), {"use_perp_neg": True, "get_text_embeddings_perp_neg": lambda *a, **k: (torch.zeros(4, 77, 1024), torch.zeros(1, 2))})(),
             torch.zeros(1), torch.zeros(1), torch.ones(1))


# ---- the oracle's ASD glue composition, pinned by the reference's own __call__ (goldens of make_goldens_diffusion.py) ------------
def fake_unet(x, t, ctx, camera=None, num_frames=1):
    """the cheap stand-in networks the golden script handed the reference guidance"""
    s = ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    if camera is not None:
        s = s + camera.mean(dim=1).view(-1, 1, 1, 1)
    return torch.tanh(0.7 * x + 3.0 * s) * (1.0 + t.float().view(-1, 1, 1, 1) / 1000.0) + 0.1 * x.flip(-1)


def fake_encode(imgs):
    pooled = torch.nn.functional.avg_pool2d(imgs, 8)
    z = torch.cat([pooled, pooled.mean(1, keepdim=True) ** 2], dim=1)
    return torch.cat([z, torch.full_like(z, -60.0)], dim=1)    # logvar -> clamp(-30): std ~ 3e-7 (sample ~ mean)


def sd_glue_case():
    from oracle import diffusion_ref as D
    from scaledreamer_amd.guidance import PromptUtils

    g = _load("diffusion_asd_glue")
    seed = int(g["seed"])
    pu = PromptUtils(rnd("prompt.vd", (4, 77, 1024), seed), rnd("prompt.uncond", (1, 77, 1024), seed).expand(4, -1, -1).contiguous(),
                     front_threshold=30.0, back_threshold=30.0)
    el, az, di = (torch.from_numpy(g[k]) for k in ("elevation", "azimuth", "camera_distances"))
    emb, w = pu.get_text_embeddings_perp_neg(el, az, di, True)
    t = torch.from_numpy(g["t"])
    t_plus = D.get_t_plus(t, int(g["min_step"]), 0.1, torch.from_numpy(g["rand"]))
    rgb = torch.sigmoid(rnd("asd.rgb", (4, 64, 64, 3), seed))
    return g, pu, (el, az, di), dict(rgb=rgb, context=torch.cat([emb, emb[:4]], 0), neg_w=w * 0.5, t=t, t_plus=t_plus,
                                     noise=torch.from_numpy(g["noise"]), post_noise=torch.zeros(4, 4, 64, 64))


def test_oracle_asd_glue_matches_reference_call():
    from oracle import diffusion_ref as D

    g, _, _, c = sd_glue_case()
    rgb = c["rgb"].clone().requires_grad_(True)
    loss, norm, io = D.asd_guidance_loss(rgb, fake_encode, fake_unet, c["context"], c["neg_w"], c["t"], c["t_plus"], c["noise"], c["post_noise"],
                                         guidance_scale=7.5, image_size=512)
    np.testing.assert_array_equal(io["t"].numpy().astype(np.float32), g["unet_in_t"])
    np.testing.assert_allclose(io["x"].numpy(), g["unet_in_latents"], rtol=1e-5, atol=1e-5)
    assert abs(loss.item() / float(g["loss_asd"]) - 1) < 1e-5 and abs(norm.item() / float(g["grad_norm"]) - 1) < 1e-5
    loss.backward()
    scale = float(np.abs(g["grad_rgb"]).max())
    np.testing.assert_allclose(rgb.grad.numpy() / scale, g["grad_rgb"] / scale, rtol=0, atol=1e-5)


def mv_glue_case():
    from oracle import diffusion_ref as D
    from scaledreamer_amd.guidance import normalize_camera

    g = _load("diffusion_mvdream_glue")
    seed = int(g["seed"])
    emb, unc = rnd("mv.prompt", (1, 77, 1024), seed), rnd("mv.uncond", (1, 77, 1024), seed)
    context = torch.cat([emb.repeat(4, 1, 1), unc.repeat(4, 1, 1), emb.repeat(4, 1, 1)], 0)
    t = torch.from_numpy(g["t"])
    t_plus = D.get_t_plus(t, 20, 0.1, torch.from_numpy(g["rand"]))
    c2w = torch.from_numpy(g["c2w"])
    rgb = torch.sigmoid(rnd("mv.rgb", (4, 64, 64, 3), seed))
    return g, (emb, unc), dict(rgb=rgb, context=context, neg_w=None, t=t.repeat(4), t_plus=t_plus.repeat(4), noise=torch.from_numpy(g["noise"]),
                               post_noise=torch.zeros(4, 4, 32, 32), camera=normalize_camera(c2w).repeat(3, 1), c2w=c2w)


def test_oracle_mvdream_glue_matches_reference_call():
    from oracle import diffusion_ref as D

    g, _, c = mv_glue_case()
    np.testing.assert_allclose(c["camera"].numpy(), g["unet_in_camera"], rtol=1e-6, atol=1e-7)
    rgb = c["rgb"].clone().requires_grad_(True)
    loss, norm, io = D.asd_guidance_loss(rgb, fake_encode, fake_unet, c["context"], None, c["t"], c["t_plus"], c["noise"], c["post_noise"],
                                         guidance_scale=7.5, image_size=256, unet_kw=dict(camera=c["camera"], num_frames=4))
    np.testing.assert_array_equal(io["t"].numpy(), g["unet_in_t"])
    np.testing.assert_allclose(io["x"].numpy(), g["unet_in_x"], rtol=1e-5, atol=1e-5)
    assert abs(loss.item() / float(g["loss_asd"]) - 1) < 1e-5 and abs(norm.item() / float(g["grad_norm"]) - 1) < 1e-5
    loss.backward()
    scale = float(np.abs(g["grad_rgb"]).max())
    np.testing.assert_allclose(rgb.grad.numpy() / scale, g["grad_rgb"] / scale, rtol=0, atol=1e-5)
