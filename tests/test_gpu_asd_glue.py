"""GPU parity of the fused ASD glue (csrc/asd_glue.hip) and of the product guidance built on it:
  * each kernel against the torch fp32 restatement of its reference lines (oracle/diffusion_ref.py),
  * SDTimestepShiftedScoreDistillationGuidance / MVDream...Guidance end to end against the goldens of the reference's own __call__
    (tests/golden/diffusion_asd_glue.npz, diffusion_mvdream_glue.npz) with the same stand-in networks the golden script used.
Tolerances: the image and the context reach the networks in fp16 (as in the reference's fp16 pipeline), so quantities downstream of
that cast are compared at 2e-3; the glue arithmetic itself is fp32 and is compared at 1e-5 where no fp16 cast intervenes."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_goldens_diffusion_cpu import fake_encode, fake_unet, mv_glue_case, sd_glue_case

pytestmark = pytest.mark.gpu


def _lib():
    from scaledreamer_amd import _lib as L

    return L


@pytest.mark.parametrize("B,h,H", [(1, 64, 512), (2, 64, 256), (1, 48, 200), (1, 256, 512), (1, 64, 64)])
def test_image_prep_matches_interpolate_and_its_adjoint(B, h, H):
    L = _lib()
    torch.manual_seed(0)
    rgb = torch.rand(B, h, h, 3, device="cuda")
    x = torch.empty(B, H, H, 32, device="cuda", dtype=torch.float16)
    L.check(L.lib().asd_image_prep_fwd(L.ptr(rgb), L.i32(B), L.i32(h), L.i32(h), L.i32(H), L.i32(H), L.ptr(x), L.stream()))
    rr = rgb.clone().requires_grad_(True)
    ref = F.interpolate(rr.permute(0, 3, 1, 2), (H, H), mode="bilinear", align_corners=False) * 2.0 - 1.0
    assert float((x[..., :3].float() - ref.permute(0, 2, 3, 1)).abs().max()) < 1.1e-3      # fp16 rounding of values in [-1, 1]
    assert float(x[..., 3:].abs().max()) == 0.0
    dx = torch.zeros(B, H, H, 32, device="cuda", dtype=torch.float16)
    dx[..., :3] = torch.randn(B, H, H, 3, device="cuda")
    dx[..., 3:] = 7.0                                                                           # must be ignored
    d_rgb = torch.empty(B, h, h, 3, device="cuda")
    L.check(L.lib().asd_image_prep_bwd(L.ptr(dx), L.i32(B), L.i32(h), L.i32(h), L.i32(H), L.i32(H), L.ptr(d_rgb), L.stream()))
    ref.backward(dx[..., :3].float().permute(0, 3, 1, 2))
    torch.testing.assert_close(d_rgb, rr.grad, rtol=1e-4, atol=1e-4 * float(rr.grad.abs().max()))


@pytest.mark.parametrize("B,n_neg,weighting,clip", [(4, 2, 0, 0.0), (1, 2, 2, 0.0), (3, 0, 0, 0.0), (2, 0, 1, 0.5), (2, 2, 0, 3.0)])
def test_latents_and_score_kernels_match_the_torch_restatement(B, n_neg, weighting, clip):
    from oracle import diffusion_ref as D

    L = _lib()
    l = L.lib()
    g = torch.Generator().manual_seed(B * 10 + n_neg)
    hl, Cc = 16, 4
    n_rep = 2 + n_neg
    moments = torch.randn(B, hl, hl, 8, generator=g)
    moments[..., 4:] = moments[..., 4:] * 12          # some log-variances beyond the clamp (-30, 20)
    pn, noise = torch.randn(B, Cc, hl, hl, generator=g), torch.randn(B, Cc, hl, hl, generator=g)
    t = torch.randint(20, 980, (B,), generator=g)
    t_plus = (t + torch.randint(0, 19, (B,), generator=g)).clamp(1, 999)
    alphas = D.alphas_cumprod()
    dev = lambda x: x.cuda().contiguous()
    lat = torch.empty(B, Cc, hl, hl, device="cuda")
    ux = torch.full(((n_rep + 1) * B, hl, hl, 32), 9.0, device="cuda", dtype=torch.float16)
    ut = torch.empty((n_rep + 1) * B, device="cuda")
    mo_d, pn_d, no_d, t_d, tp_d, al_d = dev(moments), dev(pn), dev(noise), dev(t), dev(t_plus), dev(alphas)
    L.check(l.asd_latents_fwd(L.ptr(mo_d), L.ptr(pn_d), L.ptr(no_d), L.ptr(t_d), L.ptr(tp_d), L.ptr(al_d), L.i32(B), L.i32(Cc), L.i32(hl), L.i32(hl),
                              L.f32(0.18215), L.i32(n_rep), L.ptr(lat), L.ptr(ux), L.ptr(ut), L.stream()))
    m_nchw = moments.permute(0, 3, 1, 2)
    z = D.sample_posterior(m_nchw, pn)
    torch.testing.assert_close(lat.cpu(), z, rtol=1e-5, atol=1e-6)
    want_x = torch.cat([D.add_noise(alphas, z, noise, t)] * n_rep + [D.add_noise(alphas, z, noise, t_plus)], 0)
    got_x = ux[..., :4].float().cpu().permute(0, 3, 1, 2)
    assert float((got_x - want_x).abs().max()) <= 1e-3 * float(want_x.abs().max()) + 1e-3           # fp16 store
    assert float(ux[..., 4:].abs().max()) == 0.0
    torch.testing.assert_close(ut.cpu(), torch.cat([t] * n_rep + [t_plus]).float())
    # score
    eps = torch.randn((n_rep + 1) * B, hl, hl, Cc, generator=g)
    if B > 1 or n_neg == 0:   # non-finite network outputs must come out finite (nan_to_num); with Perp-Neg they poison the whole sample
        eps[0, 0, 0, 0], eps[0, 0, 0, 1] = float("nan"), float("inf")
    neg_w = torch.randn(B, n_neg, generator=g) if n_neg else None
    grad, scr = torch.empty(B, Cc, hl, hl, device="cuda"), torch.empty(B + 2, device="cuda")
    eps_d = dev(eps)
    nw_d = None if neg_w is None else dev(neg_w)
    L.check(l.asd_score_fwd(L.ptr(eps_d), L.i32(B), L.i32(Cc), L.i32(hl * hl), L.i32(n_neg), L.ptr(nw_d), L.f32(7.5), L.ptr(t_d), L.ptr(al_d), L.i32(weighting),
                            L.f32(clip), L.ptr(grad), L.ptr(scr), L.ptr(scr[B:]), L.stream()))
    e = eps.permute(0, 3, 1, 2)
    first, second = D.asd_eps_aggregate(e, B, 7.5, neg_w)
    a = alphas[t].view(-1, 1, 1, 1)
    w = [1 - a, torch.ones_like(a), a.sqrt() * (1 - a)][weighting]
    ref = torch.nan_to_num(w * (first - second))
    if clip > 0:
        ref = ref.clamp(-clip, clip)
    got = grad.cpu()
    finite = ref.abs() < 1e30
    if n_neg:   # a NaN/inf inside a sample's dot products poisons that whole sample in the reference too (then nan_to_num): compare where defined
        finite &= torch.isfinite(first - second)
    torch.testing.assert_close(got[finite], ref[finite], rtol=2e-4, atol=2e-5 * float(ref[finite].abs().max()))
    assert bool(torch.isfinite(got).all())
    ss = float((got.double() ** 2).sum())
    if ss < 1e30:
        assert abs(float(scr[B]) / (0.5 * ss / B) - 1) < 1e-4 and abs(float(scr[B + 1]) / ss ** 0.5 - 1) < 1e-4
    # backward through the posterior sample
    up = torch.tensor([0.7], device="cuda")
    d_m = torch.empty(B, hl, hl, 8, device="cuda")
    L.check(l.asd_latents_bwd(L.ptr(grad), L.ptr(mo_d), L.ptr(pn_d), L.ptr(up), L.i32(B), L.i32(Cc), L.i32(hl), L.i32(hl), L.f32(0.18215), L.ptr(d_m), L.stream()))
    mm = m_nchw.clone().requires_grad_(True)
    zz = D.sample_posterior(mm, pn)
    zz.backward(got * 0.7 / B)
    ok = torch.isfinite(mm.grad)
    torch.testing.assert_close(d_m.cpu().permute(0, 3, 1, 2)[ok], mm.grad[ok], rtol=1e-4, atol=1e-6 * float(mm.grad[ok].abs().max()) + 1e-12)


class _StandInBackend:
    """the golden script's cheap UNet / VAE, on the GPU, behind the buffer protocol adaptors of DiffusionBackend"""

    def __new__(cls, camera_dim=0):
        from scaledreamer_amd.guidance import DiffusionBackend

        class B(DiffusionBackend):
            def unet(self, x, t, ctx, camera=None, num_frames=1):
                self.calls = dict(x=x.clone(), t=t.clone(), ctx=ctx.clone(), camera=None if camera is None else camera.clone(), nf=num_frames)
                return fake_unet(x, t, ctx, camera)

            def encode(self, imgs):
                return fake_encode(imgs)
        b = B()
        b.camera_dim, b.device = camera_dim, "cuda"
        return b


def test_sd_guidance_matches_reference_call_golden():
    from scaledreamer_amd.guidance import SDTimestepShiftedScoreDistillationGuidance as G

    g, pu, (el, az, di), c = sd_glue_case()
    be = _StandInBackend()
    guid = G({"guidance_scale": 7.5, "plus_ratio": 0.1, "plus_random": True, "guidance_perp_neg": -0.5, "min_step_percent": 0.5,
              "max_step_percent": 0.98}, backend=be)
    guid.posterior_noise_fn = torch.zeros_like
    guid.noise_fn = lambda like: torch.from_numpy(g["noise"]).cuda()
    guid.timestep_fn = lambda lo, hi, n, device: torch.from_numpy(g["t"]).cuda()
    guid.rand_fn = lambda shape, device: torch.from_numpy(g["rand"]).cuda()
    for k in ("text_embeddings_vd", "uncond_text_embeddings_vd", "text_embeddings", "uncond_text_embeddings"):
        if getattr(pu, k) is not None:
            setattr(pu, k, getattr(pu, k).cuda())
    rgb = c["rgb"].cuda().requires_grad_(True)
    out = guid(rgb, pu, el.cuda(), az.cuda(), di.cuda())
    np.testing.assert_array_equal(be.calls["t"].cpu().numpy(), g["unet_in_t"])
    want = g["unet_in_latents"]
    assert float(np.abs(be.calls["x"].cpu().numpy() - want).max()) < 2e-3 * float(np.abs(want).max())
    np.testing.assert_allclose(be.calls["ctx"].mean(dim=2).cpu().numpy(), g["unet_in_ctx_mean"], rtol=0, atol=2e-3)
    assert abs(out["loss_asd"].item() / float(g["loss_asd"]) - 1) < 2e-3 and abs(out["grad_norm"].item() / float(g["grad_norm"]) - 1) < 2e-3
    assert (out["min_step"], out["max_step"]) == (int(g["min_step"]), int(g["max_step"]))
    (out["loss_asd"] * 1.0).backward()
    scale = float(np.abs(g["grad_rgb"]).max())
    np.testing.assert_allclose(rgb.grad.cpu().numpy() / scale, g["grad_rgb"] / scale, rtol=0, atol=3e-3)


def test_mvdream_guidance_matches_reference_call_golden():
    from scaledreamer_amd.guidance import MVDreamTimestepShiftedScoreDistillationGuidance as G, PromptUtils

    g, (emb, unc), c = mv_glue_case()
    be = _StandInBackend(camera_dim=16)
    guid = G({"guidance_scale": 7.5, "plus_ratio": 0.1, "plus_random": True, "n_view": 4}, backend=be)
    assert (guid.min_step, guid.max_step) == (20, 980)
    guid.posterior_noise_fn = torch.zeros_like
    guid.noise_fn = lambda like: torch.from_numpy(g["noise"]).cuda()
    guid.timestep_fn = lambda lo, hi, n, device: torch.from_numpy(g["t"]).cuda()
    guid.rand_fn = lambda shape, device: torch.from_numpy(g["rand"]).cuda()
    pu = PromptUtils(emb.expand(4, -1, -1).cuda(), unc.expand(4, -1, -1).cuda(), emb.cuda(), unc.cuda(), use_perp_neg=False)
    el, az, di = (torch.from_numpy(g[k]).cuda() for k in ("elevation", "azimuth", "camera_distances"))
    rgb = c["rgb"].cuda().requires_grad_(True)
    out = guid(rgb, pu, el, az, di, c["c2w"].cuda())
    assert be.calls["nf"] == int(g["num_frames"]) == 4
    np.testing.assert_array_equal(be.calls["t"].cpu().numpy(), g["unet_in_t"].astype(np.float32))
    np.testing.assert_allclose(be.calls["camera"].cpu().numpy(), g["unet_in_camera"], rtol=0, atol=1e-3)
    want = g["unet_in_x"]
    assert float(np.abs(be.calls["x"].cpu().numpy() - want).max()) < 2e-3 * float(np.abs(want).max())
    assert abs(out["loss_asd"].item() / float(g["loss_asd"]) - 1) < 2e-3 and abs(out["grad_norm"].item() / float(g["grad_norm"]) - 1) < 2e-3
    # this golden's loss is O(1): the stand-in encoder's image gradient (~1e-6 per pixel) would sit in fp16's subnormals on its way
    # through the fp16 dx buffer, so the upstream gradient is scaled up (and the result down) — it is a linear map
    (out["loss_asd"] * 4096.0).backward()
    scale = float(np.abs(g["grad_rgb"]).max())
    # |first - second| is ~0.05 rms in this golden while the UNet input reaches the network rounded to fp16 (5e-4 relative) and CFG
    # multiplies the resulting eps error by 7.5: per-pixel deviations of ~1 % of the largest gradient are that rounding, not the glue
    got = rgb.grad.cpu().numpy() / 4096.0 / scale
    np.testing.assert_allclose(got, g["grad_rgb"] / scale, rtol=0, atol=1.5e-2)
    assert float(np.sqrt(np.mean((got - g["grad_rgb"] / scale) ** 2))) < 2e-3
    with pytest.raises(NotImplementedError):
        guid(rgb, pu, el, az, di, None)


@pytest.mark.parametrize("perp_neg", [True, False])
def test_prompt_context_kernel_matches_the_prompt_processor(perp_neg):
    """asd_prompt_context vs PromptUtils.get_text_embeddings[_perp_neg] (pinned against the reference's own outputs in
    tests/test_goldens_diffusion_cpu.py) over every branch: overhead, front / side / back, both azimuth signs, wrap-around at +-180"""
    import ctypes as C

    from scaledreamer_amd._lib import check, f32, i32, lib, ptr, stream
    from scaledreamer_amd.guidance import PromptUtils

    dev = torch.device("cuda", 0)
    pu = PromptUtils.synthetic(seed=5)
    el = torch.tensor([10.0, 75.0, 0.0, -20.0, 30.0, 61.0, 15.0, 5.0, 59.9, 12.0, 3.0])
    az = torch.tensor([0.0, 10.0, 44.0, -60.0, 100.0, 170.0, -179.0, 200.0, -100.0, 89.5, -400.0])
    B, n_tok, dim = el.shape[0], 77, 1024
    stride = 80
    n_rep = 4 if perp_neg else 2
    ctx = torch.zeros(((n_rep + 1) * B * stride, dim), device=dev, dtype=torch.float16)
    w = torch.full((B, 2), 7.0, device=dev)
    params = [pu.overhead_threshold, pu.front_threshold, pu.back_threshold, *pu.perp_neg_f_sb, *pu.perp_neg_f_fsb, *pu.perp_neg_f_fs, *pu.perp_neg_f_sf]
    text, unc = pu.text_embeddings_vd.to(dev).contiguous(), pu.uncond_text_embeddings_vd.to(dev).contiguous()
    el_d, az_d = el.to(dev), az.to(dev)               # named: a temporary would be freed (and its block reused) before the launch
    check(lib().asd_prompt_context(ptr(text), ptr(unc), i32(4), i32(n_tok), i32(dim), ptr(el_d), ptr(az_d), i32(B), i32(int(perp_neg)),
                                   (C.c_float * 15)(*params), f32(-0.5), ptr(ctx), i32(stride), ptr(w) if perp_neg else None, stream()))
    got = ctx.view((n_rep + 1) * B, stride, dim)
    assert float(got[:, n_tok:].abs().max()) == 0.0                      # padding rows untouched
    if perp_neg:
        emb, wr = pu.get_text_embeddings_perp_neg(el, az, torch.ones(B), True)
        want = torch.cat([emb, emb[:B]], 0)
        torch.testing.assert_close(w.cpu(), wr * -0.5, rtol=2e-6, atol=1e-7)
    else:
        emb = pu.get_text_embeddings(el, az, torch.ones(B), True)
        want = torch.cat([emb, emb[:B]], 0)
    assert torch.equal(got[:, :n_tok].cpu(), want.half())                # same fp32 blend, same rounding
    # no view dependence: one embedding for every camera
    ctx.zero_()
    t1, u1 = pu.text_embeddings[None].to(dev).contiguous(), pu.uncond_text_embeddings[None].to(dev).contiguous()
    check(lib().asd_prompt_context(ptr(t1), ptr(u1), i32(1), i32(n_tok), i32(dim), ptr(el_d), ptr(az_d), i32(B), i32(0),
                                   (C.c_float * 15)(*params), f32(0.0), ptr(ctx), i32(stride), None, stream()))
    emb = pu.get_text_embeddings(el, az, torch.ones(B), False)
    assert torch.equal(ctx[:3 * B * stride].view(3 * B, stride, dim)[:, :n_tok].cpu(), torch.cat([emb, emb[:B]], 0).half())


@pytest.mark.parametrize("case", ["asd_sd_nerf_step0", "asd_sd_nerf_step10001", "all_terms", "coarse_geometry"])
def test_training_step_loss_assembly_on_the_device_matches_the_reference(case):
    """StableDreamer.training_step with CUDA tensors — the weighted terms and the per-ray regularisers go through asd_loss_tail_fwd / _bwd
    (one launch each way) — against the values the reference's training_step (scaledreamer.py:48-170) produced on the same tensors
    (tests/golden/system_training_step.npz): loss, every logged value, the gradient w.r.t. every renderer output."""
    import os

    import numpy as np
    from golden_util import GOLDEN_DIR
    from test_host_logic_cpu import A10_CASES, _a10_out

    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    g = dict(np.load(os.path.join(GOLDEN_DIR, "system_training_step.npz")))
    stage, loss_cfg = A10_CASES[case]
    out = {k: (v.detach().cuda().requires_grad_(v.requires_grad) if torch.is_tensor(v) else v) for k, v in _a10_out(int(g["seed"])).items()}

    def guidance(rgb, prompt_utils, rgb_as_latents=False, **batch):
        probe = torch.linspace(-1.0, 2.0, rgb.numel(), dtype=rgb.dtype).view_as(rgb).to(rgb.device)
        return {"loss_asd": (rgb * probe).sum() + 0.5 * (rgb ** 2).sum(), "grad_norm": rgb.detach().norm(), "min_step": 20, "max_step": 980}

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.cfg = ConfigDict(stage=stage, loss=ConfigDict(loss_cfg))
    s.current_epoch, s.true_global_step = 0, int(g[case + ".step"])
    s.logged = {}
    s.renderer = lambda **batch: dict(out)
    s.guidance, s.prompt_utils = guidance, None
    loss = s.training_step({"elevation": torch.zeros(1)})["loss"]
    assert type(loss.grad_fn).__name__ == "_LossTailFnBackward"
    loss.backward()
    assert float(loss) == pytest.approx(float(g[case + ".loss"]), rel=2e-6)
    want_logged = {k[len(case) + 5:]: float(v) for k, v in g.items() if k.startswith(case + ".log.")}
    assert sorted(s.logged) == sorted(want_logged)
    for k, v in want_logged.items():
        assert float(s.logged[k]) == pytest.approx(v, rel=2e-6), k
    for k, t in out.items():
        if torch.is_tensor(t) and t.requires_grad:
            want = torch.from_numpy(g[f"{case}.grad.{k}"])
            got = t.grad.cpu() if t.grad is not None else torch.zeros_like(want)
            torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-7, msg=k)


def test_loss_tail_z_variance_without_a_ray_above_one_half_is_nan_like_the_empty_mean():
    from scaledreamer_amd.system import _LossTailFn

    op = torch.full((1, 8, 8, 1), 0.25, device="cuda", requires_grad=True)
    zv = torch.rand(1, 8, 8, 1, device="cuda", requires_grad=True)
    total, values = _LossTailFn.apply(op, zv, (0.0, 0.0, 2.0), ())
    assert torch.isnan(total) and torch.isnan(values[2])
    assert torch.isnan(zv[op > 0.5].mean())                            # what the tensor-op form gives


def test_timestep_plus_kernel_matches_the_tensor_op_form():
    """asd_timestep_plus (one launch) against get_t_plus' tensor-op form evaluated on the CPU (stable_diffusion_asd_guidance.py:294-316): every
    timestep of the schedule x several uniform draws incl. 0 and the largest float below 1, with and without plus_random — bit-exact"""
    from scaledreamer_amd.guidance import _AsdGuidanceBase

    class G(_AsdGuidanceBase):
        def __init__(self, plus_random, min_step, u):
            self.cfg = type("Cfg", (), {"plus_ratio": 0.1, "plus_random": plus_random})()
            self.num_train_timesteps, self.min_step = 1000, min_step
            self.rand_fn = lambda shape, device: u.to(device)

    T = 1000
    t = torch.arange(0, T, dtype=torch.long).repeat(7)
    g = torch.Generator().manual_seed(5)
    u = torch.rand(t.shape, generator=g)
    u[:T] = 0.0
    u[T:2 * T] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    for plus_random in (False, True):
        for min_step in (20, 500, 979):
            want = G(plus_random, min_step, u).get_t_plus(t)                       # CPU tensors: the tensor-op form
            got = G(plus_random, min_step, u).get_t_plus(t.cuda())                 # device tensors: the kernel
            assert got.dtype == torch.long and torch.equal(got.cpu(), want), (plus_random, min_step)
    # other plus_ratio values round differently in float32: sweep a few
    for ratio in (0.0, 0.05, 0.3, 1.0, 2.5):
        a, b = G(True, 20, u), G(True, 20, u)
        a.cfg.plus_ratio = b.cfg.plus_ratio = ratio
        assert torch.equal(b.get_t_plus(t.cuda()).cpu(), a.get_t_plus(t)), ratio


def test_multiprompt_perp_neg_conditioning_without_read_back_is_bit_identical():
    """MultiPromptUtils.get_text_embeddings_perp_neg on device tensors (selections instead of per-element host branches) against the
    branching form (the reference's, prompt_processors/base.py:470-533) evaluated on the same values: every direction class, the
    class boundaries, the 90-degree switch between the two blends"""
    from scaledreamer_amd.multiprompt import SyntheticMultiPromptProcessor

    prompts = [f"p{i}" for i in range(12)]
    pu_dev = SyntheticMultiPromptProcessor(prompts, device="cuda")(prompts)
    pu_cpu = SyntheticMultiPromptProcessor(prompts, device="cpu")(prompts)
    az = torch.tensor([0.0, 30.0, 45.0, -45.0, 89.99, 90.0, 120.0, 135.0, -135.0, 179.0, -100.0, 10.0])
    el = torch.tensor([0.0, 10.0, 59.0, 60.0, 60.01, 75.0, -5.0, 20.0, 30.0, 61.0, 5.0, 89.0])
    got, w = pu_dev.get_text_embeddings_perp_neg(el.cuda(), az.cuda(), torch.ones(12).cuda(), True)
    want, ww = pu_cpu.get_text_embeddings_perp_neg(el, az, torch.ones(12), True)
    assert torch.equal(got.cpu(), want)
    torch.testing.assert_close(w.cpu(), ww, rtol=1e-6, atol=1e-7)           # exp() of the host's libm vs the device's
    got2, w2 = pu_dev.get_text_embeddings_perp_neg(el.cuda(), az.cuda(), None, True, guidance_scale_neg=-3.0)
    want2, ww2 = pu_cpu.get_text_embeddings_perp_neg(el, az, None, True, guidance_scale_neg=-3.0)
    assert torch.equal(got2.cpu(), want2)
    torch.testing.assert_close(w2.cpu(), ww2, rtol=1e-6, atol=1e-7)
