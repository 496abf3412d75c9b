"""ONE full-size training step of the headline preset (asd_sd_nerf: 64x64 rays x 512 samples, VAE at 512^2, the 865.9 M-parameter
SD-2.1 UNet at batch 5 with CFG + Perp-Neg + shifted timestep, AdamW untouched) with the guidance attached, against the oracle's
composition of the same step (oracle/ref_step.py: C renderer + torch fp32 diffusion restatement — test infrastructure) on injected
draws: the reference's StableDreamer.training_step (scaledreamer.py:48-126) around SDTimestepShiftedScoreDistillationGuidance.__call__
(stable_diffusion_asd_guidance.py:211-292, 377-428).  Every phase is pinned on its own elsewhere; this pins their product.

Tolerances: the renderer side is fp32 on both sides (1e-4); the image gradient passes through the fp16 prior (north_star: 1e-2 on the
networks' outputs) and through the classifier-free / Perp-Neg combination, which multiplies DIFFERENCES of UNet outputs by 7.5, so the
comparison is on direction and norm of the gradient image and of the hash-table gradient (cosine >= 0.99, norms within 5 %)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_size_c2_train_step_with_guidance_matches_the_oracle_step():
    import bench
    from oracle import ref_step
    from scaledreamer_amd.diffusion import weights as W

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg, system, data = bench.build_system("hip", seed=10)
    geo, bg, ren, guid = system.geometry, system.background, system.renderer, system.guidance
    system.on_train_batch_start()
    batch = bench.to_device(data.collate(), dev)
    n_rays = 64 * 64
    gen = torch.Generator(device="cpu").manual_seed(77)
    jit = torch.rand(n_rays, generator=gen).to(dev)
    ren.jitter_fn = lambda n, device: jit
    bg.rand_fn = lambda: 0.9                                    # learned background, no random colour this step
    post_noise, noise = torch.randn(1, 4, 64, 64, generator=gen).to(dev), torch.randn(1, 4, 64, 64, generator=gen).to(dev)
    t = torch.tensor([612], dtype=torch.long, device=dev)
    guid.posterior_noise_fn = lambda like: post_noise
    guid.noise_fn = lambda like: noise
    guid.timestep_fn = lambda lo, hi, n, device: t
    captured = {}
    fwd = system.forward

    def forward_and_keep(b):
        out = fwd(b)
        out["comp_rgb"].retain_grad()
        captured.update(out)
        return out

    system.forward = forward_and_keep
    system.optimizer.zero_grad(set_to_none=True)
    loss = system.training_step(batch)["loss"]
    loss.backward()
    t_plus = guid.get_t_plus(t)
    d_rgb = captured["comp_rgb"].grad.detach().float().cpu().numpy().reshape(n_rays, 3)

    # ---- the oracle's step on the same inputs ------------------------------------------------------------------------------------
    f = lambda x: x.detach().float().cpu().numpy()
    P = dict(h=64, w=64, spp=ren.cfg.num_samples_per_ray, radius=ren.cfg.radius, rays_o=f(batch["rays_o"]), rays_d=f(batch["rays_d"]),
             jitter=f(jit), occs=f(ren.estimator.occs), binaries=ren.estimator.binaries.cpu().numpy(), grid=f(geo.encoding.encoding.encoding.params),
             w1d=f(geo.density_network.layers[0].weight), w2d=f(geo.density_network.layers[2].weight), w1f=f(geo.feature_network.layers[0].weight),
             w2f=f(geo.feature_network.layers[2].weight), bgrid=f(bg.encoding.encoding.encoding.params), bw0=f(bg.network.layers[0].weight),
             bw1=f(bg.network.layers[2].weight), bw2=f(bg.network.layers[4].weight))
    context, neg_w = guid.conditioning(system.prompt_utils, batch["elevation"], batch["azimuth"], batch["camera_distances"])
    ucfg, vcfg = W.UNetConfig(), W.VAEConfig()
    layout = W.unet_layout(ucfg)
    vshapes, vplan = W.vae_encoder_layout(vcfg)
    up, vp = W.gen_params(layout[0], guid.cfg.weights_seed), W.gen_params(vshapes, guid.cfg.weights_seed + 1)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    want_loss, grads, out_ref, aux = ref_step.asd_step(P, up, layout, ucfg, vp, vplan, context.float().cpu(), neg_w.float().cpu(), noise.cpu(), t.cpu(),
                                                       t_plus.cpu(), post_noise.cpu(), guidance_scale=guid.cfg.guidance_scale,
                                                       lambda_sparsity=float(system.C(system.cfg.loss["lambda_sparsity"])))

    # renderer outputs of the step (fp32 on both sides)
    for k, c in (("comp_rgb", 3), ("opacity", 1)):
        got = captured[k].detach().cpu().numpy().reshape(n_rays, c)
        assert float(np.abs(got - out_ref[k]).max()) <= 3e-3 and float(np.abs(got - out_ref[k]).mean()) <= 1e-5, k

    def cos_and_ratio(a, b):
        a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)), float(np.linalg.norm(a) / (np.linalg.norm(b) + 1e-300))

    c, r = cos_and_ratio(d_rgb, aux["d_comp_rgb"].numpy().reshape(n_rays, 3))
    print(f"loss {float(loss.detach()):.4f} vs oracle {want_loss:.4f}; d loss / d comp_rgb: cosine {c:.5f}, norm ratio {r:.4f}")
    assert c >= 0.99 and abs(r - 1.0) <= 0.05, (c, r)
    assert abs(float(loss.detach()) - want_loss) <= 3e-2 * abs(want_loss), (float(loss.detach()), want_loss)
    for name, got in (("grid", geo.encoding.encoding.encoding.params.grad), ("w2f", geo.feature_network.layers[2].weight.grad),
                      ("w1d", geo.density_network.layers[0].weight.grad), ("bgrid", bg.encoding.encoding.encoding.params.grad)):
        c, r = cos_and_ratio(f(got), grads[name])
        print(f"d loss / d {name}: cosine {c:.5f}, norm ratio {r:.4f}")
        assert c >= 0.99 and abs(r - 1.0) <= 0.05, (name, c, r)
