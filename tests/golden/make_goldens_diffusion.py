"""Golden vectors for the diffusion half of the ASD step, produced by the REFERENCE's own classes.

  python tests/golden/make_goldens_diffusion.py     # writes tests/golden/diffusion_*.npz  (build container only)

Runs, imported in place from /root/reference under ref_harness.py:
  * extern.mvdream ... openaimodel.UNetModel / MultiViewUNetModel  (reduced width + the full SD-2.1 shape)
  * extern.mvdream ... model.Encoder (+ a 1x1 quant_conv as in autoencoder.py:32,81-85), forward and input gradient
  * extern.mvdream ... util.make_beta_schedule (the DDPM schedule of interface.py:48-76)
  * threestudio ... stable_diffusion_asd_guidance.SDTimestepShiftedScoreDistillationGuidance.__call__ / get_eps /
    get_t_plus and prompt_processors.base.PromptProcessorOutput.get_text_embeddings_perp_neg, driven with a
    cheap stand-in UNet / VAE so that only the reference's glue arithmetic is exercised.
Weights follow scaledreamer_amd.diffusion.weights.gen_params (seeded, name-keyed): loading them with
load_state_dict(strict=True) also proves that the parameter names / shapes of that layout are the reference's.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install()

from scaledreamer_amd.diffusion import weights as W  # noqa: E402

from extern.mvdream.ldm.modules.diffusionmodules.model import Encoder  # noqa: E402
from extern.mvdream.ldm.modules.diffusionmodules.openaimodel import MultiViewUNetModel, UNetModel  # noqa: E402
from extern.mvdream.ldm.modules.diffusionmodules.util import make_beta_schedule  # noqa: E402


def rnd(name, shape, seed=0):
    """seeded input tensors share the name-keyed rule of the weights (plain N(0,1))."""
    import zlib
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g)


def unet_kwargs(cfg: W.UNetConfig):
    kw = dict(image_size=32, in_channels=cfg.in_channels, out_channels=cfg.out_channels, model_channels=cfg.model_channels,
              attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks,
              channel_mult=list(cfg.channel_mult), num_head_channels=cfg.num_head_channels, use_spatial_transformer=True,
              use_linear_in_transformer=True, transformer_depth=cfg.transformer_depth, context_dim=cfg.context_dim,
              use_checkpoint=False, legacy=False)
    if cfg.camera_dim is not None:
        kw["camera_dim"] = cfg.camera_dim
    return kw


def make_unet_golden(name, cfg: W.UNetConfig, batch, hw, n_ctx, seed, num_frames=1):
    shapes, *_ = W.unet_layout(cfg)
    cls = MultiViewUNetModel if cfg.camera_dim is not None else UNetModel
    model = cls(**unet_kwargs(cfg)).eval()
    assert sum(p.numel() for p in model.parameters()) == W.count_params(shapes)
    model.load_state_dict(W.gen_params(shapes, seed), strict=True)
    x = rnd("in.x", (batch, cfg.in_channels, hw, hw), seed)
    t = torch.tensor([(37 * i + 501) % 1000 for i in range(batch)], dtype=torch.long)
    if cfg.camera_dim is not None:  # MVDream shares t across the frames of a group
        t = t.view(-1, num_frames)[:, :1].expand(-1, num_frames).reshape(-1)
    ctx = rnd("in.context", (batch, n_ctx, cfg.context_dim), seed)
    kw = {}
    if cfg.camera_dim is not None:
        kw = dict(camera=rnd("in.camera", (batch, cfg.camera_dim), seed), num_frames=num_frames)
    with torch.no_grad():
        eps = model(x, t, context=ctx, **kw)
    save = dict(seed=seed, batch=batch, hw=hw, n_ctx=n_ctx, num_frames=num_frames, t=t.numpy(), eps=eps.numpy(),
                cfg=np.array([cfg.model_channels, cfg.num_head_channels, cfg.context_dim, cfg.camera_dim or 0]),
                channel_mult=np.array(cfg.channel_mult), attention_resolutions=np.array(cfg.attention_resolutions))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **save)
    print(f"{name}: params={W.count_params(shapes)} eps mean|.|={eps.abs().mean():.4f} std={eps.std():.4f}")


def make_unet_shared_golden(name, cfg: W.UNetConfig, reps, group, hw, n_ctx, seed, num_frames=1):
    """The batch layout the ASD step hands to the UNet (stable_diffusion_asd_guidance.py:377-394, mvdream_asd_guidance.py:231-246):
    `reps` repetitions of `group` inputs (x, t[, camera]) under DIFFERENT text contexts, then `group` more inputs at the shifted
    timestep t+ — the form asd_unet_fwd_shared computes with the prefix in front of the first cross-attention shared."""
    shapes, *_ = W.unet_layout(cfg)
    cls = MultiViewUNetModel if cfg.camera_dim is not None else UNetModel
    model = cls(**unet_kwargs(cfg)).eval()
    model.load_state_dict(W.gen_params(shapes, seed), strict=True)
    N = (reps + 1) * group
    xa, xb = rnd("in.xa", (group, cfg.in_channels, hw, hw), seed), rnd("in.xb", (group, cfg.in_channels, hw, hw), seed)
    x = torch.cat([xa] * reps + [xb])
    t0, t1 = 611, 640
    t = torch.tensor([t0] * (reps * group) + [t1] * group, dtype=torch.long)
    ctx = rnd("in.context", (N, n_ctx, cfg.context_dim), seed)
    kw = {}
    if cfg.camera_dim is not None:
        kw = dict(camera=torch.cat([rnd("in.camera", (group, cfg.camera_dim), seed)] * (reps + 1)), num_frames=num_frames)
    with torch.no_grad():
        eps = model(x, t, context=ctx, **kw)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, reps=reps, group=group, hw=hw, n_ctx=n_ctx, num_frames=num_frames,
                        t=t.numpy(), eps=eps.numpy())
    print(f"{name}: batch={N} eps mean|.|={eps.abs().mean():.4f} std={eps.std():.4f}")


def make_vae_golden(name, cfg: W.VAEConfig, batch, res, seed, grad_stride):
    shapes, plan = W.vae_encoder_layout(cfg)
    enc = Encoder(ch=cfg.ch, out_ch=3, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks, attn_resolutions=[],
                  dropout=0.0, in_channels=cfg.in_channels, resolution=res, z_channels=cfg.z_channels, double_z=True).eval()
    quant = torch.nn.Conv2d(2 * cfg.z_channels, 2 * cfg.embed_dim, 1)
    p = W.gen_params(shapes, seed)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}, strict=True)
    quant.load_state_dict({"weight": p["quant_conv.weight"], "bias": p["quant_conv.bias"]})
    for q in list(enc.parameters()) + list(quant.parameters()):
        q.requires_grad_(False)  # frozen, as stable_diffusion_asd_guidance.py:101-102
    x = (torch.tanh(rnd("in.img", (batch, 3, res, res), seed))).requires_grad_(True)
    moments = quant(enc(x))
    gm = rnd("in.gmoments", tuple(moments.shape), seed)
    (moments * gm).sum().backward()
    gx = x.grad[:, :, ::grad_stride, ::grad_stride].contiguous()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, batch=batch, res=res, grad_stride=grad_stride,
                        moments=moments.detach().numpy(), grad_x_sub=gx.numpy(),
                        cfg=np.array([cfg.ch, cfg.num_res_blocks, cfg.z_channels]), ch_mult=np.array(cfg.ch_mult))
    print(f"{name}: params={W.count_params(shapes)} moments std={moments.std():.4f} |gx|={x.grad.abs().mean():.5f}")


def make_schedule_golden():
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120, cosine_s=8e-3)
    ac = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64), axis=0)
    np.savez_compressed(os.path.join(HERE, "diffusion_schedule.npz"), alphas_cumprod=ac)
    print("schedule: alphas_cumprod[0,500,999] =", ac[0], ac[500], ac[999])


def make_asd_glue_golden(seed=5):
    import types
    dummy = type("Dummy", (), {})
    H._mod("diffusers", DDPMScheduler=dummy, DPMSolverMultistepScheduler=dummy, StableDiffusionPipeline=dummy,
           UNet2DConditionModel=dummy)
    H._mod("diffusers.utils")
    H._mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    from threestudio.models.guidance.stable_diffusion_asd_guidance import SDTimestepShiftedScoreDistillationGuidance as G
    from threestudio.models.prompt_processors.base import DirectionConfig, PromptProcessorOutput, shift_azimuth_deg

    B = 4
    guid = object.__new__(G)
    guid.cfg = G.Config(guidance_scale=7.5, plus_ratio=0.1, plus_random=True, guidance_perp_neg=-0.5,
                        min_step_percent=0.5, max_step_percent=0.98)
    guid.device = torch.device("cpu")
    guid.weights_dtype = torch.float32
    guid.num_train_timesteps = 1000
    guid.set_min_max_steps(0.5, 0.98)
    ac = torch.from_numpy(np.load(os.path.join(HERE, "diffusion_schedule.npz"))["alphas_cumprod"]).float()
    guid.alphas = ac
    guid.grad_clip_val = None
    guid.use_perp_neg = True

    class Sched:  # DDPMScheduler.add_noise (diffusers, un-vendored): x_t = sqrt(abar) x + sqrt(1-abar) eps
        def add_noise(self, x, noise, t):
            a = ac[t].view(-1, 1, 1, 1)
            return a.sqrt() * x + (1 - a).sqrt() * noise
    guid.scheduler = Sched()

    calls = {}

    def fake_unet(latents, t, encoder_hidden_states=None):
        calls.update(latents=latents.clone(), t=t.clone(), ctx=encoder_hidden_states.clone())
        s = encoder_hidden_states.mean(dim=(1, 2)).view(-1, 1, 1, 1)
        out = torch.tanh(0.7 * latents + 3.0 * s) * (1.0 + t.view(-1, 1, 1, 1) / 1000.0) + 0.1 * latents.flip(-1)
        return types.SimpleNamespace(sample=out)
    guid.unet = fake_unet

    class FakeVAE:
        config = types.SimpleNamespace(scaling_factor=0.18215)

        def encode(self, imgs):
            pooled = torch.nn.functional.avg_pool2d(imgs, 8)                   # [B,3,64,64]
            z = torch.cat([pooled, pooled.mean(1, keepdim=True) ** 2], dim=1)   # [B,4,64,64]
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: z))
    guid.vae = FakeVAE()

    # prompt side: the reference's own PromptProcessorOutput with synthetic embeddings ("synthetic random prompts")
    emb_vd = rnd("prompt.vd", (4, 77, 1024), seed)
    unc_vd = rnd("prompt.uncond", (1, 77, 1024), seed).expand(4, -1, -1).contiguous()
    front_thr = back_thr = 30.0
    directions = [
        DirectionConfig("side", lambda s: s, lambda s: s, lambda ele, azi, dis: torch.ones_like(ele, dtype=torch.bool)),
        DirectionConfig("front", lambda s: s, lambda s: s,
                        lambda ele, azi, dis: (shift_azimuth_deg(azi) > -front_thr) & (shift_azimuth_deg(azi) < front_thr)),
        DirectionConfig("back", lambda s: s, lambda s: s,
                        lambda ele, azi, dis: (shift_azimuth_deg(azi) > 180 - back_thr) | (shift_azimuth_deg(azi) < -180 + back_thr)),
        DirectionConfig("overhead", lambda s: s, lambda s: s, lambda ele, azi, dis: ele > 60.0),
    ]
    pu = PromptProcessorOutput(text_embeddings=emb_vd[0], uncond_text_embeddings=unc_vd[0], text_embeddings_vd=emb_vd,
                               uncond_text_embeddings_vd=unc_vd, directions=directions,
                               direction2idx={d.name: i for i, d in enumerate(directions)}, use_perp_neg=True,
                               perp_neg_f_sb=(1, 0.5, -0.606), perp_neg_f_fsb=(1, 0.5, +0.967), perp_neg_f_fs=(4, 0.5, -2.426),
                               perp_neg_f_sf=(4, 0.5, -2.426), prompt="", prompts_vd=[""] * 4)
    elevation = torch.tensor([10.0, 25.0, 70.0, -5.0])
    azimuth = torch.tensor([12.0, -100.0, 40.0, 170.0])
    distances = torch.tensor([1.2, 1.1, 1.4, 1.0])
    rgb = torch.sigmoid(rnd("asd.rgb", (B, 64, 64, 3), seed)).requires_grad_(True)

    rec = {}
    real = dict(randn_like=torch.randn_like, randint=torch.randint, rand=torch.rand)

    def wrap(name):
        def f(*a, **k):
            out = real[name](*a, **k)
            rec[name] = out.clone()
            return out
        return f
    torch.manual_seed(seed)
    torch.randn_like, torch.randint, torch.rand = wrap("randn_like"), wrap("randint"), wrap("rand")
    try:
        out = guid(rgb, pu, elevation, azimuth, distances)
    finally:
        torch.randn_like, torch.randint, torch.rand = real["randn_like"], real["randint"], real["rand"]
    out["loss_asd"].backward()
    temb, w = pu.get_text_embeddings_perp_neg(elevation, azimuth, distances, True)
    t_plus = guid.get_t_plus(rec["randint"])  # consumes fresh randomness; recomputed below with the recorded rand
    np.savez_compressed(
        os.path.join(HERE, "diffusion_asd_glue.npz"), seed=seed, elevation=elevation.numpy(), azimuth=azimuth.numpy(),
        camera_distances=distances.numpy(), noise=rec["randn_like"].numpy(), t=rec["randint"].numpy(), rand=rec["rand"].numpy(),
        min_step=guid.min_step, max_step=guid.max_step, loss_asd=np.float64(out["loss_asd"].item()),
        grad_norm=np.float64(out["grad_norm"].item()), grad_rgb=rgb.grad.numpy(), unet_in_t=calls["t"].numpy(),
        unet_in_latents=calls["latents"].numpy(), unet_in_ctx_mean=calls["ctx"].mean(dim=2).numpy(),
        perp_neg_weights=w.numpy(), text_embeddings_mean=temb.mean(dim=2).numpy(),
    )
    print(f"asd glue: loss={out['loss_asd'].item():.5f} grad_norm={out['grad_norm'].item():.5f} t={rec['randint'].tolist()} "
          f"unet t={calls['t'].tolist()}")


def make_mvdream_glue_golden(seed=8):
    """MVDreamTimestepShiftedScoreDistillationGuidance.__call__ (mvdream_asd_guidance.py:167-304) with a stand-in
    LatentDiffusionInterface (q_sample / apply_model / encode_first_stage / get_first_stage_encoding)."""
    import types
    H._mod("extern.mvdream.model_zoo", build_model=None)
    from threestudio.models.guidance.mvdream_asd_guidance import MVDreamTimestepShiftedScoreDistillationGuidance as G
    from threestudio.models.prompt_processors.base import DirectionConfig, PromptProcessorOutput

    B = 4
    guid = object.__new__(G)
    guid.cfg = G.Config(guidance_scale=7.5, plus_ratio=0.1, plus_random=True, n_view=4)
    guid.device = torch.device("cpu")
    guid.num_train_timesteps = 1000
    guid.min_step, guid.max_step = 20, 980
    guid.grad_clip_val = None
    ac = torch.from_numpy(np.load(os.path.join(HERE, "diffusion_schedule.npz"))["alphas_cumprod"]).float()
    guid.alphas = ac
    calls = {}

    class Model:
        alphas_cumprod = ac

        def q_sample(self, x, t, noise=None):
            a = ac[t].view(-1, 1, 1, 1)
            return a.sqrt() * x + (1 - a).sqrt() * noise

        def apply_model(self, x, t, cond):
            calls.update(x=x.clone(), t=t.clone(), ctx=cond["context"].clone(), camera=cond["camera"].clone(), nf=cond["num_frames"])
            s = cond["context"].mean(dim=(1, 2)).view(-1, 1, 1, 1) + cond["camera"].mean(dim=1).view(-1, 1, 1, 1)
            return torch.tanh(0.7 * x + 3.0 * s) * (1.0 + t.view(-1, 1, 1, 1) / 1000.0) + 0.1 * x.flip(-1)

        def encode_first_stage(self, imgs):
            pooled = torch.nn.functional.avg_pool2d(imgs, 8)
            return torch.cat([pooled, pooled.mean(1, keepdim=True) ** 2], dim=1)

        def get_first_stage_encoding(self, z):
            return 0.18215 * z
    guid.model = Model()
    emb = rnd("mv.prompt", (1, 77, 1024), seed)
    unc = rnd("mv.uncond", (1, 77, 1024), seed)
    side = DirectionConfig("side", lambda s: s, lambda s: s, lambda ele, azi, dis: torch.ones_like(ele, dtype=torch.bool))
    pu = PromptProcessorOutput(text_embeddings=emb, uncond_text_embeddings=unc, text_embeddings_vd=emb.expand(4, -1, -1),
                               uncond_text_embeddings_vd=unc.expand(4, -1, -1), directions=[side], direction2idx={"side": 0},
                               use_perp_neg=False, perp_neg_f_sb=(1, 0.5, -0.606), perp_neg_f_fsb=(1, 0.5, 0.967),
                               perp_neg_f_fs=(4, 0.5, -2.426), perp_neg_f_sf=(4, 0.5, -2.426), prompt="", prompts_vd=[""] * 4)
    elevation = torch.tensor([10.0, 10.0, 10.0, 10.0])
    azimuth = torch.tensor([12.0, 102.0, -168.0, -78.0])
    distances = torch.full((4,), 1.2)
    c2w = torch.eye(4).repeat(4, 1, 1)
    c2w[:, :3, :3] = torch.linalg.qr(rnd("mv.rot", (4, 3, 3), seed))[0]
    c2w[:, :3, 3] = rnd("mv.pos", (4, 3), seed) * 1.3
    rgb = torch.sigmoid(rnd("mv.rgb", (B, 64, 64, 3), seed)).requires_grad_(True)
    rec = {}
    real = dict(randn_like=torch.randn_like, randint=torch.randint, rand=torch.rand)

    def wrap(name):
        def f(*a, **k):
            out = real[name](*a, **k)
            rec[name] = out.clone()
            return out
        return f
    torch.manual_seed(seed)
    torch.randn_like, torch.randint, torch.rand = wrap("randn_like"), wrap("randint"), wrap("rand")
    try:
        out = guid(rgb, pu, elevation, azimuth, distances, c2w.clone())
    finally:
        torch.randn_like, torch.randint, torch.rand = real["randn_like"], real["randint"], real["rand"]
    out["loss_asd"].backward()
    np.savez_compressed(os.path.join(HERE, "diffusion_mvdream_glue.npz"), seed=seed, elevation=elevation.numpy(), azimuth=azimuth.numpy(),
                        camera_distances=distances.numpy(), c2w=c2w.numpy(), noise=rec["randn_like"].numpy(), t=rec["randint"].numpy(),
                        rand=rec["rand"].numpy(), loss_asd=np.float64(out["loss_asd"].item()), grad_norm=np.float64(out["grad_norm"].item()),
                        grad_rgb=rgb.grad.numpy(), unet_in_t=calls["t"].numpy(), unet_in_x=calls["x"].numpy(),
                        unet_in_camera=calls["camera"].numpy(), unet_in_ctx_mean=calls["ctx"].mean(dim=2).numpy(), num_frames=calls["nf"])
    print(f"mvdream glue: loss={out['loss_asd'].item():.5f} t={calls['t'].tolist()}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_schedule_golden()
    make_asd_glue_golden()
    make_mvdream_glue_golden()
    if "--glue-only" in sys.argv:
        sys.exit(0)
    if "--round2" in sys.argv:   # the headline config's own shapes: only these two files are (re)written
        make_vae_golden("diffusion_vae_full_512", W.VAEConfig(), batch=1, res=512, seed=2, grad_stride=8)
        make_unet_golden("diffusion_mvunet_full_b12", W.UNetConfig(camera_dim=16), batch=12, hw=32, n_ctx=77, seed=4, num_frames=4)
        sys.exit(0)
    if "--round3" in sys.argv:   # the call forms the bench times: shared-prefix batches, 4 x 256^2 VAE
        make_unet_shared_golden("diffusion_unet_sd21_shared_b5", W.UNetConfig(), reps=4, group=1, hw=64, n_ctx=77, seed=1)
        make_unet_shared_golden("diffusion_mvunet_shared_b12", W.UNetConfig(camera_dim=16), reps=2, group=4, hw=32, n_ctx=77, seed=4, num_frames=4)
        make_vae_golden("diffusion_vae_full_256_b4", W.VAEConfig(), batch=4, res=256, seed=2, grad_stride=4)
        sys.exit(0)
    small = W.UNetConfig(model_channels=64, num_head_channels=32, context_dim=96)
    make_unet_golden("diffusion_unet_small", small, batch=3, hw=16, n_ctx=7, seed=3)
    small_mv = W.UNetConfig(model_channels=64, num_head_channels=32, context_dim=96, camera_dim=16)
    make_unet_golden("diffusion_mvunet_small", small_mv, batch=8, hw=8, n_ctx=7, seed=4, num_frames=4)
    make_vae_golden("diffusion_vae_small", W.VAEConfig(ch=32), batch=2, res=64, seed=6, grad_stride=1)
    if "--full" in sys.argv or True:
        make_unet_golden("diffusion_unet_sd21_full", W.UNetConfig(), batch=1, hw=64, n_ctx=77, seed=1)
        make_vae_golden("diffusion_vae_full_256", W.VAEConfig(), batch=1, res=256, seed=2, grad_stride=4)
