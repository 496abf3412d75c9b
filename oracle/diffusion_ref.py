"""CPU fp32 restatement (plain torch functional ops, NCHW) of the frozen diffusion prior of the ASD step.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by scaledreamer_amd/.  (For a floating-point path the tier rules allow a torch fp32 reference.)

Follows the reference's vendored LDM code, which is architecturally identical to the diffusers models the
SD path loads (same parameter counts):
  UNetModel / MultiViewUNetModel  extern/mvdream/ldm/modules/diffusionmodules/openaimodel.py:422-808, 811-1213
  ResBlock :163-275, Downsample/Upsample :91-160, timestep_embedding diffusionmodules/util.py:165-186
  SpatialTransformer(3D) / BasicTransformerBlock(3D) / CrossAttention / GEGLU  modules/attention.py:49-76,145-194,246-412
  VAE Encoder / ResnetBlock / AttnBlock / Downsample  modules/diffusionmodules/model.py:40-203,452-543
  quant_conv + DiagonalGaussianDistribution.sample  models/autoencoder.py:81-85, modules/distributions/distributions.py:24-37
Pinned by tests/golden/diffusion_*.npz, which were produced by running those reference classes themselves
(tests/golden/make_goldens_diffusion.py).  Parameters come in as a dict keyed by the LDM state-dict names.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]


def _gn(p: P, name: str, x, eps: float):
    return F.group_norm(x.float(), 32, p[name + ".weight"], p[name + ".bias"], eps).type(x.dtype)


def _conv(p: P, name: str, x, stride=1, padding=1):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def _lin(p: P, name: str, x):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def timestep_embedding(t, dim: int, max_period: int = 10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _resblock(p: P, name: str, x, emb):
    h = _conv(p, name + ".in_layers.2", F.silu(_gn(p, name + ".in_layers.0", x, 1e-5)))
    h = h + _lin(p, name + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = _conv(p, name + ".out_layers.3", F.silu(_gn(p, name + ".out_layers.0", h, 1e-5)))
    if name + ".skip_connection.weight" in p:
        x = _conv(p, name + ".skip_connection", x, padding=0)
    return x + h


def _attention(p: P, name: str, x, context, heads: int):
    q, k, v = _lin(p, name + ".to_q", x), _lin(p, name + ".to_k", context), _lin(p, name + ".to_v", context)
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.view(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (d ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    return _lin(p, name + ".to_out.0", out.permute(0, 2, 1, 3).reshape(b, n, c))


def _transformer(p: P, name: str, x, context, head_dim: int, depth: int, num_frames: int = 1, three_d: bool = False):
    b, c, h, w = x.shape
    x_in = x
    x = _gn(p, name + ".norm", x, 1e-6)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = _lin(p, name + ".proj_in", x)
    heads = c // head_dim
    for d in range(depth):
        blk = f"{name}.transformer_blocks.{d}"
        y = F.layer_norm(x, (c,), p[blk + ".norm1.weight"], p[blk + ".norm1.bias"])
        if three_d:  # BasicTransformerBlock3D: self-attention over all frames' tokens (attention.py:348-354)
            y = y.reshape(b // num_frames, num_frames * h * w, c)
            y = _attention(p, blk + ".attn1", y, y, heads).reshape(b, h * w, c)
        else:
            y = _attention(p, blk + ".attn1", y, y, heads)
        x = y + x
        x = _attention(p, blk + ".attn2", F.layer_norm(x, (c,), p[blk + ".norm2.weight"], p[blk + ".norm2.bias"]), context, heads) + x
        y = F.layer_norm(x, (c,), p[blk + ".norm3.weight"], p[blk + ".norm3.bias"])
        a, gate = _lin(p, blk + ".ff.net.0.proj", y).chunk(2, dim=-1)
        x = _lin(p, blk + ".ff.net.2", a * F.gelu(gate)) + x
    x = _lin(p, name + ".proj_out", x)
    return x.reshape(b, h, w, c).permute(0, 3, 1, 2) + x_in


def _apply(p: P, layers, h, emb, context, cfg, num_frames, three_d):
    for kind, name, cin, cout in layers:
        if kind == "conv":
            h = _conv(p, name, h)
        elif kind == "res":
            h = _resblock(p, name, h, emb)
        elif kind == "attn":
            h = _transformer(p, name, h, context, cfg.num_head_channels, cfg.transformer_depth, num_frames, three_d)
        elif kind == "down":
            h = _conv(p, name, h, stride=2)
        elif kind == "up":
            h = _conv(p, name, F.interpolate(h, scale_factor=2, mode="nearest"))
        else:
            raise ValueError(kind)
    return h


def unet_forward(p: P, layout, cfg, x, timesteps, context, camera: Optional[torch.Tensor] = None, num_frames: int = 1):
    """eps = UNet(x [N,4,H,W], t [N], context [N,77,1024]); layout = weights.unet_layout(cfg)."""
    _, inputs, middle, outputs = layout
    three_d = cfg.camera_dim is not None
    emb = _lin(p, "time_embed.2", F.silu(_lin(p, "time_embed.0", timestep_embedding(timesteps, cfg.model_channels))))
    if camera is not None:
        emb = emb + _lin(p, "camera_embed.2", F.silu(_lin(p, "camera_embed.0", camera)))
    hs, h = [], x
    for blk in inputs:
        h = _apply(p, blk.layers, h, emb, context, cfg, num_frames, three_d)
        hs.append(h)
    h = _apply(p, middle.layers, h, emb, context, cfg, num_frames, three_d)
    for blk in outputs:
        h = _apply(p, blk.layers, torch.cat([h, hs.pop()], dim=1), emb, context, cfg, num_frames, three_d)
    return _conv(p, "out.2", F.silu(_gn(p, "out.0", h, 1e-5)))


# ---- VAE encoder ---------------------------------------------------------------------------------
def _vae_res(p: P, name: str, x):
    h = _conv(p, name + ".conv1", F.silu(_gn(p, name + ".norm1", x, 1e-6)))
    h = _conv(p, name + ".conv2", F.silu(_gn(p, name + ".norm2", h, 1e-6)))
    if name + ".nin_shortcut.weight" in p:
        x = _conv(p, name + ".nin_shortcut", x, padding=0)
    return x + h


def _vae_attn(p: P, name: str, x):
    h = _gn(p, name + ".norm", x, 1e-6)
    q, k, v = (_conv(p, f"{name}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(p, name + ".proj_out", h, padding=0)


def vae_encode_moments(p: P, plan, x):
    """moments [B, 2*embed_dim, H/8, W/8] = quant_conv(Encoder(x)); plan = weights.vae_encoder_layout(cfg)[1]."""
    h = x
    for kind, name, cin, cout in plan:
        if kind == "conv":
            h = _conv(p, name, h)
        elif kind == "res":
            h = _vae_res(p, name, h)
        elif kind == "down":  # asymmetric (0,1,0,1) zero pad + stride-2 conv (model.py:80-85)
            h = _conv(p, name, F.pad(h, (0, 1, 0, 1), mode="constant", value=0), stride=2, padding=0)
        elif kind == "attn":
            h = _vae_attn(p, name, h)
        elif kind == "out":
            h = _conv(p, name + ".conv_out", F.silu(_gn(p, name + ".norm_out", h, 1e-6)))
        elif kind == "quant":
            h = _conv(p, name, h, padding=0)
        else:
            raise ValueError(kind)
    return h


def sample_posterior(moments, noise, scale_factor: float = 0.18215):
    """DiagonalGaussianDistribution.sample() * scale_factor with the noise injected."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return (mean + torch.exp(0.5 * logvar) * noise) * scale_factor


# ---- scheduler + ASD glue (stable_diffusion_asd_guidance.py:211-428) --------
def alphas_cumprod(n: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012):
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2
    return torch.cumprod(1.0 - betas, dim=0).float()


def add_noise(alphas, x, noise, t):
    a = alphas[t].view(-1, 1, 1, 1)
    return a.sqrt() * x + (1 - a).sqrt() * noise


def get_t_plus(t, min_step: int, plus_ratio: float, rand: Optional[torch.Tensor], num_train_timesteps: int = 1000):
    """stable_diffusion_asd_guidance.py:294-316 (the later of the two definitions is the live one)."""
    t_plus = plus_ratio * (t - min_step)
    t_plus = t_plus.clamp(torch.zeros_like(t), num_train_timesteps - t - 1)
    if rand is not None:
        t_plus = t_plus * rand
    t_plus = t + t_plus.to(torch.long)
    return torch.clamp(t_plus, 1, max=num_train_timesteps - 1)


def perpendicular_component(x, y):
    eps = torch.ones_like(x[:, 0, 0, 0]) * 1e-6
    return x - (torch.mul(x, y).sum(dim=[1, 2, 3]) / torch.maximum(torch.mul(y, y).sum(dim=[1, 2, 3]), eps)).view(-1, 1, 1, 1) * y


def asd_eps_aggregate(noise_pred, batch_size: int, guidance_scale: float, neg_guidance_weights: Optional[torch.Tensor]):
    """CFG (+ Perp-Neg) aggregation of one batched UNet call (stable_diffusion_asd_guidance.py:405-428)."""
    B = batch_size
    text, uncond = noise_pred[0:B], noise_pred[B:2 * B]
    eps_pos = text - uncond
    if neg_guidance_weights is not None:
        neg, second = noise_pred[2 * B:4 * B], noise_pred[4 * B:5 * B]
        accum = 0
        n_neg = neg_guidance_weights.shape[-1]
        for i in range(n_neg):
            eps_neg = neg[i::n_neg] - uncond
            accum = accum + neg_guidance_weights[:, i].view(-1, 1, 1, 1) * perpendicular_component(eps_neg, eps_pos)
        return (eps_pos + accum) * guidance_scale + uncond, second
    return eps_pos * guidance_scale + uncond, noise_pred[2 * B:3 * B]


def asd_guidance_loss(rgb, encode_fn, unet_fn, context, neg_w, t, t_plus, noise, post_noise, guidance_scale: float, image_size: int,
                      weighting: str = "sds", scale_factor: float = 0.18215, grad_clip: Optional[float] = None, unet_kw=None):
    """One ASD guidance evaluation, composed from the primitives above in the order of
    SDTimestepShiftedScoreDistillationGuidance.__call__ (stable_diffusion_asd_guidance.py:211-292) / the MVDream variant
    (mvdream_asd_guidance.py:167-304): resize -> encode -> posterior sample -> two noisings -> one batched UNet call ->
    CFG / Perp-Neg -> w(t) -> nan_to_num -> MSE re-parameterisation.  `context` is already in the UNet's batch order
    (text | uncond | [negatives] | text); t, t_plus are per sample.  Returns (loss, grad_norm, unet inputs)."""
    import torch.nn.functional as F

    B = rgb.shape[0]
    imgs = F.interpolate(rgb.permute(0, 3, 1, 2), (image_size, image_size), mode="bilinear", align_corners=False) * 2.0 - 1.0
    latents = sample_posterior(encode_fn(imgs), post_noise, scale_factor)
    alphas = alphas_cumprod()
    n_rep = context.shape[0] // B - 1
    with torch.no_grad():
        x_in = torch.cat([add_noise(alphas, latents, noise, t)] * n_rep + [add_noise(alphas, latents, noise, t_plus)], dim=0)
        t_in = torch.cat([t] * n_rep + [t_plus], dim=0)
        eps = unet_fn(x_in, t_in, context, **(unet_kw or {}))
        first, second = asd_eps_aggregate(eps, B, guidance_scale, neg_w)
        a = alphas[t].view(-1, 1, 1, 1)
        w = {"sds": 1 - a, "uniform": torch.ones_like(a), "fantasia3d": a.sqrt() * (1 - a)}[weighting]
        grad = torch.nan_to_num(w * (first - second))
        if grad_clip is not None:
            grad = grad.clamp(-grad_clip, grad_clip)
    target = (latents - grad).detach()
    loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B
    return loss, grad.norm(), dict(x=x_in, t=t_in)
