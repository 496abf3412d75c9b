"""One full ASD training step on the CPU, composed only of oracle code (asd_oracle.c via ref_renderer.py +
the torch fp32 diffusion restatement).  TEST INFRASTRUCTURE: the CPU baseline of bench.py and the checker of
__graft_entry__.smoke(); never imported by scaledreamer_amd/.

Order of operations = StableDreamer.training_step (threestudio/systems/scaledreamer.py:48-103) for the
asd_sd_nerf configuration: render -> ASD guidance (stable_diffusion_asd_guidance.py:211-292) -> sparsity
regulariser -> backward to every field parameter.  The optimizer step is not included (it is <1 % of the step).
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import diffusion_ref as D
from . import ref_renderer as R


def asd_step(P: dict, unet_p, unet_layout, unet_cfg, vae_p, vae_plan, text_embeddings, neg_w, noise, t, t_plus,
             posterior_noise, guidance_scale: float = 7.5, lambda_sparsity: float = 30.0, timings: Optional[dict] = None):
    """P: renderer inputs (see ref_renderer.forward) with h, w.  text_embeddings [5B,77,1024] already assembled
    as [vd, uncond, neg1, neg2, vd]; neg_w [B,2] already multiplied by -guidance_perp_neg."""
    tm = timings if timings is not None else {}
    h, w = int(P["h"]), int(P["w"])
    B = P["rays_o"].reshape(-1, h * w, 3).shape[0]
    t0 = time.perf_counter()
    out, ctx = R.forward(P)
    tm["render_fwd"] = time.perf_counter() - t0
    comp = torch.tensor(out["comp_rgb"].reshape(B, h, w, 3), requires_grad=True)
    opacity = torch.tensor(out["opacity"].reshape(B, h, w, 1), requires_grad=True)
    t0 = time.perf_counter()
    rgb512 = F.interpolate(comp.permute(0, 3, 1, 2), (512, 512), mode="bilinear", align_corners=False)
    moments = D.vae_encode_moments(vae_p, vae_plan, rgb512 * 2.0 - 1.0)
    latents = D.sample_posterior(moments, posterior_noise)
    tm["vae_fwd"] = time.perf_counter() - t0
    alphas = D.alphas_cumprod()
    t0 = time.perf_counter()
    with torch.no_grad():
        x_t, x_tp = D.add_noise(alphas, latents, noise, t), D.add_noise(alphas, latents, noise, t_plus)
        n_rep = text_embeddings.shape[0] // B - 1
        eps = D.unet_forward(unet_p, unet_layout, unet_cfg, torch.cat([x_t] * n_rep + [x_tp]), torch.cat([t] * n_rep + [t_plus]),
                             text_embeddings)
        eps_p, eps_second = D.asd_eps_aggregate(eps, B, guidance_scale, neg_w)
        grad = torch.nan_to_num((1 - alphas[t]).view(-1, 1, 1, 1) * (eps_p - eps_second))
    tm["unet_fwd"] = time.perf_counter() - t0
    target = (latents - grad).detach()
    loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / B
    loss = loss + lambda_sparsity * (opacity ** 2 + 0.01).sqrt().mean()
    t0 = time.perf_counter()
    loss.backward()
    tm["vae_bwd"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    grads = R.backward(P, ctx, d_comp_rgb=comp.grad.numpy().reshape(-1, 3), d_opacity=opacity.grad.numpy().reshape(-1, 1))
    tm["render_bwd"] = time.perf_counter() - t0
    return float(loss.item()), grads, out, dict(latents=latents.detach(), eps=eps, grad=grad, d_comp_rgb=comp.grad)
