#!/usr/bin/env python
"""ASD training-step benchmark (BASELINE.json metric: ASD train steps/sec, 64x64 render, SD-2.1 guidance).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = StableDreamer.train_one_step on the asd_sd_nerf configuration (configs[1] of BASELINE.json):
random camera -> HIP render (march, prune, field, composite) -> VAE encode 512^2 -> SD-2.1 UNet x5 (CFG +
Perp-Neg + shifted timestep) -> ASD loss -> backward through VAE and renderer into the hash grid / MLPs ->
(N > 1: RCCL mean all-reduce of the gradients) -> AdamW step.  Synthetic prompts (N(0,1) embeddings) and
seeded random-init diffusion weights (no checkpoints exist offline); nothing inside the step is skipped.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F16_PEAK_TF = 2500.0  # dense bf16/f16 MFMA


def build_system(backend: str, seed: int):
    from scaledreamer_amd import presets
    from scaledreamer_amd.data import RandomCameraIterableDataset
    from scaledreamer_amd.guidance import PromptUtils
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    cfg = presets.asd_sd_nerf(guidance_backend=backend)
    torch.manual_seed(seed)
    random.seed(seed)
    pp = cfg["system"]["prompt_processor"]
    dev = torch.device("cuda", torch.cuda.current_device())
    prompt_utils = PromptUtils.synthetic(seed=1234, device=dev, front_threshold=pp["front_threshold"],
                                         back_threshold=pp["back_threshold"])
    system = find(cfg["system_type"])(cfg["system"], prompt_utils=prompt_utils)
    system.train()
    data = RandomCameraIterableDataset(cfg["data"])
    return cfg, system, data


def to_device(batch, dev):
    return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def roofline_field_kernel(system, batch, reps: int = 20):
    """Average duration (HIP events on the launch stream) of the dominant hand-written renderer kernel,
    field_fwd_kernel, on the live samples of one step; algorithmic bytes per DESIGN.md / SURVEY.md §8d."""
    from scaledreamer_amd import ops

    ren, geo = system.renderer, system.geometry
    with torch.no_grad():
        ri, t0, t1, pts, dirs, off, cnt = ren._sample(batch["rays_o"].reshape(-1, 3).contiguous(), batch["rays_d"].reshape(-1, 3).contiguous())
    n = int(pts.shape[0])
    if n == 0:
        return None
    grid = geo.encoding.encoding.encoding.params.detach()
    w = [t.detach() for t in geo._weights()]
    for _ in range(3):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # per kept sample: 4 encodes x 1024 B gathered + 12 B position in + 28 B (sigma, features, normal) out
    # + 128 B centre encoding saved for the backward pass
    bytes_per_sample = 4 * 1024 + 12 + 28 + 128
    achieved = n * bytes_per_sample / (ms * 1e-3) / 1e9
    return {"kernel": "field_fwd_kernel<16,64,3>", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "samples_per_launch": n,
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_sample": bytes_per_sample}


def cpu_baseline(system, batch, seed: int):
    """The oracle (a scalar/OpenMP C port + torch fp32) timed on this host's cores on ONE full step of the
    same workload (same camera, same parameters, same synthetic prompt embeddings)."""
    import numpy as np
    from oracle import oracle as O  # noqa: F401
    from oracle import ref_step
    from scaledreamer_amd.diffusion import weights as W

    torch.set_num_threads(os.cpu_count() or 1)
    geo, bg, ren, guid = system.geometry, system.background, system.renderer, system.guidance
    f = lambda t: t.detach().float().cpu().numpy()
    P = dict(h=64, w=64, spp=ren.cfg.num_samples_per_ray, radius=ren.cfg.radius, rays_o=f(batch["rays_o"]), rays_d=f(batch["rays_d"]),
             jitter=np.random.default_rng(seed).uniform(0, 1, 4096).astype(np.float32), occs=f(ren.estimator.occs),
             binaries=ren.estimator.binaries.cpu().numpy(), grid=f(geo.encoding.encoding.encoding.params),
             w1d=f(geo.density_network.layers[0].weight), w2d=f(geo.density_network.layers[2].weight),
             w1f=f(geo.feature_network.layers[0].weight), w2f=f(geo.feature_network.layers[2].weight),
             bgrid=f(bg.encoding.encoding.encoding.params), bw0=f(bg.network.layers[0].weight),
             bw1=f(bg.network.layers[2].weight), bw2=f(bg.network.layers[4].weight))
    ucfg, vcfg = W.UNetConfig(), W.VAEConfig()
    layout = W.unet_layout(ucfg)
    vshapes, vplan = W.vae_encoder_layout(vcfg)
    wseed = guid.cfg.weights_seed
    up, vp = W.gen_params(layout[0], wseed), W.gen_params(vshapes, wseed + 1)
    pu = system.prompt_utils
    el, az, cd = (batch[k].cpu() for k in ("elevation", "azimuth", "camera_distances"))
    cpu_pu = type(pu)(pu.text_embeddings_vd.cpu(), pu.uncond_text_embeddings_vd.cpu(), front_threshold=pu.front_threshold,
                      back_threshold=pu.back_threshold)
    temb, negw = cpu_pu.get_text_embeddings_perp_neg(el, az, cd, True)
    temb = torch.cat([temb[0:1], temb[1:2], temb[2:4], temb[0:1]], 0)
    negw = negw * -1 * guid.cfg.guidance_perp_neg
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(1, 4, 64, 64, generator=g)
    t = torch.randint(guid.min_step, guid.max_step + 1, (1,), generator=g)
    t_plus = (t + (0.1 * (t - guid.min_step)).long()).clamp(1, 999)
    timings = {}
    t0 = time.perf_counter()
    ref_step.asd_step(P, up, layout, ucfg, vp, vplan, temb, negw, noise, t, t_plus, torch.randn(1, 4, 64, 64, generator=g),
                      timings=timings)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 5), "unit": "steps/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "1 full asd_sd_nerf step (4096 rays x 512 spp render fwd+bwd, VAE 512^2 fwd+bwd, UNet batch 5), "
                      "oracle C/OpenMP renderer + torch fp32 diffusion, weight generation excluded",
            "seconds": round(dt, 2), "phases_s": {k: round(v, 3) for k, v in timings.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backend", default=os.environ.get("ASD_BACKEND", "hip"), choices=["hip", "eager"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="also report per-phase milliseconds (adds syncs; untimed extra steps)")
    args = ap.parse_args()

    from scaledreamer_amd import dist as asd_dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    if world > 1:
        asd_dist.init_from_env("nccl")
    dev = torch.device("cuda", local_rank)

    # per-rank seed = cfg.seed + rank (launch.py:171): different cameras / noise / t per rank
    cfg, system, data = build_system(args.backend, seed=10 + rank)
    asd_dist.broadcast_parameters(system)  # identical initial parameters (DDP wrap-time broadcast)

    def step():
        batch = to_device(data.collate(), dev)
        return system.train_one_step(batch), batch

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, batch = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        steps_per_s = world * args.steps / dt
        out = {
            "metric": "ASD train steps/sec (64x64 render, SD2.1)", "value": round(steps_per_s, 4), "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 renderer / f16 diffusion (f32 accumulate)", "data": "synthetic",
            "rays_per_sec": round(steps_per_s * 4096, 1),
            "config": {"workload": "asd_sd_nerf: 1 view/GPU, 64x64 rays, 512 spp occgrid march, implicit-volume iNGP "
                                   "(16x2 hash grid 2^19, MLP 64), SD-2.1 UNet batch 5 (CFG+Perp-Neg+shifted t), VAE 512^2 "
                                   "fwd+bwd, AdamW", "views_per_gpu": 1, "parallelism": f"dp{world}",
                       "diffusion_backend": args.backend, "diffusion_weights": "seeded random init"},
            "loss": float(loss.item()), "kept_samples_last_step": int(system.renderer.last_n_samples) if hasattr(system.renderer, "last_n_samples") else None,
        }
        out["roofline"] = roofline_field_kernel(system, batch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(system, batch, seed=10)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
