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


def build_system(backend: str, seed: int, workload: str = "asd_sd_nerf"):
    from scaledreamer_amd import presets
    from scaledreamer_amd.data import RandomCameraIterableDataset, RandomMultiviewCameraIterableDataset

    from scaledreamer_amd.guidance import PromptUtils
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    if backend == "eager":   # A/B tool: the library-op backend is registered only on demand (tools/eager_backend.py)
        import tools.eager_backend  # noqa: F401
    mv = workload == "asd_mv_nerf"
    if workload in ("asd_sd_hyper_ingp", "asd_sd_3dconv_net", "asd_mv_triplane"):
        return build_hyper_system(backend, seed, workload)
    with presets.random_weights_allowed():   # synthetic prior: no checkpoint exists offline (the JSON line says "data": "synthetic")
        cfg = presets.asd_mv_nerf() if mv else presets.asd_sd_nerf(guidance_backend=backend)
    torch.manual_seed(seed)
    random.seed(seed)
    pp = cfg["system"]["prompt_processor"]
    dev = torch.device("cuda", torch.cuda.current_device())
    prompt_utils = PromptUtils.synthetic(seed=1234, device=dev, front_threshold=pp["front_threshold"],
                                         back_threshold=pp["back_threshold"])
    system = find(cfg["system_type"])(cfg["system"], prompt_utils=prompt_utils)
    system.train()
    data = (RandomMultiviewCameraIterableDataset if mv else RandomCameraIterableDataset)(cfg["data"])
    return cfg, system, data


def build_hyper_system(backend: str, seed: int, workload: str = "asd_sd_hyper_ingp"):
    """the multi-prompt amortized configs (secondary workloads): Hyper-iNGP, 3DConv-net (StyleGAN-3D volume), triplane transformer"""
    from scaledreamer_amd import presets
    from scaledreamer_amd.multiprompt import SyntheticMultiPromptProcessor

    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401

    with presets.random_weights_allowed():
        cfg = {"asd_sd_hyper_ingp": lambda: presets.asd_sd_hyper_ingp(guidance_backend=backend),
               "asd_sd_3dconv_net": lambda: presets.asd_sd_3dconv_net(guidance_backend=backend),
               "asd_mv_triplane": lambda: presets.asd_mv_triplane_transformer(n_gpus=int(os.environ.get("WORLD_SIZE", "1")))}[workload]()
    torch.manual_seed(seed)
    random.seed(seed)
    dev = torch.device("cuda", torch.cuda.current_device())
    pp = cfg["system"]["prompt_processor"]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    data = find(cfg["data_type"])(cfg["data"], rank=rank, n_ranks=world)
    proc = SyntheticMultiPromptProcessor(cfg["data"]["prompt_library"]["train"], seed=1234, device=dev,
                                         front_threshold=pp.get("front_threshold", 45.0), back_threshold=pp.get("back_threshold", 45.0),
                                         use_local_text_embeddings=pp.get("use_local_text_embeddings", False),
                                         use_perp_neg=pp.get("use_perp_neg", False))
    system = presets.apply_trainer(find(cfg["system_type"])(cfg["system"], prompt_processor=proc), cfg)
    system.train()
    return cfg, system, data


def build_stub_system(seed: int):
    """ASD_BENCH_STUB=1 — NOT a benchmark: a CPU stand-in (tiny torch renderer + quadratic guidance) behind the real
    StableDreamer.train_one_step / GradientExchange wiring, so that the multi-process contract of this script (per-rank seeds,
    parameter broadcast, barrier order, max-over-ranks timing, rank-0-only JSON line, exchange timing) is exercised on CPU with gloo
    by tests/test_dist_cpu.py before the first N > 1 run on real GPUs.  Its JSON line says `"data": "stub"`."""
    from scaledreamer_amd.base import Updateable
    from scaledreamer_amd.config import ConfigDict
    from scaledreamer_amd.system import StableDreamer

    torch.manual_seed(seed)

    class Renderer(torch.nn.Module, Updateable):
        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(1_500_000) * 1e-3)       # > IN_PLACE_BYTES: its own in-place unit
            self.mlp = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))

        def forward(self, rays_d, **kw):
            rgb = torch.sigmoid(self.mlp(rays_d) + self.table[:3])
            return {"comp_rgb": rgb, "opacity": rgb.mean(-1, keepdim=True).clamp(0, 1)}

    class Guidance(Updateable):
        def __call__(self, rgb, prompt_utils, rgb_as_latents=False, **batch):
            return {"loss_asd": (rgb ** 2).sum(), "grad_norm": rgb.detach().norm()}

    class Data:
        def collate(self):
            return {"rays_d": torch.randn(1, 8, 8, 3)}

    s = object.__new__(StableDreamer)
    torch.nn.Module.__init__(s)
    s.cfg = ConfigDict(stage="coarse", loss=ConfigDict(lambda_asd=1.0, lambda_orient=0.0, lambda_sparsity=2.0, lambda_opaque=0.0, lambda_z_variance=0.0))
    s.current_epoch, s.true_global_step, s.logged = 0, 0, {}
    s.renderer, s.guidance, s.prompt_utils = Renderer(), Guidance(), None
    s.optimizer = torch.optim.SGD(s.renderer.parameters(), lr=0.1)
    return {}, s, Data()


def to_device(batch, dev):
    return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


# the kernel with the largest share of GPU time in the committed rocprofv3 summary of `python bench.py` (first row of
# profiles/r06_kernel_stats.csv) and inside the timed steps (profiles/r06_step_breakdown.txt).  Round 6: the cold re-tune of the plan table
# moved seven of the VAE's 24 ping-pong launches to other window kernels, so gemm_f16_kernel<256,64> (the 64x64 / 32x32 levels' linears)
# now leads both tables; the ping-pong convolution keeps its own line (`roofline_pp_conv`).
DOMINANT = "gemm256"
DOMINANT_SOURCE = ("profiles/r06_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py`): first row gemm_f16_kernel<256,64,4,2> (11.05 % of the "
                   "run, 26.7 us average); inside the timed steps (profiles/r06_step_breakdown.txt) it is first too (55 launches on 15 shapes: the linears of the "
                   "UNet's 64x64 and 32x32 transformer blocks, 1.48 ms of each step), then conv3x3_pp_kernel<4,4> (17 launches, 1.22 ms: `roofline_pp_conv`), "
                   "attention, and the hash-grid gradient (`roofline_field_bwd`)")

# every launch of gemm_f16_kernel<256,64,4,2> (plain GEMM) in one step (tools/gemm_shapes.py trace, gpurun_out/gemm_order.txt of the round-6 tree):
# (M, N, K, act [2 = fused GEGLU: output N / 2 columns], residual, GroupNorm records in the epilogue, launches per step)
GEMM256_LAUNCHES = [(20480, 320, 320, 0, 0, 0, 9), (20480, 320, 320, 0, 1, 0, 9), (20480, 2560, 320, 2, 0, 0, 5), (20480, 320, 1280, 0, 1, 0, 5),
                    (20480, 320, 320, 0, 1, 1, 5), (5120, 1280, 640, 0, 0, 0, 5), (5120, 5120, 640, 2, 0, 0, 5), (20480, 640, 320, 0, 0, 0, 4),
                    (20480, 320, 640, 0, 0, 0, 2), (65536, 256, 128, 0, 0, 0, 1), (16384, 512, 256, 0, 0, 0, 1), (400, 12480, 1024, 0, 0, 0, 1),
                    (320, 8192, 320, 0, 0, 0, 1), (20480, 320, 960, 0, 0, 0, 1), (65536, 128, 256, 0, 1, 0, 1)]

# every launch of conv3x3_pp_kernel<4,4> in one step (tools/gemm_shapes.py trace of the step, gpurun_out/gemm_shapes.txt): (H = W, Cin, Cout,
# residual, GroupNorm records in the epilogue: 0 none (the input-gradient launches) / 1 forward statistics / 2 the backward reductions of the
# GroupNorm in front (off by default since round 3, csrc/net.hip), launches per step)
PP44_LAUNCHES = [(512, 128, 128, 0, 0, 4), (512, 128, 128, 1, 1, 2), (512, 128, 128, 0, 1, 2), (128, 512, 512, 0, 0, 3), (128, 512, 512, 1, 1, 2),
                 (128, 512, 512, 0, 1, 1), (256, 256, 256, 0, 0, 3), (256, 256, 256, 1, 1, 2), (256, 256, 256, 0, 1, 1), (128, 256, 512, 0, 1, 1),
                 (256, 128, 256, 0, 1, 1), (256, 256, 128, 0, 0, 1), (512, 32, 128, 0, 1, 1)]
PROFILE_ROUND = "r06"


def pmc_traffic_of(*kernel_substrs: str):
    """HBM bytes per launch of a kernel (or the sum over several kernels that form one span) from the COMMITTED per-kernel PMC summaries of this
    round (profiles/r06_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv: separate rocprofv3 --pmc passes of `python bench.py`, tools/r6_pmc.sh; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for wide reads on gfx950) — read at run time, so the figure follows the profile committed next to
    the code; None when a kernel is missing from either file (no fall-back to an older round)."""
    import csv

    tot = 0.0
    for name in kernel_substrs:
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_{cnt}_per_kernel.csv")
            val = None
            if os.path.exists(path):
                with open(path) as fh:
                    for row in csv.reader(fh):          # kernel, dispatches, sum KB, avg KB, avg MB per dispatch (FETCH_SIZE already doubled there)
                        if len(row) >= 5 and name in row[0]:
                            val = float(row[4]) * 1e6
                            break
            if val is None:
                return None
            tot += val
    return tot


def pmc_span_traffic(span_kernel: str, *other_kernels: str):
    """HBM bytes per SPAN of several kernels (one launch of `span_kernel` + however many launches of the others belong to it: the paged scatter runs
    once or twice per field gradient, depending on the row capacity) from the committed per-kernel PMC summaries: the kernels' summed counters divided by
    the dispatches of `span_kernel`; FETCH_SIZE doubled (see pmc_traffic_of).  None when a kernel is missing."""
    import csv

    tot, spans = 0.0, None
    for cnt, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_{cnt}_per_kernel.csv")
        if not os.path.exists(path):
            return None
        rows = [r for r in csv.reader(open(path)) if len(r) >= 5 and r[1].isdigit()]
        for name in (span_kernel,) + other_kernels:
            hit = next((r for r in rows if name in r[0]), None)
            if hit is None:
                return None
            tot += float(hit[2]) * 1024.0 * scale
            if name == span_kernel:
                if spans is not None and spans != int(hit[1]):
                    return None          # the two passes must have seen the same launches
                spans = int(hit[1])
    return tot / spans if spans else None


def pmc_span_samples():
    """samples per field-gradient span in the PMC passes behind pmc_span_traffic (profiles/r06_pmc_field_span.json: the sample count drifts with the
    occupancy grid, so the counters' bytes belong to THAT count, not to the one of this run)"""
    import json

    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_field_span.json")
    return json.load(open(path)).get("samples_per_launch") if os.path.exists(path) else None


def pmc_shape_traffic(which: str):
    """HBM bytes per launch of one of the single-shape loops below ("gemm", "vae512", "unet64") from this round's per-shape PMC passes
    (profiles/r06_pmc_shapes.json: tools/r6_pmc_shapes.sh — separate FETCH_SIZE / WRITE_SIZE passes of that loop alone, FETCH_SIZE doubled); None without it"""
    import json

    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_shapes.json")
    if not os.path.exists(path):
        return None
    v = json.load(open(path)).get(which, {}).get("bytes_per_launch")
    return None if v is None else round(v)


def trace_mark():
    """an empty launch named asd_trace_mark_kernel on the current stream: the per-step tables of profiles/ are cut between two of them"""
    import ctypes as C

    from scaledreamer_amd._lib import lib

    lib().asd_probe_mark(C.c_void_p(torch.cuda.current_stream().cuda_stream))


def roofline_pp_kernel(reps: int = 3):
    """conv3x3_pp_kernel<4,4> (csrc/gemm_pp.hip) over ALL its launches of one step, each with the epilogue it has in the step (residual,
    GroupNorm statistics / backward reductions), back to back on the launch stream between HIP events.  achieved = algorithmic flops of
    the 24 launches (2 M N K each) / their summed duration; avg_launch_ms is what the rocprofv3 summary's average for
    this kernel has to agree with."""
    from scaledreamer_amd.diffusion import hip_ops as H

    flops = secs = launches = 0.0
    alg_bytes = 0.0
    skipped = []
    for hw, cin, cout, res, gn, count in PP44_LAUNCHES:
        plan = tuple(H.plan_table().get((hw * hw, cout, 9 * cin, (hw, cin, 1, 0, 1)), (0, 1)))
        if plan != (24, 1):
            skipped.append((hw, cin, cout))
            continue
        # HBM-cold operands, as inside the step (the review of round 3: replaying ONE operand set back to back keeps <= 67 MB resident in
        # the 256 MB Infinity Cache and reads 6 % fast): rotate through enough distinct sets to exceed the cache twice over
        set_bytes = 2.0 * hw * hw * (cin + cout * (1 + (1 if res else 0) + (1 if gn == 2 else 0)))
        nsets = int(min(12, max(2, -(-600e6 // set_bytes))))
        sets = []
        for _ in range(nsets):
            x = torch.randn(1, hw, hw, cin, device="cuda").half()
            w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * (9 * cin) ** -0.5)
            kw = dict(gn_rows=hw * hw) if gn else {}
            if res:
                kw["residual"] = torch.randn(hw * hw, cout, device="cuda").half()
            if gn == 2:
                xg = torch.randn(1, hw * hw, cout, device="cuda").half()
                gamma, beta = torch.ones(cout, device="cuda").half(), torch.zeros(cout, device="cuda").half()
                _, fstats = H.groupnorm(xg, gamma, beta, 1e-6, True, return_stats=True)
                kw["gn_bwd"] = dict(x=xg, fstats=fstats, gamma=gamma, beta=beta, eps=1e-6, silu=True)
            sets.append((x, w, kw))
        for x, w, kw in sets[:2]:
            H.conv3x3(x, w, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps * count):
            x, w, kw = sets[i % nsets]
            H.conv3x3(x, w, **kw)
        e1.record()
        torch.cuda.synchronize()
        secs += e0.elapsed_time(e1) * 1e-3 / reps
        flops += count * 2.0 * hw * hw * cout * 9 * cin
        alg_bytes += count * 2.0 * hw * hw * (cin + cout * (1 + (1 if res or gn == 2 else 0)))
        launches += count
        del sets
    if launches == 0:
        return None
    achieved = flops / secs / 1e12
    return {"kernel": "conv3x3_pp_kernel<4,4> (ping-pong LDS-window 3x3 convolution, csrc/gemm_pp.hip: 16x16-pixel patch x 128 channels, eight waves in two groups "
                      "staggered by a barrier, counted vmcnt, 32-channel k-steps); all its launches of one step (VAE encoder forward + input gradient), each with its "
                      "step epilogue", "bound": "mfma",
            "achieved": round(achieved, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_PEAK_TF, 4),
            "traffic": pmc_traffic_of("conv3x3_pp_kernel<4, 4>"),
            "traffic_unit": "bytes/launch, average over the kernel's launches (PMC FETCH_SIZE x2 + WRITE_SIZE, read from profiles/r06_pmc_*_per_kernel.csv at run time)",
            "operands": "HBM-cold: >= 600 MB of distinct operand sets rotated per shape (as inside the step)",
            "launches_per_step": int(launches), "flops_per_launch": flops / launches, "algorithmic_bytes_per_launch": alg_bytes / launches,
            "avg_launch_ms": round(secs / launches * 1e3, 4), "ms_per_step": round(secs * 1e3, 3),
            "shapes_not_on_this_kernel_under_the_loaded_plans": skipped}


def roofline_gemm256_kernel(reps: int = 3):
    """gemm_f16_kernel<256,64,4,2> (csrc/gemm.hip) over ALL its 55 launches of one step, each with the epilogue it has in the step (bias,
    residual, fused GEGLU, GroupNorm records), back to back between HIP events with HBM-cold operands (>= 600 MB of distinct sets rotated per
    shape).  These are the K <= 1280 linears of the 64x64 / 32x32 transformer blocks: A + W + C (+ residual) once through HBM bounds them
    (sum of bytes / 8 TB/s = 283 us per step against sum of flops / 2.5 PF/s = 255 us), so achieved = algorithmic bytes / summed duration;
    avg_launch_ms is what the rocprofv3 summary's average for this kernel has to agree with."""
    from scaledreamer_amd.diffusion import hip_ops as H

    bytes_ = flops = secs = launches = 0.0
    for M, N, K, act, res, gn, count in GEMM256_LAUNCHES:
        n_out = N // 2 if act == 2 else N
        set_bytes = 2.0 * (M * K + N * K + M * n_out * (1 + res))
        nsets = int(min(24, max(2, -(-600e6 // set_bytes))))
        sets = []
        for _ in range(nsets):
            a = torch.randn(M, K, device="cuda").half()
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
            b = torch.randn(N, device="cuda").half()
            if act == 2:
                w, b = H.pack_geglu_weight(w, b)
            kw = dict(bias=b, act=act, tile_cfg=3, split_k=1)
            if res:
                kw["residual"] = torch.randn(M, n_out, device="cuda").half()
            if gn:
                kw["gn_rows"] = 4096
            sets.append((a, w.contiguous(), kw))
        for a, w, kw in sets[:2]:
            H.gemm(a, w, **kw)
        torch.cuda.synchronize()
        # GPU-paced: these launches are shorter than the host side of an eager call, so the pass over the operand sets is captured into a HIP
        # graph and the replay is timed (events on the stream the graph is launched on)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps * count):
                a, w, kw = sets[i % nsets]
                H.gemm(a, w, **kw)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        secs += e0.elapsed_time(e1) * 1e-3 / reps
        del g
        bytes_ += count * set_bytes
        flops += count * 2.0 * M * N * K
        launches += count
        del sets
    achieved = bytes_ / secs / 1e9
    return {"kernel": "gemm_f16_kernel<256,64,4,2> (csrc/gemm.hip: 256 x 64 tile, eight waves, LDS-DMA operand tiles, two stages; bias / residual / fused GEGLU / "
                      "GroupNorm-records epilogues); all its 55 launches of one step — the linears of the UNet's 64x64 and 32x32 transformer blocks (proj_in / "
                      "proj_out, q|k, q, attention output projections, GEGLU projection, ff.net.2) — each with its step epilogue",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic_of("gemm_f16_kernel<256, 64, 4, 2, false, 2, 1, false>"),
            "traffic_unit": "bytes/launch, average over the kernel's launches (PMC FETCH_SIZE x2 + WRITE_SIZE, read from profiles/r06_pmc_*_per_kernel.csv at run time)",
            "operands": "HBM-cold: >= 600 MB of distinct operand sets rotated per shape (as inside the step)",
            "launches_per_step": int(launches), "algorithmic_bytes_per_launch": bytes_ / launches, "flops_per_launch": flops / launches,
            "mfma_floor_fraction": round(flops / 2.5e15 / secs, 4),
            "avg_launch_ms": round(secs / launches * 1e3, 4), "ms_per_step": round(secs * 1e3, 3)}


def roofline_conv3d(reps: int = 6):
    """conv3d_pp_kernel<2> (csrc/conv3d.hip) on the generator's heaviest layer, 64 -> 64 channels at 128^3 (forward and input gradient of
    SynthesisBlock(128).conv1: 2 launches per C4 step): duration of the kernel alone between HIP events on the launch stream (asd_probe_events;
    the split / pack passes of the call stay outside), fresh operands every launch.  Flops: 2 M N K fp32-equivalent, x 3 fp16 MFMA products."""
    import ctypes as C
    from scaledreamer_amd import ops
    from scaledreamer_amd._lib import lib

    R, Cc = 128, 64
    sets = [(torch.randn(1, R, R, R, Cc, device="cuda"), torch.randn(1, Cc, Cc, 3, 3, 3, device="cuda") * (27 * Cc) ** -0.5) for _ in range(2)]
    ops.conv3d_fwd(*sets[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize()
    lib().asd_probe_events(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event))
    ms = []
    try:
        for i in range(reps):
            ops.conv3d_fwd(*sets[i % 2])
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
    finally:
        lib().asd_probe_events(None, None)
    t = sum(ms) / len(ms)
    f32_flops = 2.0 * R ** 3 * Cc * 27 * Cc
    achieved = 3 * f32_flops / (t * 1e-3) / 1e12
    return {"kernel": "conv3d_pp_kernel<2> (split-fp16 3x3x3 convolution, csrc/conv3d.hip: 16x16-voxel patch x 64 channels, both fp16 planes of the 18x18 window "
                      "LDS-resident per (depth tap, 32-channel chunk), three MFMA products per fragment pair) on 64->64 @128^3", "bound": "mfma",
            "achieved": round(achieved, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s (fp16 products)", "frac": round(achieved / MFMA_F16_PEAK_TF, 4),
            "fp32_equivalent_tflops": round(achieved / 3, 1), "traffic": None, "flops_per_launch": 3 * f32_flops, "avg_launch_ms": round(t, 4),
            "algorithmic_bytes_per_launch": R ** 3 * Cc * (4 + 4)}


def roofline_conv_kernel(which: str = "vae512", reps: int = 30):
    """The dominant kernel family of the step is the fp16 MFMA convolution (rocprofv3, profiles/r01_final7_kernel_stats_top70.csv:
    conv3x3_win2_kernel<128> + conv3x3_win_kernel<128> + conv3x3_win2_kernel<64> + conv3x3_win_kernel<64> 19.4 % of the kernel
    time, gemm_f16_kernel variants 27 %).  Average duration of one launch, HIP events on
    the launch stream, on its heaviest single shape:
      vae512: 3x3 conv 128->128 at 512x512 (VAE encoder level 0, forward and input-gradient: 8 launches per step) M=262144 N=128 K=1152
      unet64: 3x3 conv 320->320 at 64x64 for the UNet batch of 5 (7 launches per UNet forward)                 M=20480  N=320 K=2880
    Algorithmic flops per launch = 2*M*N*K."""
    from scaledreamer_amd.diffusion import hip_ops as H

    B, hw, cin, cout = (1, 512, 128, 128) if which == "vae512" else (5, 64, 320, 320)
    x = torch.randn(B, hw, hw, cin, device="cuda").half()
    w = H.pack_conv3x3_weight(torch.randn(cout, cin, 3, 3, device="cuda").half() * 0.02)
    for _ in range(5):
        H.conv3x3(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        H.conv3x3(x, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * B * hw * hw * cout * cin * 9
    achieved = flops / (ms * 1e-3) / 1e12
    plan = tuple(H.plan_table().get((B * hw * hw, cout, 9 * cin, (hw, cin, 1, 0, 1)), (0, 1)))
    if plan[0] - 1 in H.PP_TILES:
        t = plan[0] - 1
        kern = (f"conv3x3_pp_kernel<{H.TILE_BM[t] // 64},{H.TILE_BN[t] // 32}> (ping-pong LDS-window kernel, csrc/gemm_pp.hip: {H.TILE_BM[t] // 16}x16-pixel patch x "
                f"{H.TILE_BN[t]} channels, eight waves in two groups staggered by a barrier, counted vmcnt, 32-channel k-steps)")
    elif plan[0] - 1 in H.WINDOW_TILES:
        kname = "conv3x3_win2_kernel" if plan[0] - 1 >= 10 else "conv3x3_win_kernel"
        kern = f"{kname}<{H.TILE_BN[plan[0] - 1]}> (16x16-pixel patch, LDS-resident 18x18 input window{', two blocks per CU' if plan[0] - 1 >= 10 else ''})"
    else:
        kern = f"gemm_f16_kernel<{H.TILE_BM[plan[0] - 1]}x{H.TILE_BN[plan[0] - 1]},conv>" if plan[0] else "gemm_f16_kernel<model tile,conv>"
    shape = "3x3 conv 128->128 @512x512 (VAE encoder)" if which == "vae512" else "3x3 conv 320->320 @64x64, UNet batch 5"
    return {"kernel": f"{kern} split_k={plan[1]} on {shape}; autotuned plan", "bound": "mfma",
            "achieved": round(achieved, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_PEAK_TF, 4),
            "traffic": pmc_shape_traffic(which), "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE of this loop alone, operands cache-warm: profiles/r06_pmc_shapes.json)",
            "flops_per_launch": flops, "algorithmic_bytes_per_launch": 2.0 * (B * hw * hw * (cin + cout) + 9 * cin * cout),
            "avg_launch_ms": round(ms, 4)}


def roofline_gemm_kernel(reps: int = 50):
    """The plain GEMMs of the K <= 1280 transformer linears are the largest MFMA family after the window convolutions
    (profiles/r01_final7_kernel_stats_top70.csv: gemm_f16_kernel<64,64> 10.8 % + <256,64> 4.9 %, ~155 launches per step).  Their most
    frequent shape, the 320 -> 320 linear on the 64x64 tokens of the UNet batch (M=20480, N=320, K=320; 25 launches per step), has
    161 FLOP per algorithmic byte, below the machine balance of 2500 / 8 = 312: HBM-bound.  Algorithmic bytes = A + W + C in fp16, each touched once."""
    from scaledreamer_amd.diffusion import hip_ops as H

    M, N, K = 20480, 320, 320
    a, w = torch.randn(M, K, device="cuda").half(), torch.randn(N, K, device="cuda").half()
    for _ in range(5):
        H.gemm(a, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        H.gemm(a, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 2.0 * (M * K + N * K + M * N)
    plan = tuple(H.plan_table().get((M, N, K, K), (0, 1)))   # plan key of a plain GEMM: (M, N, K, lda)
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": f"gemm_f16_kernel<{H.TILE_BM[plan[0] - 1]}x{H.TILE_BN[plan[0] - 1]}> split_k={plan[1]} on linear 320->320, M=20480 (UNet 64x64 tokens x batch 5)"
                      if plan[0] else "gemm_f16_kernel<model tile> on linear 320->320, M=20480",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_shape_traffic("gemm"), "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE of this loop alone, operands cache-warm: profiles/r06_pmc_shapes.json)",
            "bytes_per_launch": nbytes,
            "avg_launch_ms": round(ms, 4)}


def roofline_field_kernel(system, batch, reps: int = 20):
    """Average duration (HIP events on the launch stream) of the dominant hand-written renderer kernel,
    field_fwd_kernel, on the live samples of one step; algorithmic bytes per DESIGN.md / SURVEY.md §8d."""
    from scaledreamer_amd import ops

    ren, geo = system.renderer, system.geometry
    with torch.no_grad():
        ri, t0, t1, pts, dirs, off, cnt, _ = ren._sample(batch["rays_o"].reshape(-1, 3).contiguous(), batch["rays_d"].reshape(-1, 3).contiguous())
    n = int(pts.shape[0])
    if n == 0:
        return None
    grid = geo.encoding.encoding.encoding.params.detach()
    w = [t.detach() for t in geo._weights()]
    # as inside the step (asd_render_fwd): no finite-difference normal (lambda_orient = 0 in this configuration), one encode per sample
    for _ in range(3):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # per kept sample: 16 levels x 8 corners x 8 B gathered + 12 B position in + 16 B (sigma, features) out
    # + 128 B centre encoding saved for the backward pass
    bytes_per_sample = 1024 + 12 + 16 + 128
    achieved = n * bytes_per_sample / (ms * 1e-3) / 1e9
    return {"kernel": "field_fwd_kernel<16,64,3> (one encode per sample, no normal: the form asd_render_fwd launches)", "bound": "hbm",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "samples_per_launch": n,
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_sample": bytes_per_sample}


def roofline_field_bwd(system, batch, reps: int = 10):
    """asd_field_bwd (field_bwd_sample_kernel + the tall-skinny weight-gradient GEMM) on the live samples of one step: the hash-table
    gradient scatter.  Algorithmic bytes per kept sample (SURVEY.md §8d, C2 with lambda_orient = 0: one encode's scatter): 128 corner
    updates x 8 B + 128 B saved encoding + 16 B position / sigma in; the bound that actually binds is the request rate of the
    atomic units (tools/atomic_probe2.hip: ~21 G requests/s whatever their width), reported next to the HBM fraction."""
    from scaledreamer_amd import ops

    ren, geo = system.renderer, system.geometry
    with torch.no_grad():
        ri, t0, t1, pts, dirs, off, cnt, _ = ren._sample(batch["rays_o"].reshape(-1, 3).contiguous(), batch["rays_d"].reshape(-1, 3).contiguous())
        n = int(pts.shape[0])
        if n == 0:
            return None
        grid = geo.encoding.encoding.encoding.params.detach()
        w = [t.detach() for t in geo._weights()]
        sigma, feats, normal, enc = ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, False)
        g = torch.Generator(device="cuda").manual_seed(0)
        d_sigma, d_feats = torch.randn(n, device="cuda", generator=g), torch.randn(n, 3, device="cuda", generator=g)
        d_grid = torch.zeros_like(grid)
        for _ in range(2):
            ops.field_bwd(geo._meta, geo._fcfg, grid, *w, pts, enc, sigma, d_sigma, d_feats, None, d_grid)
        # the scatter kernel alone: the library records the caller's two events immediately around field_bwd_sample_kernel on the
        # launch stream (asd_probe_events), the other launches of the call (weight-gradient GEMM, slab reductions) stay outside
        import ctypes as C
        from scaledreamer_amd._lib import lib
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()          # creates the hipEvent_t handles
        torch.cuda.synchronize()
        lib().asd_probe_events(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event))
        ms_k = []
        try:
            for _ in range(reps):
                d_grid.zero_()           # as in the step: the gradient table is cleared once per step, the scatter starts on clean lines
                ops.field_bwd(geo._meta, geo._fcfg, grid, *w, pts, enc, sigma, d_sigma, d_feats, None, d_grid)
                torch.cuda.synchronize()
                ms_k.append(e0.elapsed_time(e1))
        finally:
            lib().asd_probe_events(None, None)
        t0_, t1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0_.record()
        for _ in range(reps):
            ops.field_bwd(geo._meta, geo._fcfg, grid, *w, pts, enc, sigma, d_sigma, d_feats, None, d_grid)
        t1_.record()
        torch.cuda.synchronize()
    ms = sum(ms_k) / len(ms_k)
    ms_call = t0_.elapsed_time(t1_) / reps
    bytes_per_sample = 128 * 8 + 128 + 16
    achieved = n * bytes_per_sample / (ms * 1e-3) / 1e9
    # PMC: the span is field_bwd_sample_kernel (MLP backward + the coarse levels' run-aggregated atomics) + pg_fill_kernel + pg_accum_kernel (the
    # paged scatter of the eleven hashed levels); bytes per launch from this round's per-kernel passes, where every launch of these kernels has
    # the step's sample count
    traffic = pmc_span_traffic("field_bwd_sample_kernel", "field_bwd_mlp_mfma_kernel", "pg_fill_kernel", "pg_accum_kernel")
    return {"kernel": "asd_field_bwd's gradient span: field_bwd_mlp_mfma_kernel (csrc/field_mfma.hip: both MLP heads' backward and the first-layer weight gradient as "
                      "split-fp16 MFMA products, 64 samples per wave) + field_bwd_sample_kernel<16,64,3,0,PRE> (scatter only: the dense levels 0-4 as run-aggregated "
                      "fp32 atomics with per-XCD copies of levels 0-2) + pg_fill_kernel + pg_accum_kernel (csrc/field_paged.hip: the eleven hashed levels 5-15 binned by "
                      "64 KB table page, one workgroup per page, tag-arbitrated plain LDS adds — no global atomics); one span per step", "bound": "hbm",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None if traffic is None else round(traffic), "traffic_samples_per_launch": pmc_span_samples(),
            "traffic_unit": "bytes/span at traffic_samples_per_launch samples (PMC FETCH_SIZE x2 + WRITE_SIZE summed over the four kernels' launches / spans, profiles/r06_pmc_*_per_kernel.csv)",
            "samples_per_launch": n, "avg_launch_ms": round(ms, 4), "asd_field_bwd_call_ms": round(ms_call, 4),
            "algorithmic_bytes_per_sample": bytes_per_sample, "algorithmic_bytes_per_launch": n * bytes_per_sample,
            "binding_limit": "inside the span (round 6): the scatter-only sample kernel (0.19 ms: ~3.5 M coarse-level atomic requests), the matrix-pipe MLP pass "
                             "(0.17 ms incl. the first-layer weight gradient), then the item stream of the paged scatter (128 B per sample and hashed level written + "
                             "read) and its 704 pages on 512 workgroup slots (DESIGN.md 4.7, 4.14)"}


def cpu_baseline(system, batch, seed: int):
    """The oracle (C/OpenMP renderer port + torch fp32 diffusion restatement) timed on this host's cores on ONE full step of the
    same workload: render forward+backward of the step's camera (4096 rays x 512 spp), the UNet at batch 5 (77-token context) and
    the VAE encoder forward + input gradient at 512x512.  Weight generation is excluded.  ~10-30 s of CPU work."""
    import platform

    import numpy as np
    from oracle import diffusion_ref as D
    from oracle import ref_renderer as R
    from scaledreamer_amd.diffusion import weights as W

    # 32 threads: measured on the 2 x 64-core EPYC 9575F box (256 hardware threads), torch's CPU kernels are 10x SLOWER with all 256
    # threads (UNet batch 5: 164 s vs ~8 s; gpurun_out/r2/bench_full1.json) — oversubscribed oneDNN convolutions on small images
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
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
    up, vp = W.gen_params(layout[0], guid.cfg.weights_seed), W.gen_params(vshapes, guid.cfg.weights_seed + 1)

    def one_step():
        tm = {}
        t0 = time.perf_counter()
        out, ctx = R.forward(P)
        tm["render_fwd"] = time.perf_counter() - t0
        rng = np.random.default_rng(seed)
        t0 = time.perf_counter()
        R.backward(P, ctx, d_comp_rgb=rng.normal(size=(4096, 3)).astype(np.float32), d_opacity=rng.normal(size=(4096, 1)).astype(np.float32))
        tm["render_bwd"] = time.perf_counter() - t0
        g = torch.Generator().manual_seed(seed)
        x, ctx_e = torch.randn(5, 4, 64, 64, generator=g), torch.randn(5, 77, 1024, generator=g)
        with torch.no_grad():
            t0 = time.perf_counter()
            D.unet_forward(up, layout, ucfg, x, torch.tensor([700, 700, 700, 700, 730]), ctx_e)
            tm["unet_fwd_b5"] = time.perf_counter() - t0
        img = torch.rand(1, 3, 512, 512, generator=g).requires_grad_(True)
        t0 = time.perf_counter()
        m = D.vae_encode_moments(vp, vplan, img * 2 - 1)
        tm["vae_fwd_512"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        m.sum().backward()
        tm["vae_bwd_512"] = time.perf_counter() - t0
        return tm

    # one warm-up step (page-in, oneDNN primitive caches, OpenMP pool), then three timed ones; the median is reported
    one_step()
    runs = [one_step() for _ in range(3)]
    totals = sorted(sum(r.values()) for r in runs)
    tm = runs[[sum(r.values()) for r in runs].index(totals[1])]
    total = totals[1]
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), cpu_model)
    except OSError:
        pass
    return {"value": round(1.0 / total, 5), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": "median of THREE full steps after one warm-up step: render fwd+bwd (4096 rays x 512 spp) + UNet batch 5 + VAE fwd+input-grad "
                      "at 512^2; oracle C/OpenMP renderer + torch fp32 diffusion restatement; weight generation excluded",
            "step_seconds": round(total, 2), "step_seconds_all": [round(t, 2) for t in totals], "cpu_model": cpu_model, "host_threads_available": os.cpu_count(), "phases_s": {k: round(v, 3) for k, v in tm.items()}}


def dist_evidence(world, rank, local_rank, dev, ex):
    """What a multi-GPU line must carry so that it proves itself: the world size as an all-reduce of ones sees it (RCCL, not the
    environment), every rank's device, the bytes one gradient exchange moves per rank and how the exchange is cut (launch.py:233-240:
    the reference's DDP all-reduce of the field parameters).  On one GPU: the same keys with the trivial values."""
    import socket

    mine = {"rank": rank, "local_rank": local_rank, "device_index": int(torch.cuda.current_device()) if dev.type == "cuda" else -1,
            "device": torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu", "host": socket.gethostname(),
            "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None) if dev.type == "cuda" else None}
    out = {"rccl_ranks": 1, "ranks": [mine], "allreduce_bytes": None, "allreduce_units": None, "collective_backend": None}
    if world > 1:
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        every = [None] * world
        torch.distributed.all_gather_object(every, mine)
        out.update(rccl_ranks=int(ones.item()), ranks=every, collective_backend=torch.distributed.get_backend())
    if ex is not None:
        sizes = [int((u["flat"] if u["flat"] is not None else u["params"][0]).numel() * 4) for u in ex.units]
        out.update(allreduce_bytes=sum(sizes), allreduce_units=len(sizes),
                   collective_backend=("asd_allreduce_mean_f32 (library communicator, RCCL)" if getattr(ex, "_own", None) is not None
                                       else out["collective_backend"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of >= 5 s at ~14 ms per step, so that a sampler outside this process (rocm-smi, the driver's gpu_busy
    # probe at 5-s intervals) sees the GPU work of the timed steps and not only the weight generation and roofline legs around them
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--backend", default=os.environ.get("ASD_BACKEND", "hip"), choices=["hip", "eager"])
    ap.add_argument("--workload", default="asd_sd_nerf", choices=["asd_sd_nerf", "asd_mv_nerf", "asd_sd_hyper_ingp", "asd_sd_3dconv_net", "asd_mv_triplane"],
                    help="asd_sd_nerf = BASELINE configs[1] (the headline metric); asd_mv_nerf = SURVEY C3 (MVDream, 4 views), secondary")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="same-box A/B runs (tools/r6_ab_env.sh): the step line only, no roofline legs")
    ap.add_argument("--render", type=int, default=0, help="secondary workloads only: render size override (e.g. 256 for BASELINE's wording of configs[4])")
    ap.add_argument("--phases", action="store_true", help="also report per-phase milliseconds (adds syncs; untimed extra steps)")
    args = ap.parse_args()

    from scaledreamer_amd import dist as asd_dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("ASD_BENCH_STUB") == "1":
        return main_stub(args, rank, world)
    torch.cuda.set_device(local_rank)
    # the host side of a step is ~60 tiny CPU tensor ops (camera sampling): with torch's default of one OpenMP
    # thread per core (256 on the GPU box) every one of them pays a fork/join; keep the host path single-threaded
    torch.set_num_threads(1)
    if world > 1:
        asd_dist.init_from_env("nccl")
    dev = torch.device("cuda", local_rank)

    # per-rank seed = cfg.seed + rank (launch.py:171): different cameras / noise / t per rank
    cfg, system, data = build_system(args.backend, seed=10 + rank, workload=args.workload)
    if args.render and args.workload != "asd_sd_nerf":
        data.width = data.height = args.render
    asd_dist.broadcast_parameters(system)  # identical initial parameters (DDP wrap-time broadcast)

    # The camera batch of step i+1 is sampled (host) and uploaded while the GPU still executes step i — the prefetch a
    # DataLoader gives the reference; nothing of the step itself moves out of the timed region.
    state = {"batch": to_device(data.collate(), dev)}

    def step():
        batch = state["batch"]
        loss = system.train_one_step(batch)
        state["batch"] = to_device(data.collate(), dev)
        return loss, batch

    for _ in range(args.warmup):
        step()
    trace_mark()                                      # outside the timed region: opens it in a kernel trace (tools/db_steps.py)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # one record per step: no sync, ~1 us each
    marks[0].record()
    for i in range(args.steps):
        loss, batch = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trace_mark()                                      # ... and closes it
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: round(step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))], 3)
    ex = system.gradient_exchange() if hasattr(system, "gradient_exchange") else None
    allreduce_ms = round(ex.exposed_ms(), 3) if ex is not None else None
    evidence = dist_evidence(world, rank, local_rank, dev, ex)
    phases = None
    if args.phases and rank == 0:
        # untimed extra steps with events around the phases of train_one_step (adds host syncs: not part of `value`)
        def ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        acc = {}
        for _ in range(5):
            b = to_device(data.collate(), dev)
            e0 = ev(); system.on_train_batch_start()
            if ex is None:
                system.optimizer.zero_grad(set_to_none=True)
            else:
                ex.prepare()
            e1 = ev(); out_r = system(b)
            e2 = ev(); g_out = system.guidance(out_r["comp_rgb"], system.prompt_utils, **b)
            loss_p = g_out["loss_asd"] + 30.0 * (out_r["opacity"] ** 2 + 0.01).sqrt().mean()
            e3 = ev(); loss_p.backward()
            e4 = ev()
            if ex is not None:
                ex.finish()
            system.optimizer.step(); system.true_global_step += 1
            e5 = ev(); torch.cuda.synchronize()
            for k, (a, b_) in {"update_hooks": (e0, e1), "render_fwd": (e1, e2), "vae_fwd+unet_x5+asd": (e2, e3),
                               "backward(vae+render)": (e3, e4), "allreduce+adamw": (e4, e5)}.items():
                acc[k] = acc.get(k, 0.0) + a.elapsed_time(b_) / 5
        phases = {k: round(v, 3) for k, v in acc.items()}

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
                       "diffusion_backend": args.backend, "diffusion_weights": getattr(getattr(system.guidance, "backend", None), "weights_source", "seeded random init"),
                       "exact_restructurings": "UNet upsampling convs as four 2x2 parity convs (4/9 of their multiply-adds); layers in front of the first "
                                               "cross-attention computed once per distinct (x, t) of the batch of 5 (2 distinct) — same eps within fp16 rounding, "
                                               "tests/test_gpu_unet_engine.py, tests/test_gpu_diffusion_ops.py"},
            "step_ms_gpu": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9)},
            "allreduce_exposed_ms": allreduce_ms, **evidence,
            "loss": float(loss.item()), "kept_samples_last_step": int(system.renderer.last_n_samples) if hasattr(system.renderer, "last_n_samples") else None,
        }
        if args.workload == "asd_mv_nerf":  # secondary line (not BASELINE's metric): 4 views / step / GPU
            out.update({"metric": "ASD train steps/sec (4 views x 64x64 render, MVDream)", "rays_per_sec": round(steps_per_s * 16384, 1),
                        "dtype": "f32 renderer / f16 diffusion with f32 accumulate — the reference runs this prior in fp32 (mvdream_asd_guidance.py:40,67); "
                                 "eps deviation < 1e-2 at full width, batch 12: tests/test_gpu_unet_engine.py::test_full_mvdream_unet_b12_matches_reference_golden"})
            out["config"].update({"workload": "asd_mv_nerf: 4 views/GPU, 4x64x64 rays, 256 spp occgrid march, implicit-volume iNGP, MVDream "
                                              "UNet batch 12 @32x32 latents (CFG + shifted t, cross-view attention), VAE 4x256^2 fwd+bwd, AdamW",
                                  "views_per_gpu": 4})
        if args.workload in ("asd_sd_3dconv_net", "asd_mv_triplane"):
            out.update({"metric": f"ASD train steps/sec ({args.workload}, amortized)"})
            out["config"].update({"workload": {"asd_sd_3dconv_net": "asd_sd_3dconv_net: StyleGAN-3D generator on the HIP path (split-fp16 3x3x3 convolutions fwd / dgrad / wgrad, "
                                                                       "csrc/conv3d.hip) -> [32,128^3] volume, fused trilinear lookup + MLP heads, VolSDF renderer, SD-2.1 guidance, "
                                                                       "1 prompt+view/GPU",
                                               "asd_mv_triplane": "asd_mv_triplane_transformer: 12-layer triplane transformer on the HIP path (csrc/tritx.hip: split-fp16 linears + "
                                                                  "flash attention at fp32-class accuracy) -> 3x[32,64,64] planes, fused tri-plane field on the matrix pipe (lookups + both MLP heads + finite-difference normal, "
                                                                  "csrc/trifield_mfma.hip; sorted plane scatter), VolSDF renderer, MVDream guidance, 4 views/GPU, Adan"}[args.workload]})
            out.pop("kept_samples_last_step", None)
            if args.render:
                out["config"]["render"] = f"{args.render}x{args.render} (override; the shipped YAML renders 64x64)"
                out["rays_per_sec"] = round(steps_per_s * args.render * args.render * (4 if args.workload == "asd_mv_triplane" else 1), 1)
        if args.workload == "asd_sd_hyper_ingp":  # secondary line: amortized multi-prompt training
            out.update({"metric": "ASD train steps/sec (64x64 render, SD2.1, Hyper-iNGP amortized)"})
            out["config"].update({"workload": "asd_sd_hyper_iNGP: 1 prompt+view/GPU, 64x64 rays, importance-sampled VolSDF renderer (128 proposal + "
                                              "193 shading intervals per ray, finite-difference sdf_grad), hypernetwork MLP weights, SD-2.1 "
                                              "UNet batch 5, VAE 512^2 fwd+bwd, Adam"})
            out.pop("kept_samples_last_step", None)
        if phases:
            out["phases_ms"] = phases
        # `roofline` = the kernel rocprofv3 ranks first for this step (profiles/: see DOMINANT above); the other families follow
        if args.no_roofline:
            print(json.dumps(out), flush=True)
            return
        lines = {"gemm": roofline_gemm_kernel(), "vae_conv": roofline_conv_kernel("vae512"), "unet_conv": roofline_conv_kernel("unet64")}
        if args.workload == "asd_sd_nerf":
            lines["pp_conv"] = roofline_pp_kernel()
            lines["gemm256"] = roofline_gemm256_kernel()
        if args.workload in ("asd_sd_nerf", "asd_mv_nerf"):
            lines["field_bwd"] = roofline_field_bwd(system, batch, reps=5)
            lines["renderer"] = roofline_field_kernel(system, batch)
        if args.workload == "asd_sd_3dconv_net":
            lines["conv3d"] = roofline_conv3d()
        dom = DOMINANT if DOMINANT in lines and lines[DOMINANT] is not None else ("conv3d" if "conv3d" in lines else "gemm")
        out["roofline"] = lines.pop(dom)
        out["roofline"]["rank_source"] = (DOMINANT_SOURCE if dom == DOMINANT else
                                          "profiles/r04_c4_3dconv_step_breakdown.txt (generator unchanged in round 5): its convolution kernels lead this workload's step" if dom == "conv3d"
                                          else "secondary workload: most frequent GEMM of the diffusion prior")
        for k, v in lines.items():
            out["roofline_" + k] = v
        if world == 1 and not args.no_cpu_baseline and args.workload == "asd_sd_nerf":
            out["cpu_baseline"] = cpu_baseline(system, batch, seed=10)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        asd_dist.shutdown()
    elif ex is not None:
        ex.close()


def main_stub(args, rank, world):
    """the timing / reporting skeleton of main() on CPU tensors (see build_stub_system): same barrier order, same max-over-ranks
    reduction, same single JSON line from rank 0"""
    from scaledreamer_amd import dist as asd_dist

    torch.set_num_threads(1)
    if world > 1:
        asd_dist.init_from_env("gloo")
    cfg, system, data = build_stub_system(seed=10 + rank)
    asd_dist.broadcast_parameters(system)
    for _ in range(args.warmup):
        system.train_one_step(data.collate())
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = system.train_one_step(data.collate())
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    ex = system.gradient_exchange()
    evidence = dist_evidence(world, rank, int(os.environ.get("LOCAL_RANK", "0")), torch.device("cpu"), ex)
    digest = torch.cat([p.detach().reshape(-1)[:64] for p in system.renderer.parameters()]).double().sum()
    if world > 1:       # replicas must have stayed identical
        lo, hi = digest.clone(), digest.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert float(hi - lo) <= 1e-9 * max(1.0, abs(float(hi))), "replicas diverged"
    if rank == 0:
        print(json.dumps({"metric": "STUB (CPU plumbing test of the multi-process contract, not a measurement)", "value": round(world * args.steps / dt, 4),
                          "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
                          "config": {"workload": "stub", "parallelism": f"dp{world}"},
                          "allreduce_exposed_ms": None if ex is None else round(ex.exposed_ms(), 3), **evidence,
                          "exchange_units": None if ex is None else len(ex.units), "exchange_steps": None if ex is None else ex.prepare_called,
                          "loss": float(loss.item())}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        asd_dist.shutdown()


if __name__ == "__main__":
    main()
