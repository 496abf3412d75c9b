"""A/B of libasd_hip.so builds on field_bwd_sample_kernel alone (tools/build_variant.sh ... field.hip; ASD_HIP_LIB selects the build).
   python tools/field_bwd_ab.py dump   -> 12 training steps of the headline config, the live samples + field parameters to /tmp/fb.pt
   python tools/field_bwd_ab.py time   -> geometry module only, the dumped samples: asd_field_bwd's scatter kernel through the probe events
The geometry's gradient table after one call is checksummed so that builds can be compared for equal results (up to atomic order)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from scaledreamer_amd import ops, presets
from scaledreamer_amd._lib import lib

dev = torch.device("cuda", 0)
if sys.argv[1] == "dump":
    cfg, system, data = bench.build_system("hip", seed=10, workload="asd_sd_nerf")
    for _ in range(25):
        system.train_one_step(bench.to_device(data.collate(), dev))
    batch = bench.to_device(data.collate(), dev)
    ren, geo = system.renderer, system.geometry
    with torch.no_grad():
        ri, t0, t1, pts, dirs, off, cnt, _ = ren._sample(batch["rays_o"].reshape(-1, 3).contiguous(), batch["rays_d"].reshape(-1, 3).contiguous())
    torch.save({"pts": pts.cpu(), "geo": {k: v.cpu() for k, v in geo.state_dict().items()}}, "/tmp/fb.pt")
    print("dumped", pts.shape[0], "samples")
else:
    from scaledreamer_amd.registry import find
    import scaledreamer_amd.plugins  # noqa: F401
    presets.ALLOW_RANDOM_WEIGHTS = True
    cfg = presets.asd_sd_nerf(guidance_backend="hip")
    geo = find(cfg["system"]["geometry_type"])(cfg["system"]["geometry"]).to(dev)
    d = torch.load("/tmp/fb.pt")
    geo.load_state_dict(d["geo"])
    pts = d["pts"].to(dev)
    n = pts.shape[0]
    grid = geo.encoding.encoding.encoding.params.detach()
    w = [t.detach() for t in geo._weights()]
    with torch.no_grad():
        sigma, feats, normal, enc = ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, False)
        g = torch.Generator(device="cuda").manual_seed(0)
        d_sigma, d_feats = torch.randn(n, device="cuda", generator=g), torch.randn(n, 3, device="cuda", generator=g)
        if os.environ.get("FB_NO_FEAT"):      # ablation: no feature-network gradient (the kernel skips that MLP)
            d_feats = None
        d_grid = torch.zeros_like(grid)
        ops.field_bwd(geo._meta, geo._fcfg, grid, *w, pts, enc, sigma, d_sigma, d_feats, None, d_grid)
        chk = (float(d_grid.double().sum()), float(d_grid.double().abs().sum()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record(); torch.cuda.synchronize()
        lib().asd_probe_events(C.c_void_p(e0.cuda_event), C.c_void_p(e1.cuda_event))
        ms = []
        for _ in range(12):
            d_grid.zero_()
            ops.field_bwd(geo._meta, geo._fcfg, grid, *w, pts, enc, sigma, d_sigma, d_feats, None, d_grid)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        lib().asd_probe_events(None, None)
    def timed(fn, reps=12):
        fn(); fn()
        t = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            t.append(a.elapsed_time(b))
        t.sort()
        return round(t[len(t) // 2] * 1e3, 1)
    with torch.no_grad():
        fwd_n = timed(lambda: ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, True))
        fwd_1 = timed(lambda: ops.field_fwd(geo._meta, geo._fcfg, grid, *w, pts, False))
        dens = timed(lambda: ops.field_density(geo._meta, geo._fcfg, grid, w[0], w[1], pts))
    print("   field_fwd(normal) us", fwd_n, " field_fwd(no normal) us", fwd_1, " field_density us", dens, "(call times incl. launch)")
    ms.sort()
    print(os.path.basename(os.environ.get("ASD_HIP_LIB", "") or "default"), "samples", n, "median_us", round(ms[len(ms) // 2] * 1e3, 1),
          "min_us", round(ms[0] * 1e3, 1), "checksum", chk)
