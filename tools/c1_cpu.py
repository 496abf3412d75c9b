"""BASELINE.json configs[0] on the host cores: single prompt, 32x32 rays x 16 samples per ray, NeRF-only render (no diffusion) —
forward + backward of the CPU oracle's renderer (oracle/ref_renderer.py, C/OpenMP).  Plumbing case, no GPU.
   python tools/c1_cpu.py [steps]      -> steps/s, rays/s, cores"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from oracle import ref_renderer as R

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
m, mb = O.grid_meta(), O.grid_meta(4, 2, 19, 4, 4.0)
H = W = 32
# one camera on the reference's distribution (uncond.py: distance U[1,1.5], fovy U[40,70], elevation / azimuth drawn), looking at the origin
elev, azim, dist, fovy = np.deg2rad(15.0), np.deg2rad(40.0), 1.3, np.deg2rad(55.0)
pos = dist * np.array([np.cos(elev) * np.cos(azim), np.cos(elev) * np.sin(azim), np.sin(elev)])
fwd = -pos / np.linalg.norm(pos)
right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
up = np.cross(right, fwd)
c2w = np.eye(4, dtype=np.float32)
c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, pos
rays_o, rays_d = O.generate_rays(c2w[None], np.array([0.5 * H / np.tan(0.5 * fovy)], np.float32), H, W)
P = dict(h=H, w=W, spp=16, radius=1.0, rays_o=rays_o, rays_d=rays_d, jitter=None, occs=np.full(32 ** 3, 1.0, np.float32),
         binaries=np.ones(32 ** 3, bool), grid=rng.uniform(-1e-4, 1e-4, m.n_params).astype(np.float32),
         w1d=rng.normal(0, 0.2, (64, 32)).astype(np.float32), w2d=rng.normal(0, 0.2, (1, 64)).astype(np.float32),
         w1f=rng.normal(0, 0.2, (64, 32)).astype(np.float32), w2f=rng.normal(0, 0.2, (3, 64)).astype(np.float32),
         bgrid=rng.uniform(-1e-4, 1e-4, mb.n_params).astype(np.float32), bw0=rng.normal(0, 0.3, (16, 8)).astype(np.float32),
         bw1=rng.normal(0, 0.3, (16, 16)).astype(np.float32), bw2=rng.normal(0, 0.3, (3, 16)).astype(np.float32))
d_rgb, d_op = rng.normal(size=(H * W, 3)).astype(np.float32), rng.normal(size=(H * W, 1)).astype(np.float32)
out, ctx = R.forward(P)
R.backward(P, ctx, d_comp_rgb=d_rgb, d_opacity=d_op)
t0 = time.perf_counter()
for _ in range(steps):
    out, ctx = R.forward(P)
    R.backward(P, ctx, d_comp_rgb=d_rgb, d_opacity=d_op)
dt = (time.perf_counter() - t0) / steps
print(f"C1 (32x32 rays x 16 spp, NeRF-only fwd+bwd, CPU oracle): {1 / dt:.2f} steps/s, {H * W / dt:.0f} rays/s, {dt * 1e3:.1f} ms/step, "
      f"{out['weights'].shape[0]} kept samples, cores = {os.cpu_count()} (OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'default')})")
