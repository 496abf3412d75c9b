"""Generate golden vectors by running the REFERENCE's own Python glue in this container.

  python tests/golden/make_goldens.py            # writes tests/golden/*.npz

What runs is the reference's unchanged code imported from /root/reference under ref_harness.py:
NeRFVolumeRenderer.forward, ImplicitVolume, NoMaterial, NeuralEnvironmentMapBackground, VanillaMLP,
get_ray_directions/get_rays, the scaledreamer-system loss terms — on top of the build's CPU oracle injected
as `tinycudann` / `nerfacc` (those two packages are not vendored by the reference).
The .npz files are data only (inputs, seeds / generation rules, expected outputs).
"""
from __future__ import annotations

import math
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install()

from threestudio.models.background.neural_environment_map_background import NeuralEnvironmentMapBackground  # noqa: E402
from threestudio.models.geometry.implicit_volume import ImplicitVolume  # noqa: E402
from threestudio.models.materials.no_material import NoMaterial  # noqa: E402
from threestudio.models.renderers.nerf_volume_renderer import NeRFVolumeRenderer  # noqa: E402
from threestudio.utils.ops import binary_cross_entropy, dot, get_ray_directions, get_rays  # noqa: E402

BG_ENC = {"otype": "HashGrid", "n_features_per_level": 2, "log2_hashmap_size": 19, "n_levels": 4, "base_resolution": 4,
          "per_level_scale": 4.0}


def grid_params(seed: int, n: int, amp: float) -> np.ndarray:
    """Generation rule of the hash-table parameters (restated in tests/golden_util.py)."""
    return np.random.default_rng(seed).uniform(-amp, amp, n).astype(np.float32)


def camera_rays(h, w, elev_deg, azim_deg, dist, fovy_deg):
    """One camera with the reference's conventions (threestudio/data/uncond.py:216-305)."""
    elev, azim, fovy = (math.radians(v) for v in (elev_deg, azim_deg, fovy_deg))
    pos = torch.tensor([[dist * math.cos(elev) * math.cos(azim), dist * math.cos(elev) * math.sin(azim), dist * math.sin(elev)]])
    up = torch.tensor([[0.0, 0.0, 1.0]])
    lookat = torch.nn.functional.normalize(-pos, dim=-1)
    right = torch.nn.functional.normalize(torch.cross(lookat, up, dim=-1), dim=-1)
    up = torch.nn.functional.normalize(torch.cross(right, lookat, dim=-1), dim=-1)
    c2w = torch.cat([torch.stack([right, up, -lookat], dim=-1), pos[:, :, None]], dim=-1)
    c2w = torch.cat([c2w, torch.zeros_like(c2w[:, :1])], dim=1)
    c2w[:, 3, 3] = 1.0
    dirs = get_ray_directions(H=h, W=w, focal=1.0)[None].clone()
    focal = 0.5 * h / math.tan(0.5 * fovy)
    dirs[..., :2] = dirs[..., :2] / focal
    rays_o, rays_d = get_rays(dirs, c2w, keepdim=True, normalize=True)
    return rays_o.contiguous(), rays_d.contiguous(), pos


def build(spp: int, seed: int, grid_amp: float):
    torch.manual_seed(seed)
    geo = ImplicitVolume({"radius": 1.0, "normal_type": "finite_difference"})
    mat = NoMaterial({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True})
    bg = NeuralEnvironmentMapBackground({"color_activation": "sigmoid", "random_aug": True, "random_aug_prob": 0.5,
                                         "dir_encoding_config": BG_ENC})
    ren = NeRFVolumeRenderer({"radius": 1.0, "num_samples_per_ray": spp}, geometry=geo, material=mat, background=bg)
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.copy_(torch.from_numpy(grid_params(seed, 12_599_920, grid_amp)))
        bg.encoding.encoding.encoding.params.copy_(torch.from_numpy(grid_params(seed + 1, 1_581_184, 0.5)))
        for p in list(geo.density_network.parameters()) + list(geo.feature_network.parameters()):
            p.mul_(2.0)  # a little more signal than nn.Linear's default init
    return geo, mat, bg, ren


def occupancy_from_field(geo, ren):
    """The nerfacc warm-up update at step 0 without jitter: occ = sigma(cell centre) * step."""
    res = 32
    ix, iy, iz = torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij")
    x = (torch.stack([ix, iy, iz], -1).reshape(-1, 3).float() + 0.5) / res * 2 - 1
    with torch.no_grad():
        occ = geo.forward_density(x)[..., 0] * ren.render_step_size
    thre = min(float(occ.mean()), 0.01)
    return occ, (occ > thre).view(1, res, res, res)


def make_renderer_golden(name, h, w, spp, seed, grid_amp, cam):
    geo, mat, bg, ren = build(spp, seed, grid_amp)
    occ, binaries = occupancy_from_field(geo, ren)
    ren.estimator.occs.copy_(occ)
    ren.estimator.binaries.copy_(binaries)
    rays_o, rays_d, cam_pos = camera_rays(h, w, *cam)
    rng = np.random.default_rng(seed + 7)
    jitter = rng.uniform(0, 1, h * w).astype(np.float32)
    ren.estimator.jitter = jitter
    random.random = lambda: 0.9  # no random-colour background this step
    ren.train()
    geo.train(); bg.train(); mat.train()
    out = ren(rays_o=rays_o, rays_d=rays_d, light_positions=cam_pos)
    # loss: linear probes on the image-space outputs + the reference's own regularisers
    # (threestudio/systems/scaledreamer.py:69-91: orient, sparsity, opaque)
    g_rgb = torch.from_numpy(rng.normal(size=(1, h, w, 3)).astype(np.float32))
    g_depth = torch.from_numpy(rng.normal(size=(1, h, w, 1)).astype(np.float32))
    loss_probe = (out["comp_rgb"] * g_rgb).sum() + 0.1 * (out["depth"] * g_depth).sum()
    loss_orient = (out["weights"].detach() * dot(out["normal"], out["t_dirs"]).clamp_min(0.0) ** 2).sum() / (out["opacity"] > 0).sum()
    loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
    oc = out["opacity"].clamp(1.0e-3, 1.0 - 1.0e-3)
    loss_opaque = binary_cross_entropy(oc, oc)
    loss_zvar = out["z_variance"][out["opacity"] > 0.5].mean() if (out["opacity"] > 0.5).any() else out["z_variance"].sum() * 0
    loss = loss_probe + 10.0 * loss_orient + 30.0 * loss_sparsity + 5.0 * loss_opaque + 3.0 * loss_zvar
    loss.backward()

    save = dict(
        h=h, w=w, spp=spp, seed=seed, grid_amp=grid_amp, cam=np.array(cam, np.float32),
        rays_o=rays_o.numpy(), rays_d=rays_d.numpy(), light_positions=cam_pos.numpy(), jitter=jitter,
        occs=occ.numpy(), binaries=binaries.numpy(), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
        w1d=geo.density_network.layers[0].weight.detach().numpy(), w2d=geo.density_network.layers[2].weight.detach().numpy(),
        w1f=geo.feature_network.layers[0].weight.detach().numpy(), w2f=geo.feature_network.layers[2].weight.detach().numpy(),
        bw0=bg.network.layers[0].weight.detach().numpy(), bw1=bg.network.layers[2].weight.detach().numpy(),
        bw2=bg.network.layers[4].weight.detach().numpy(),
        loss=np.float64(loss.item()), loss_orient=np.float64(loss_orient.item()), loss_sparsity=np.float64(loss_sparsity.item()),
        loss_opaque=np.float64(loss_opaque.item()), loss_zvar=np.float64(loss_zvar.item()),
    )
    for k, v in out.items():
        save["out_" + k] = v.detach().numpy()
    save["g_w1d"] = geo.density_network.layers[0].weight.grad.numpy()
    save["g_w2d"] = geo.density_network.layers[2].weight.grad.numpy()
    save["g_w1f"] = geo.feature_network.layers[0].weight.grad.numpy()
    save["g_w2f"] = geo.feature_network.layers[2].weight.grad.numpy()
    save["g_bw0"] = bg.network.layers[0].weight.grad.numpy()
    save["g_bw1"] = bg.network.layers[2].weight.grad.numpy()
    save["g_bw2"] = bg.network.layers[4].weight.grad.numpy()
    gg = geo.encoding.encoding.encoding.params.grad.numpy()
    top = np.argsort(-np.abs(gg))[:20000].astype(np.int64)
    save.update(g_grid_idx=top, g_grid_val=gg[top], g_grid_l2=np.float64(np.linalg.norm(gg.astype(np.float64))),
                g_grid_nnz=np.int64((gg != 0).sum()))
    gb = bg.encoding.encoding.encoding.params.grad.numpy()
    topb = np.argsort(-np.abs(gb))[:5000].astype(np.int64)
    save.update(g_bgrid_idx=topb, g_bgrid_val=gb[topb], g_bgrid_l2=np.float64(np.linalg.norm(gb.astype(np.float64))))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **save)
    n = out["weights"].shape[0]
    print(f"{name}: N={n} samples, loss={loss.item():.6f}, opacity mean={out['opacity'].mean().item():.4f} -> {path} "
          f"({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    # BASELINE config 1: single prompt, 32x32 rays, 16 samples per ray, NeRF only
    make_renderer_golden("renderer_c1_32x32x16", 32, 32, 16, seed=11, grid_amp=0.2, cam=(15.0, 30.0, 1.25, 55.0))
    # a C2-shaped miniature: 512 samples per ray (step 0.006766), 12x12 rays
    make_renderer_golden("renderer_c2mini_12x12x512", 12, 12, 512, seed=23, grid_amp=0.1, cam=(35.0, -110.0, 1.1, 45.0))
