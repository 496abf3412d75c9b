"""Golden vectors for the amortized (multi-prompt) renderer path, produced by the REFERENCE's own code (build container only):
  custom/amortized/models/geometry/utils.py          get_trilinear_feature, sample_from_planes, contract_to_unisphere_custom
  threestudio/models/renderers/neus_volume_renderer.py  volsdf_density
  threestudio/models/estimators.py                   ImportanceEstimator.sampling   (nerfacc.pdf / volrend calls -> oracle)
  custom/amortized/models/geometry/hyper_iNGP.py     LinearHyperNetwork, Hypernet_Sdf.forward (tinycudann -> oracle)
  custom/amortized/models/background/multiprompt_neural_environment_hashgrid_map_background.py
  custom/amortized/models/renderers/generative_space_volsdf_volume_renderer.py  GenerativeSpaceVolSDFVolumeRenderer._forward
    python tests/golden/make_goldens_amortized.py   ->  tests/golden/amortized_*.npz
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install_amortized()

from make_goldens import camera_rays, grid_params  # noqa: E402

from custom.amortized.models.geometry.utils import contract_to_unisphere_custom, get_trilinear_feature, sample_from_planes  # noqa: E402
from threestudio.models.estimators import ImportanceEstimator  # noqa: E402
from threestudio.models.renderers.neus_volume_renderer import volsdf_density  # noqa: E402


def seeded(name, shape, seed, scale=1.0):
    import zlib
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g) * scale


def make_samplers_golden(seed=3):
    # get_trilinear_feature's final reshape(df, -1).T (utils.py:108) is only meaningful for B = 1 (the reference runs one
    # prompt per GPU with this geometry), so the voxel golden uses B = 1
    vox = seeded("voxel", (1, 8, 5, 7, 9), seed).requires_grad_(True)         # [B, C, D, H, W]
    pts = (torch.rand(2, 80, 3, generator=torch.Generator().manual_seed(seed)) * 2.4 - 1.2)   # some fall outside [-1, 1]
    pts[0, :4] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [0.999, -0.999, 0.5]])
    f = get_trilinear_feature(pts[:1], vox)                                   # [1, M, C]
    g = seeded("g_vox", tuple(f.shape), seed)
    (f * g).sum().backward()
    planes = seeded("planes", (2, 3, 4, 6, 8), seed).requires_grad_(True)     # [B, 3, C, H, W]
    fp = sample_from_planes(planes, pts)                                      # [B, M, 3C]
    gp = seeded("g_pl", tuple(fp.shape), seed)
    (fp * gp).sum().backward()
    sdf = seeded("sdf", (500,), seed, 0.2)
    sdf[:3] = torch.tensor([0.0, 1e-4, -1e-4])
    bbox = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]])
    np.savez_compressed(os.path.join(HERE, "amortized_samplers.npz"), voxel=vox.detach().numpy(), points=pts.numpy(),
                        tri_out=f.detach().numpy(), tri_g=g.numpy(), tri_dvoxel=vox.grad.numpy(),
                        planes=planes.detach().numpy(), plane_out=fp.detach().numpy(), plane_g=gp.numpy(), plane_dplanes=planes.grad.numpy(),
                        sdf=sdf.numpy(), volsdf_30=volsdf_density(sdf, torch.tensor(30.0)).numpy(),
                        volsdf_200=volsdf_density(sdf, torch.tensor(200.0)).numpy(),
                        contract=contract_to_unisphere_custom(pts * 2, bbox, False).numpy())
    print("samplers golden written")


def make_importance_golden(seed=4, n_rays=64, n_prop=32, n_fine=16, near=0.1, far=4.0):
    est = ImportanceEstimator()
    rng = np.random.default_rng(seed)
    centre = torch.from_numpy(rng.uniform(1.0, 3.0, (n_rays, 1)).astype(np.float32))
    width = torch.from_numpy(rng.uniform(0.05, 0.6, (n_rays, 1)).astype(np.float32))
    amp = torch.from_numpy(rng.uniform(0.0, 40.0, (n_rays, 1)).astype(np.float32))
    amp[:3] = 0.0                                                              # empty rays: cdf is flat until the final 1

    def sigma_fn(t0, t1):
        tm = (t0 + t1) / 2
        return amp * torch.exp(-0.5 * ((tm - centre) / width) ** 2)
    for stratified in (True, False):
        H.IMPORTANCE_LOG.clear()
        jit = [rng.uniform(0, 1, n_rays).astype(np.float32) for _ in range(2)]
        H.IMPORTANCE_JITTER[:] = [j.copy() for j in jit]
        t0, t1 = est.sampling(prop_sigma_fns=[sigma_fn], prop_samples=[n_prop], num_samples=n_fine, n_rays=n_rays, near_plane=near,
                              far_plane=far, sampling_type="uniform", stratified=stratified)
        tag = "strat" if stratified else "det"
        save = dict(centre=centre.numpy(), width=width.numpy(), amp=amp.numpy(), near=near, far=far, n_prop=n_prop, n_fine=n_fine,
                    t_starts=t0.numpy(), t_ends=t1.numpy(), jitter0=jit[0], jitter1=jit[1],
                    cdfs=H.IMPORTANCE_LOG[1]["cdfs"], s_prop=H.IMPORTANCE_LOG[0]["out"], s_fine=H.IMPORTANCE_LOG[1]["out"])
        np.savez_compressed(os.path.join(HERE, f"amortized_importance_{tag}.npz"), **save)
    print("importance goldens written", t0.shape)


ENC = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
       "per_level_scale": 1.447269237440378}
BG_ENC = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16,
          "per_level_scale": 1.0}
HYPER = {"c_dim": 1024, "out_dims": {"sdf_weights": [64, 1], "feature_weights": [64, 3]}, "spectral_norm": False, "n_neurons": 64,
         "n_hidden_layers": 1}


def make_hyper_renderer_golden(name="amortized_hyper_ingp_2x4x4", B=2, h=4, w=4, n_prop=128, n_fine=64, seed=31):
    from custom.amortized.models.background.multiprompt_neural_environment_hashgrid_map_background import \
        MultipromptNeuralHashgridEnvironmentMapBackground as BG
    from custom.amortized.models.geometry.hyper_iNGP import Hypernet_Sdf
    from custom.amortized.models.renderers.generative_space_volsdf_volume_renderer import GenerativeSpaceVolSDFVolumeRenderer as R
    from threestudio.models.materials.no_material import NoMaterial

    torch.manual_seed(seed)
    geo = Hypernet_Sdf({"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere",
                        "sdf_bias_params": 0.5, "shape_init": "sphere", "shape_init_params": 0.5, "hypernet_config": HYPER,
                        "pos_encoding_config": ENC})
    mat = NoMaterial({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True})
    bg = BG({"color_activation": "sigmoid", "random_aug": True, "random_aug_prob": 0.2, "pos_encoding_config": BG_ENC})
    ren = R({"radius": 2.0, "use_volsdf": True, "trainable_variance": False, "learned_variance_init": 0.340119, "estimator": "importance",
             "num_samples_per_ray": n_fine, "num_samples_per_ray_importance": n_prop, "near_plane": 0.1, "far_plane": 4.0,
             "train_chunk_size": 0}, geometry=geo, material=mat, background=bg)
    geo.update_step(0, 0)
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.copy_(torch.from_numpy(grid_params(seed, 12_599_920, 0.004)))
        nbg = bg.encoding.encoding.encoding.params.numel()
        bg.encoding.encoding.encoding.params.copy_(torch.from_numpy(grid_params(seed + 1, nbg, 0.5)))
        # hypernetwork weights by the seeded name-keyed rule (regenerated, not stored, by the tests): xavier-like scales
        for tag, net in (("geo_hyper", geo.hypernet), ("bg_hyper", bg.hypernet)):
            for k, p in net.named_parameters():
                if p.ndim == 2:
                    p.copy_(seeded(f"{tag}.{k}", tuple(p.shape), seed, (2.0 / (p.shape[0] + p.shape[1])) ** 0.5))
                elif k.endswith("bias"):
                    p.copy_(seeded(f"{tag}.{k}", tuple(p.shape), seed, 0.02))
                else:
                    p.copy_(1.0 + seeded(f"{tag}.{k}", tuple(p.shape), seed, 0.05))
    text_embed = seeded("text_embed", (B, 1024), seed)
    rays_o, rays_d, pos = [], [], []
    for b, cam in enumerate([(15.0, 30.0, 1.8, 50.0), (40.0, -100.0, 2.2, 45.0)][:B]):
        o, d, p = camera_rays(h, w, *cam)
        rays_o.append(o), rays_d.append(d), pos.append(p)
    rays_o, rays_d, pos = torch.cat(rays_o), torch.cat(rays_d), torch.cat(pos)
    rng = np.random.default_rng(seed + 7)
    jit = [rng.uniform(0, 1, B * h * w).astype(np.float32) for _ in range(2)]
    H.IMPORTANCE_JITTER[:] = [j.copy() for j in jit]
    random.random = lambda: 0.9
    ren.train(); geo.train(); bg.train(); mat.train()
    out = ren(rays_o=rays_o, rays_d=rays_d, light_positions=pos, text_embed=text_embed)
    g_rgb = torch.from_numpy(rng.normal(size=(B, h, w, 3)).astype(np.float32))
    loss_eik = ((torch.linalg.norm(out["sdf_grad"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
    loss = (out["comp_rgb"] * g_rgb).sum() + 20.0 * loss_sparsity + 100.0 * loss_eik
    loss.backward()
    save = dict(B=B, h=h, w=w, n_prop=n_prop, n_fine=n_fine, seed=seed, rays_o=rays_o.numpy(), rays_d=rays_d.numpy(),
                light_positions=pos.numpy(), text_embed=text_embed.numpy(), jitter0=jit[0], jitter1=jit[1], g_rgb=g_rgb.numpy(),
                loss=np.float64(loss.item()), loss_eikonal=np.float64(loss_eik.item()))
    save["hyper_param_names"] = np.array([f"{t}.{k}" for t, n in (("geo_hyper", geo.hypernet), ("bg_hyper", bg.hypernet))
                                          for k, _ in n.named_parameters()])
    for k, v in out.items():
        if k not in ("shading_normal", "t_dirs", "ray_indices"):
            save["out_" + k] = v.detach().numpy()
    for tag, net in (("geo_hyper", geo.hypernet), ("bg_hyper", bg.hypernet)):
        for k, p in net.named_parameters():   # strided subsample + norm of every hypernetwork gradient
            g = p.grad.reshape(-1)
            save[f"g_{tag}.{k}"] = g[::7].numpy().copy()
            save[f"gl2_{tag}.{k}"] = np.float64(g.double().norm().item())
    gg = geo.encoding.encoding.encoding.params.grad.numpy()
    top = np.argsort(-np.abs(gg))[:6000].astype(np.int64)
    save.update(g_grid_idx=top, g_grid_val=gg[top], g_grid_l2=np.float64(np.linalg.norm(gg.astype(np.float64))))
    gb = bg.encoding.encoding.encoding.params.grad.numpy()
    topb = np.argsort(-np.abs(gb))[:3000].astype(np.int64)
    save.update(g_bgrid_idx=topb, g_bgrid_val=gb[topb], g_bgrid_l2=np.float64(np.linalg.norm(gb.astype(np.float64))))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **save)
    print(f"{name}: N={out['weights'].shape[0]} loss={loss.item():.5f} eik={loss_eik.item():.5f} opacity={out['opacity'].mean().item():.4f} "
          f"inv_std={float(out['inv_std']):.3f} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    make_samplers_golden()
    make_importance_golden()
    make_hyper_renderer_golden()
