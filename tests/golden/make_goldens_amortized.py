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


def _seed_params(module, tag, seed, scale=1.0):
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(seeded(f"{tag}.{k}", tuple(p.shape), seed, scale))


def make_generator_goldens(seed=12):
    """the reference's own generator backbones (in-tree torch code): StyleGAN-3D `Generator` at 16^3 and `TriplaneTransformer`
    (its diffusers `Attention` dependency is un-vendored: the harness injects scaledreamer_amd.generators.Attention there)."""
    from scaledreamer_amd import generators as G
    H._mod("diffusers")
    H._mod("diffusers.models")
    H._mod("diffusers.models.attention_processor", Attention=G.Attention)
    from custom.amortized.extern.stylegan_3dconv_modules import Generator
    from custom.amortized.extern.triplane_transformer_modules import TriplaneTransformer

    kw = dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16, img_channels=32, channel_multiplier=1)
    gen = Generator(**kw)
    _seed_params(gen, "gen3d", seed, 0.3)
    with torch.no_grad():
        for m in gen.modules():
            if hasattr(m, "noise_const"):
                m.noise_const.copy_(seeded("gen3d.noise." + str(tuple(m.noise_const.shape)), tuple(m.noise_const.shape), seed))
    z, c = seeded("gen3d.z", (2, 64), seed), seeded("gen3d.c", (2, 1024), seed)
    img = gen(z, c, noise_mode="const")["image"]
    g = seeded("gen3d.g", tuple(img.shape), seed)
    (img * g).sum().backward()
    grads = {k: p.grad for k, p in gen.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "amortized_generator3d_16.npz"), seed=seed, keys=np.array(list(gen.state_dict().keys())),
                        image_sub=img.detach()[:, ::4, ::2, ::2, ::2].numpy(), image_l2=np.float64(img.double().norm().item()),
                        g_affine=grads["synthesis.blocks.1.conv0.affine.weight"].numpy(),
                        g_conv_l2=np.float64(grads["synthesis.blocks.0.conv1.weight"].double().norm().item()),
                        g_embed_l2=np.float64(grads["mapping.embed.weight"].double().norm().item()),
                        g_const_l2=np.float64(grads["synthesis.first_block.const"].double().norm().item()))
    for local in (True, False):
        tkw = dict(inner_dim=64, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2, num_heads=4,
                   local_text=local, mlp_ratio=4)
        tt = TriplaneTransformer(**tkw)
        _seed_params(tt, f"tri{int(local)}", seed, 0.2)
        te = seeded("tri.text", (2, 77, 128) if local else (2, 128), seed)
        planes = tt(te)
        gp = seeded("tri.g", tuple(planes.shape), seed)
        (planes * gp).sum().backward()
        tg = {k: p.grad for k, p in tt.named_parameters()}
        np.savez_compressed(os.path.join(HERE, f"amortized_triplane_transformer_local{int(local)}.npz"), seed=seed,
                            keys=np.array(list(tt.state_dict().keys())), planes=planes.detach().numpy(),
                            g_pos_embed=tg["pos_embed"].numpy(), g_deconv=tg["deconv.weight"].numpy(),
                            g_q_l2=np.float64(tg["layers.0.self_attn.to_q.weight"].double().norm().item()))
    print("generator goldens written")


TRI_HD48 = dict(inner_dim=192, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2, num_heads=4, local_text=True,
                mlp_ratio=4)


def make_tritx_hd48_golden(seed=21):
    """the reference TriplaneTransformer at the shipped configuration's HEAD DIMENSION (48 = 768 / 16; asd_mv_triplane_transformer_10k.yaml:44-50)
    on a reduced width — the shape family the HIP generator (csrc/tritx.hip) is built for: planes and the gradient of EVERY parameter,
    batch of two prompts with 77 text tokens each."""
    from scaledreamer_amd import generators as G
    H._mod("diffusers")
    H._mod("diffusers.models")
    H._mod("diffusers.models.attention_processor", Attention=G.Attention)
    from custom.amortized.extern.triplane_transformer_modules import TriplaneTransformer

    tt = TriplaneTransformer(**TRI_HD48).double()
    with torch.no_grad():
        for k, p in tt.named_parameters():
            scale = 1.0 if k.endswith(("norm1.weight", "norm2.weight", "norm3.weight")) or k == "norm.weight" else 0.2
            p.copy_(seeded(f"tri48.{k}", tuple(p.shape), seed, scale).double())
    te = seeded("tri48.text", (2, 77, 128), seed).double()
    planes = tt(te)
    gp = seeded("tri48.g", tuple(planes.shape), seed).double()
    (planes * gp).sum().backward()
    save = {"seed": seed, "keys": np.array(list(tt.state_dict().keys())), "planes": planes.detach().float().numpy()}
    for k, p in tt.named_parameters():
        save["g." + k] = p.grad.float().numpy()
    path = os.path.join(HERE, "amortized_triplane_transformer_hd48.npz")
    np.savez_compressed(path, **save)
    print(f"tritx hd48 golden: planes {tuple(planes.shape)} |planes| max {planes.abs().max().item():.3f} ({os.path.getsize(path) / 1e6:.2f} MB)")


GEN3D_SMALL = dict(z_dim=64, w_dim=256, c_dim=1024, num_layers=2, img_resolution=16, img_channels=32, channel_multiplier=1)
TRI_SMALL = dict(inner_dim=64, condition_dim=128, triplane_low_res=8, triplane_high_res=16, triplane_dim=32, num_layers=2, num_heads=4,
                 local_text=True, mlp_ratio=4)


def make_sampled_geometry_goldens(seed=14):
    """Voxel_3d_Sdf.forward / TriplaneTransformerSDF.forward of the reference on a given space cache: contraction, grid_sample
    lookups, MLP heads, sphere bias, finite-difference sdf_grad / normal, and the gradients w.r.t. the cache and the heads."""
    from scaledreamer_amd import generators as G
    H._mod("diffusers")
    H._mod("diffusers.models")
    H._mod("diffusers.models.attention_processor", Attention=G.Attention)
    from custom.amortized.models.geometry.stylegan_3dconv_net import Voxel_3d_Sdf
    from custom.amortized.models.geometry.triplane_transformer import TriplaneTransformerSDF

    common = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere", "sdf_bias_params": 0.8}
    cases = [("voxel", Voxel_3d_Sdf, dict(common, space_generator_config=GEN3D_SMALL), (1, 32, 16, 16, 16)),
             ("triplane", TriplaneTransformerSDF, dict(common, space_generator_config=TRI_SMALL), (2, 3, 32, 16, 16))]
    for name, cls, cfg, cache_shape in cases:
        geo = cls(cfg)
        geo.update_step(0, 0)
        heads = [("sdf_network", geo.sdf_network), ("feature_network", geo.feature_network)]
        for tag, net in heads:
            _seed_params(net, f"{name}.{tag}", seed, 0.25)
        cache = seeded(f"{name}.cache", cache_shape, seed).requires_grad_(True)
        B = cache_shape[0]
        pts = (torch.rand(B, 700, 3, generator=torch.Generator().manual_seed(seed)) * 4.4 - 2.2)   # some outside the radius-2 box
        out = geo(pts, cache, output_normal=True)
        gs = {k: seeded(f"{name}.g_{k}", tuple(out[k].shape), seed) for k in ("sdf", "features", "normal", "sdf_grad")}
        sum((out[k] * gs[k]).sum() for k in gs).backward()
        save = dict(seed=seed, points=pts.numpy(), d_cache_sub=cache.grad.reshape(-1)[::5].numpy().copy(),
                    d_cache_l2=np.float64(cache.grad.double().norm().item()))
        for k in gs:
            save["out_" + k] = out[k].detach().numpy()
        for tag, net in heads:
            for k, p_ in net.named_parameters():
                save[f"g_{tag}.{k}"] = p_.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"amortized_geometry_{name}.npz"), **save)
    print("sampled-geometry goldens written")


def make_adan_golden(seed=5):
    """threestudio/systems/optimizers.py Adan (pure torch, in-tree): 4 steps on two parameter groups, with and without clipping."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_optimizers", os.path.join(H.REFERENCE, "threestudio", "systems", "optimizers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    save = {}
    for tag, kw in (("plain", {}), ("clip_wd", dict(max_grad_norm=0.5, weight_decay=0.02)), ("noprox", dict(weight_decay=0.02, no_prox=True))):
        p1 = torch.nn.Parameter(seeded("adan.p1", (7, 5), seed))
        p2 = torch.nn.Parameter(seeded("adan.p2", (11,), seed))
        opt = mod.Adan([{"params": [p1], "lr": 0.01}, {"params": [p2], "lr": 0.003}], betas=(0.98, 0.92, 0.99), eps=1e-15, **kw)
        for step in range(4):
            p1.grad = seeded(f"adan.g1.{step}", (7, 5), seed)
            p2.grad = seeded(f"adan.g2.{step}", (11,), seed)
            opt.step()
        save[f"{tag}.p1"], save[f"{tag}.p2"] = p1.detach().numpy().copy(), p2.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "adan_steps.npz"), seed=seed, **save)
    print("adan golden written")


if __name__ == "__main__":
    if "--tritx" in sys.argv:
        make_tritx_hd48_golden()
        sys.exit(0)
    if "--generators" in sys.argv:
        make_generator_goldens()
        make_sampled_geometry_goldens()
        make_adan_golden()
        sys.exit(0)
    make_samplers_golden()
    make_importance_golden()
    make_hyper_renderer_golden()
