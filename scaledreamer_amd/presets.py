"""Experiment presets: the values of the reference's shipped YAML files as plain dicts (the GPU box has no
/root/reference; `config.load_config` reads the YAML files themselves when a ScaleDreamer checkout exists).
"""
from __future__ import annotations

import contextlib
import copy


# No pretrained checkpoint exists offline.  A preset-driven run therefore FAILS when its checkpoint path does not resolve (a prior of
# random weights distills nothing); bench.py, __graft_entry__.smoke(), tests/conftest.py and the tools set this switch explicitly to
# run on the seeded random prior (diffusion/checkpoint.py: resolve_params).
ALLOW_RANDOM_WEIGHTS = False


@contextlib.contextmanager
def random_weights_allowed(allow: bool = True):
    """scoped form of the switch: presets built inside the block accept a missing checkpoint, anything built after it fails closed again"""
    global ALLOW_RANDOM_WEIGHTS
    saved, ALLOW_RANDOM_WEIGHTS = ALLOW_RANDOM_WEIGHTS, allow
    try:
        yield
    finally:
        ALLOW_RANDOM_WEIGHTS = saved


def asd_sd_nerf(prompt: str = "synthetic", guidance_backend: str = "hip") -> dict:
    """configs/single-prompt_benchmark/asd_sd_nerf.yaml (BASELINE config 2: 64x64 render, SD-2.1 guidance,
    implicit-volume iNGP geometry, occupancy-grid renderer, Perp-Neg, AdamW with 5 parameter groups)."""
    return copy.deepcopy({
        "name": "asd_sd_nerf", "seed": 10,
        "data_type": "random-camera-datamodule",
        "data": {"batch_size": [1, 1], "width": [64, 256], "height": [64, 256], "resolution_milestones": [10000],
                 "camera_distance_range": [1.0, 1.5], "fovy_range": [40, 70], "elevation_range": [-10, 45],
                 "camera_perturb": 0.0, "center_perturb": 0.0, "up_perturb": 0.0, "eval_camera_distance": 1.2,
                 "eval_fovy_deg": 70.0, "n_val_views": 30},
        "system_type": "scaledreamer-system",
        "system": {
            "visualize_samples": False, "validation_via_video": True,
            "geometry_type": "implicit-volume",
            "geometry": {"radius": 1.0, "normal_type": "finite_difference", "density_bias": "blob_magic3d",
                         "density_activation": "softplus", "density_blob_scale": 10.0, "density_blob_std": 0.5,
                         "pos_encoding_config": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                                                 "log2_hashmap_size": 19, "base_resolution": 16,
                                                 "per_level_scale": 1.447269237440378}},
            "material_type": "no-material",
            "material": {"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True},
            "background_type": "neural-environment-map-background",
            "background": {"color_activation": "sigmoid", "random_aug": True, "random_aug_prob": 0.5,
                           "dir_encoding_config": {"otype": "HashGrid", "n_features_per_level": 2, "log2_hashmap_size": 19,
                                                   "n_levels": 4, "base_resolution": 4, "per_level_scale": 4.0}},
            "renderer_type": "nerf-volume-renderer",
            "renderer": {"radius": 1.0, "num_samples_per_ray": 512},
            "prompt_processor_type": "stable-diffusion-prompt-processor",
            "prompt_processor": {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "prompt": prompt,
                                 "use_perp_neg": True, "front_threshold": 30.0, "back_threshold": 30.0},
            "guidance_type": "stable-diffusion-asynchronous-score-distillation-guidance",
            "guidance": {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "guidance_scale": 7.5,
                         "plus_ratio": 0.1, "plus_random": True, "min_step_percent": [0, 0.5, 0.02, 25000],
                         "max_step_percent": [0, 0.98, 0.5, 25000], "guidance_perp_neg": -0.5,
                         "backend": guidance_backend, "allow_random_weights": ALLOW_RANDOM_WEIGHTS},
            "loggers": {"wandb": {"enable": False, "project": "threestudio", "name": "None"}},
            "loss": {"lambda_asd": 1.0, "lambda_orient": 0.0, "lambda_sparsity": 30, "lambda_opaque": [10000, 0.0, 100.0, 10001],
                     "lambda_z_variance": 0.0},
            "optimizer": {"name": "AdamW", "args": {"betas": [0.0, 0.99], "eps": 1.0e-15},
                          "params": {"geometry.encoding": {"lr": 0.01}, "geometry.density_network": {"lr": 0.001},
                                     "geometry.feature_network": {"lr": 0.001}, "background.encoding": {"lr": 0.01},
                                     "background.network": {"lr": 0.001}}},
        },
        "trainer": {"max_steps": 25000, "precision": 32},
    })


def asd_mv_nerf(prompt: str = "synthetic", guidance_backend: str = "hip-mvdream") -> dict:
    """configs/single-prompt_benchmark/asd_mv_nerf.yaml: 4 views of one prompt, 256 spp, MVDream guidance
    (shared t, batch 12 at 32x32 latents, no Perp-Neg), one optimizer group for the whole background."""
    cfg = asd_sd_nerf(prompt)
    cfg["name"] = "asd_mv_nerf"
    cfg["data_type"] = "mvdream-random-multiview-camera-datamodule"
    cfg["data"] = {"batch_size": [4, 4], "n_view": 4, "width": [64, 256], "height": [64, 256], "resolution_milestones": [10000],
                   "camera_distance_range": [0.8, 1.0], "fovy_range": [15, 60], "elevation_range": [0, 30], "camera_perturb": 0.0,
                   "center_perturb": 0.0, "up_perturb": 0.0, "eval_camera_distance": 3.0, "eval_fovy_deg": 40.0, "n_val_views": 30}
    s = cfg["system"]
    s["renderer"]["num_samples_per_ray"] = 256
    s["prompt_processor"] = {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "prompt": prompt,
                             "front_threshold": 30.0, "back_threshold": 30.0}
    s["guidance_type"] = "mvdream-asynchronous-score-distillation-guidance"
    s["guidance"] = {"model_name": "sd-v2.1-base-4view", "ckpt_path": "pretrained/sd-v2.1-base-4view.pt", "guidance_scale": 7.5,
                     "plus_ratio": 0.1, "plus_random": True, "min_step_percent": [0, 0.5, 0.02, 25000],
                     "max_step_percent": [0, 0.98, 0.5, 25000], "backend": guidance_backend, "allow_random_weights": ALLOW_RANDOM_WEIGHTS}
    s["loss"] = {"lambda_asd": 1.0, "lambda_orient": [10000, 0.0, 100.0, 10001], "lambda_sparsity": 20,
                 "lambda_opaque": [10000, 0.0, 100.0, 10001], "lambda_z_variance": 0.0}
    s["optimizer"]["params"] = {"geometry.encoding": {"lr": 0.003}, "geometry.density_network": {"lr": 0.003},
                                "geometry.feature_network": {"lr": 0.003}, "background": {"lr": 0.003}}
    return cfg


def asd_sd_hyper_ingp(prompts=None, guidance_backend: str = "hip") -> dict:
    """configs/multi-prompt_benchmark/asd_sd_hyper_iNGP_50k.yaml: multi-prompt amortized training — Hyper-iNGP SDF field
    (per-prompt MLP weights from a hypernetwork), hypernetwork hash-grid background, importance-sampled VolSDF renderer
    (128 proposal + 64 resampled edges -> 193 intervals per ray), SD-2.1 ASD guidance, Adam."""
    prompts = prompts or [f"synthetic prompt {i}" for i in range(16)]
    enc = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16}
    return copy.deepcopy({
        "name": "asd_sd_hyper_iNGP_50k", "seed": 0,
        "data_type": "multiprompt-camera-datamodule",
        "data": {"batch_size": 1, "width": 64, "height": 64, "camera_distance_range": [1.0, 1.5], "fovy_range": [40, 70],
                 "elevation_range": [-10, 45], "camera_perturb": 0.0, "center_perturb": 0.0, "up_perturb": 0.0,
                 "eval_camera_distance": 1.5, "eval_fovy_deg": 70.0, "n_val_views": 30, "dim_gaussian": 0,
                 "prompt_library": {"train": prompts, "val": prompts[:1], "test": prompts[:1]}},
        "system_type": "multiprompt-radience-field-generator-system",
        "system": {
            "stage": "coarse", "initialize_shape": False, "visualize_samples": False, "validation_via_video": True,
            "geometry_type": "Hyper-iNGP",
            "geometry": {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere",
                         "sdf_bias_params": 0.5, "shape_init": "sphere", "shape_init_params": 0.5,
                         "hypernet_config": {"c_dim": 1024, "out_dims": {"sdf_weights": [64, 1], "feature_weights": [64, 3]},
                                             "spectral_norm": False, "n_neurons": 64, "n_hidden_layers": 1}},
            "material_type": "no-material",
            "material": {"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True},
            "background_type": "multiprompt-neural-hashgrid-environment-map-background",
            "background": {"color_activation": "sigmoid", "random_aug": True, "random_aug_prob": 0.2,
                           "pos_encoding_config": dict(enc, per_level_scale=1.0)},
            "renderer_type": "generative-space-volsdf-volume-renderer",
            "renderer": {"radius": 2.0, "use_volsdf": True, "trainable_variance": False, "learned_variance_init": 0.340119,
                         "estimator": "importance", "num_samples_per_ray": 64, "num_samples_per_ray_importance": 128,
                         "near_plane": 0.1, "far_plane": 4.0, "train_chunk_size": 0},
            "prompt_processor_type": "stable-diffusion-multi-prompt-processor",
            "prompt_processor": {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "use_perp_neg": True,
                                 "front_threshold": 30.0, "back_threshold": 30.0},
            "guidance_type": "stable-diffusion-asynchronous-score-distillation-guidance",
            "guidance": {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "guidance_scale": 7.5,
                         "plus_ratio": 0.1, "plus_random": True, "min_step_percent": [0, 0.5, 0.02, 50000],
                         "max_step_percent": [0, 0.98, 0.5, 50000], "guidance_perp_neg": -0.5, "backend": guidance_backend, "allow_random_weights": ALLOW_RANDOM_WEIGHTS},
            "loggers": {"wandb": {"enable": False, "project": "threestudio", "name": "None"}},
            "loss": {"lambda_asd": 1.0, "lambda_orient": 0.0, "lambda_sparsity": 20, "lambda_opaque": [40000, 0, 10.0, 50000],
                     "lambda_z_variance": 0.0, "lambda_eikonal": [1, 100.0, 1.0, 5000]},
            "optimizer": {"name": "Adam", "args": {"betas": [0.0, 0.99], "eps": 1.0e-8},
                          "params": {"geometry": {"lr": 0.004}, "background": {"lr": 0.001}}},
        },
        "trainer": {"max_steps": 50000, "precision": 32},
    })


def asd_sd_3dconv_net(prompts=None, guidance_backend: str = "hip") -> dict:
    """configs/multi-prompt_benchmark/asd_sd_3dconv_net_*.yaml: the StyleGAN-3D generator emits a [32, 128^3] feature
    volume per prompt, sampled trilinearly; otherwise the amortized renderer / guidance of asd_sd_hyper_ingp."""
    cfg = asd_sd_hyper_ingp(prompts, guidance_backend)
    cfg["name"] = "asd_sd_3dconv_net"
    cfg["data"]["dim_gaussian"] = 64
    s = cfg["system"]
    s["geometry_type"] = "3DConv-net"
    s["geometry"] = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "activation": "none",
                     "sdf_bias": "sphere", "sdf_bias_params": 0.8,
                     "space_generator_config": {"z_dim": 64, "w_dim": 256, "c_dim": 1024, "num_layers": 2, "img_resolution": 128,
                                                "img_channels": 32, "channel_multiplier": 1}}
    s["background_type"] = "neural-environment-map-background"
    s["background"] = {"color_activation": "sigmoid", "random_aug": True}
    s["optimizer"] = {"name": "Adam", "args": {"betas": [0.0, 0.99], "eps": 1.0e-8},
                      "params": {"geometry": {"lr": 0.004}, "background": {"lr": 0.001}}}
    return cfg


def asd_mv_triplane_transformer(prompts=None, n_gpus: int = 8) -> dict:
    """configs/multi-prompt_benchmark/asd_mv_triplane_transformer_10k.yaml: text-conditioned transformer -> three
    [32, 64, 64] planes per prompt, 4 views per prompt, MVDream guidance, Adan; gradients accumulated over 2 batches on the 8-GPU node
    (:129) and over 8 in the single-GPU variant (_1GPU.yaml:131) — the same 16 prompts x 4 views per optimizer step either way."""
    cfg = asd_sd_hyper_ingp(prompts, "hip-mvdream")
    cfg["name"] = "asd_mv_triplane_100k"
    cfg["trainer"] = {"max_steps": 100000, "precision": 32, "accumulate_grad_batches": 2 if n_gpus >= 8 else 8}
    lib = cfg["data"]["prompt_library"]
    cfg["data_type"] = "multiprompt-multiview-camera-datamodule"
    cfg["data"] = {"batch_size": 4, "n_view": 4, "width": 64, "height": 64, "camera_distance_range": [0.8, 1.0], "fovy_range": [15, 60],
                   "elevation_range": [0, 30], "camera_perturb": 0.0, "center_perturb": 0.0, "up_perturb": 0.0,
                   "eval_camera_distance": 3.0, "eval_fovy_deg": 40.0, "n_val_views": 40, "prompt_library": lib, "dim_gaussian": 1}
    s = cfg["system"]
    s["geometry_type"] = "Triplane-transformer-sdf"
    s["geometry"] = {"radius": 2.0, "normal_type": "finite_difference", "finite_difference_normal_eps": 0.01, "sdf_bias": "sphere",
                     "sdf_bias_params": 0.8,
                     "space_generator_config": {"inner_dim": 768, "condition_dim": 1024, "triplane_low_res": 32, "triplane_high_res": 64,
                                                "triplane_dim": 32, "num_layers": 12, "num_heads": 16, "mlp_ratio": 4, "local_text": True}}
    s["material"] = {"n_output_dims": 3, "color_activation": "sigmoid-mipnerf", "requires_normal": True}
    s["background_type"] = "neural-environment-map-background"
    s["background"] = {"color_activation": "sigmoid-mipnerf", "random_aug": False}
    s["prompt_processor"] = {"pretrained_model_name_or_path": "pretrained/stable-diffusion-2-1-base", "use_local_text_embeddings": True}
    s["guidance_type"] = "mvdream-asynchronous-score-distillation-guidance"
    s["guidance"] = {"model_name": "sd-v2.1-base-4view", "ckpt_path": "pretrained/sd-v2.1-base-4view.pt", "guidance_scale": 7.5,
                     "plus_ratio": 0.1, "plus_random": True, "min_step_percent": [0, 0.5, 0.02, 100000],
                     "max_step_percent": [0, 0.98, 0.5, 100000], "backend": "hip-mvdream", "allow_random_weights": ALLOW_RANDOM_WEIGHTS}
    s["loss"] = {"lambda_asd": 1.0, "lambda_orient": 0.0, "lambda_sparsity": 20, "lambda_opaque": [80000, 0, 1.0, 100000],
                 "lambda_z_variance": 0.0, "lambda_eikonal": 0.01}
    s["optimizer"] = {"name": "Adan", "args": {"betas": [0.98, 0.92, 0.99], "eps": 1.0e-15},
                      "params": {"geometry": {"lr": 0.0002}, "background": {"lr": 0.0002}}}
    return cfg


def nerf_only_c1() -> dict:
    """BASELINE config 1: single prompt, 32x32 rays, 16 samples per ray, NeRF-only render (no diffusion)."""
    cfg = asd_sd_nerf()
    cfg["data"].update({"width": 32, "height": 32, "batch_size": 1, "resolution_milestones": []})
    cfg["system"]["renderer"]["num_samples_per_ray"] = 16
    cfg["system"]["guidance_type"] = ""
    return cfg


def apply_trainer(system, cfg: dict):
    """the `trainer:` keys of a config that change the step itself (Lightning options in the reference): accumulate_grad_batches"""
    system.accumulate_grad_batches = int(cfg.get("trainer", {}).get("accumulate_grad_batches", 1))
    return system
