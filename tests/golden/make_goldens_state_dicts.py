"""State-dict key / shape tables of the reference's own modules (build container only) -> tests/golden/state_dict_keys.json.
The product classes must expose exactly these keys so that ScaleDreamer checkpoints load.
(tinycudann / nerfacc are stand-ins under the harness: `...encoding.params` is tcnn's flat table; the estimator's buffers are
third-party and not compared.)"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install_amortized()
from scaledreamer_amd import generators as G  # noqa: E402

H._mod("diffusers"); H._mod("diffusers.models"); H._mod("diffusers.models.attention_processor", Attention=G.Attention)
from make_goldens_amortized import BG_ENC, ENC, GEN3D_SMALL, HYPER, TRI_SMALL  # noqa: E402

from custom.amortized.models.background.multiprompt_neural_environment_hashgrid_map_background import \
    MultipromptNeuralHashgridEnvironmentMapBackground  # noqa: E402
from custom.amortized.models.geometry.hyper_iNGP import Hypernet_Sdf  # noqa: E402
from custom.amortized.models.geometry.stylegan_3dconv_net import Voxel_3d_Sdf  # noqa: E402
from custom.amortized.models.geometry.triplane_transformer import TriplaneTransformerSDF  # noqa: E402
from custom.amortized.models.renderers.generative_space_volsdf_volume_renderer import GenerativeSpaceVolSDFVolumeRenderer  # noqa: E402
from threestudio.models.background.neural_environment_map_background import NeuralEnvironmentMapBackground  # noqa: E402
from threestudio.models.geometry.implicit_volume import ImplicitVolume  # noqa: E402
from threestudio.models.materials.no_material import NoMaterial  # noqa: E402

BG4 = {"otype": "HashGrid", "n_features_per_level": 2, "log2_hashmap_size": 19, "n_levels": 4, "base_resolution": 4, "per_level_scale": 4.0}
table = {}


def add(name, module, skip=("estimator.",)):
    table[name] = {k: list(v.shape) for k, v in module.state_dict().items() if not any(k.startswith(s) for s in skip)}


geo = ImplicitVolume({"radius": 1.0, "normal_type": "finite_difference", "pos_encoding_config": ENC})
add("implicit-volume", geo)
bg = NeuralEnvironmentMapBackground({"color_activation": "sigmoid", "random_aug": True, "dir_encoding_config": BG4})
add("neural-environment-map-background", bg)
add("no-material", NoMaterial({"n_output_dims": 3, "color_activation": "sigmoid", "requires_normal": True}))
hg = Hypernet_Sdf({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.5, "hypernet_config": HYPER, "pos_encoding_config": ENC})
add("Hyper-iNGP", hg)
hb = MultipromptNeuralHashgridEnvironmentMapBackground({"color_activation": "sigmoid", "pos_encoding_config": BG_ENC})
add("multiprompt-neural-hashgrid-environment-map-background", hb)
ren = GenerativeSpaceVolSDFVolumeRenderer({"radius": 2.0, "use_volsdf": True, "trainable_variance": False, "learned_variance_init": 0.340119,
                                           "estimator": "importance", "num_samples_per_ray": 64, "num_samples_per_ray_importance": 128},
                                          geometry=hg, material=None, background=hb)
add("generative-space-volsdf-volume-renderer", ren)
add("3DConv-net", Voxel_3d_Sdf({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.8, "space_generator_config": GEN3D_SMALL}))
add("Triplane-transformer-sdf", TriplaneTransformerSDF({"radius": 2.0, "sdf_bias": "sphere", "sdf_bias_params": 0.8, "space_generator_config": TRI_SMALL}))
with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
    json.dump(table, f, indent=0, sort_keys=True)
print({k: len(v) for k, v in table.items()})
