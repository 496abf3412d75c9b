"""ctypes binding of libasd_hip.so (C ABI: include/asd_hip.h).

The product path has no fallback: if the HIP library is missing or a call fails, an exception is raised.
PyTorch only supplies device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASD_HIP_LIB") or os.path.join(_HERE, "libasd_hip.so")   # ASD_HIP_LIB: A/B builds of the same C ABI (tools/)
ASD_MAX_LEVELS = 16

ASD_BIAS_CONST, ASD_BIAS_BLOB_MAGIC3D, ASD_BIAS_BLOB_DREAMFUSION, ASD_BIAS_SPHERE = 0, 1, 2, 3
ASD_FIELD_DENSITY, ASD_FIELD_SDF = 0, 1
ASD_ACT_SOFTPLUS, ASD_ACT_EXP, ASD_ACT_TRUNC_EXP, ASD_ACT_NONE = 0, 1, 2, 3


class AsdError(RuntimeError):
    pass


class GridMeta(C.Structure):
    _fields_ = [
        ("n_levels", C.c_uint32),
        ("n_features", C.c_uint32),
        ("n_params", C.c_uint32),
        ("reserved", C.c_uint32),
        ("scale", C.c_float * ASD_MAX_LEVELS),
        ("resolution", C.c_uint32 * ASD_MAX_LEVELS),
        ("offset", C.c_uint32 * ASD_MAX_LEVELS),
        ("size", C.c_uint32 * ASD_MAX_LEVELS),
        ("dense", C.c_uint32 * ASD_MAX_LEVELS),
    ]


class FieldCfg(C.Structure):
    _fields_ = [
        ("bbox_min", C.c_float * 3),
        ("bbox_max", C.c_float * 3),
        ("radius", C.c_float),
        ("bias_mode", C.c_int32),
        ("bias_value", C.c_float),
        ("blob_scale", C.c_float),
        ("blob_std", C.c_float),
        ("activation", C.c_int32),
        ("fd_eps", C.c_float),
        ("n_hidden", C.c_int32),
        ("n_feature_dims", C.c_int32),
        ("field_mode", C.c_int32),
    ]


class MarchCfg(C.Structure):
    _fields_ = [
        ("aabb", C.c_float * 6),
        ("resolution", C.c_int32),
        ("near_plane", C.c_float),
        ("far_plane", C.c_float),
        ("step", C.c_float),
        ("max_steps", C.c_int32),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32),
        ("bias", C.c_void_p), ("row_bias", C.c_void_p), ("rows_per_group", C.c_int32),
        ("residual", C.c_void_p), ("ldr", C.c_int32), ("act", C.c_int32), ("out_f32", C.c_int32),
        ("conv", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Hout", C.c_int32),
        ("Wout", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("upsample", C.c_int32),
        ("zero_page", C.c_void_p), ("split_k", C.c_int32), ("workspace", C.c_void_p),
        ("tile_cfg", C.c_int32), ("ld_row_bias", C.c_int32), ("group_m", C.c_int32), ("group_n", C.c_int32),
        ("gn_partials", C.c_void_p), ("gn_cg", C.c_int32), ("gn_rows", C.c_int32),
        ("gn_bwd_x", C.c_void_p), ("gn_bwd_fstats", C.c_void_p), ("gn_bwd_gamma", C.c_void_p), ("gn_bwd_beta", C.c_void_p),
        ("gn_eps", C.c_float), ("gn_silu", C.c_int32), ("wide_rows", C.c_int32),
        ("ln_mode", C.c_int32), ("ln_eps", C.c_float), ("ln_sc", C.c_void_p), ("ln_stats", C.c_void_p),
        ("a_seg_rows", C.c_int32), ("w_seg_rows", C.c_int32), ("a_seg_off", C.c_int32 * 9), ("w_seg_off", C.c_int32 * 6),
        ("partials_only", C.c_int32),
        ("gn_apply", C.c_int32), ("gn_apply_y", C.c_void_p), ("gn_apply_gamma", C.c_void_p), ("gn_apply_beta", C.c_void_p),
        ("gn_apply_eps", C.c_float), ("gn_apply_silu", C.c_int32), ("gn_apply_stats", C.c_void_p),
    ]


class RenderParams(C.Structure):
    _fields_ = [("march", MarchCfg), ("meta", C.c_void_p), ("field", C.c_void_p), ("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("n_rays", C.c_int32),
                ("occ_bits", C.c_void_p), ("jitter", C.c_void_p), ("grid", C.c_void_p), ("w1d", C.c_void_p), ("w2d", C.c_void_p), ("w1f", C.c_void_p),
                ("w2f", C.c_void_p), ("bg", C.c_void_p), ("early_stop_eps", C.c_float), ("alpha_thre", C.c_float), ("prune", C.c_int32),
                ("color_act", C.c_int32), ("capacity", C.c_int32)]


RENDER_LAYOUT_FIELDS = ["total_bytes", "count", "offset", "total", "c_ray_idx", "c_t0", "c_t1", "c_pts", "c_sigma", "keep", "kept", "koff", "n_kept",
                        "ray_idx", "t0", "t1", "pts", "dirs", "sigma", "feats", "enc", "weights", "opacity", "depth", "z_var", "rgb_fg", "comp_rgb",
                        "c_feats", "c_enc"]


class RenderLayout(C.Structure):
    _fields_ = [(k, C.c_int64) for k in RENDER_LAYOUT_FIELDS]


class Conv3dDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("amax_x", C.c_void_p), ("amax_dy", C.c_void_p)]


class Conv3dEpilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("noise", C.c_void_p), ("noise_strength", C.c_void_p), ("act", C.c_int32), ("gain", C.c_float),
                ("clamp", C.c_float), ("amax_out", C.c_void_p)]


class OptTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n2", C.c_void_p), ("prev", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("weight_decay", C.c_float), ("bias_correction1", C.c_float),
                ("bias_correction2", C.c_float), ("bias_correction2_sqrt", C.c_float)]


class TritxDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dim", C.c_int32), ("heads", C.c_int32), ("cond_dim", C.c_int32), ("cond_tokens", C.c_int32),
                ("hidden", C.c_int32), ("low_res", C.c_int32), ("out_channels", C.c_int32), ("eps", C.c_float), ("grads_prezeroed", C.c_int32)]


class WeightInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class UNetDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("model_channels", C.c_int32), ("num_res_blocks", C.c_int32),
                ("n_levels", C.c_int32), ("channel_mult", C.c_int32 * 8), ("attention_ds_mask", C.c_int32),
                ("num_head_channels", C.c_int32), ("transformer_depth", C.c_int32), ("context_dim", C.c_int32), ("camera_dim", C.c_int32)]


class VaeDesc(C.Structure):
    _fields_ = [("in_channels", C.c_int32), ("ch", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("num_res_blocks", C.c_int32), ("z_channels", C.c_int32), ("embed_dim", C.c_int32)]


_lib: Optional[C.CDLL] = None

# every symbol include/asd_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "asd_grid_meta_init", "asd_hashgrid_fwd", "asd_hashgrid_bwd",
    "asd_field_density", "asd_field_fwd", "asd_field_bwd_workspace", "asd_field_bwd",
    "asd_envmap_fwd", "asd_envmap_bwd",
    "asd_importance_resample", "asd_transmittance_cdf", "asd_merge_sorted", "asd_voxel_sample_fwd", "asd_voxel_sample_bwd", "asd_voxel_sample_bwd_rows",
    "asd_triplane_sample_fwd", "asd_triplane_sample_bwd", "asd_triplane_sample_bwd_rows", "asd_relayout_f32",
    "asd_generate_rays", "asd_march_count", "asd_scan_i32", "asd_march_write", "asd_prune_count", "asd_compact",
    "asd_occgrid_update", "asd_composite_fwd", "asd_composite_bwd",
    "asd_gemm_f16", "asd_gemm_force_tile", "asd_groupnorm_f16", "asd_groupnorm_bwd_f16", "asd_transpose_f16", "asd_layernorm_f16", "asd_softmax_f16", "asd_softmax_bwd_f16", "asd_geglu_f16", "asd_silu_f16",
    "asd_timestep_embedding_f16", "asd_concat_f16", "asd_attention_f16",
    "asd_gemm_plan_set", "asd_gemm_plan_get", "asd_gemm_plan_count", "asd_gemm_plan_generation", "asd_gemm_plan_entry", "asd_gemm_workspace_bytes", "asd_gemm_tune", "asd_gemm_gn_records", "asd_gemm_gn_applies", "asd_groupnorm_apply_f16", "asd_groupnorm_bwd_apply_f16",
    "asd_pad_cast_f16",
    "asd_image_prep_fwd", "asd_image_prep_bwd", "asd_latents_fwd", "asd_score_fwd", "asd_latents_bwd", "asd_prompt_context",
    "asd_unet_create", "asd_unet_destroy", "asd_unet_num_weights", "asd_unet_weight_info", "asd_unet_bind_weights",
    "asd_unet_workspace_bytes", "asd_unet_fwd", "asd_unet_workspace_bytes_shared", "asd_unet_fwd_shared", "asd_gather_rows_f16",
    "asd_vae_enc_create", "asd_vae_enc_destroy", "asd_vae_enc_num_weights", "asd_vae_enc_weight_info", "asd_vae_enc_bind_weights",
    "asd_vae_enc_workspace_bytes", "asd_vae_enc_fwd", "asd_vae_enc_bwd",
    "asd_adamw_f32", "asd_adan_f32",
    "asd_conv3d_workspace_bytes", "asd_conv3d_fwd", "asd_conv3d_dgrad", "asd_conv3d_wgrad", "asd_layer_act_bwd", "asd_upsample3d_fwd", "asd_upsample3d_bwd",
    "asd_torgb_fwd", "asd_torgb_bwd", "asd_absmax_f32", "asd_voxfield_fwd", "asd_voxfield_bwd_workspace", "asd_voxfield_bwd",
    "asd_trifield_fwd_workspace", "asd_trifield_fwd", "asd_trifield_bwd_workspace", "asd_trifield_bwd",
    "asd_render_layout_init", "asd_render_fwd", "asd_render_bwd_workspace", "asd_render_bwd",
    "asd_tx_pack_weight", "asd_tx_linear_workspace", "asd_tx_linear", "asd_tx_wgrad_workspace", "asd_tx_linear_wgrad", "asd_tx_layernorm_fwd", "asd_tx_layernorm_bwd",
    "asd_tx_attention_workspace", "asd_tx_attention_fwd", "asd_tx_attention_bwd",
    "asd_tritx_packed_floats", "asd_tritx_save_floats", "asd_tritx_workspace_floats", "asd_tritx_pack", "asd_tritx_fwd", "asd_tritx_bwd",
    "asd_comm_unique_id", "asd_comm_create", "asd_comm_destroy", "asd_allreduce_mean_f32",
    "asd_version", "asd_last_error", "asd_modulated_weights_fwd", "asd_modulated_weights_bwd", "asd_timestep_plus", "asd_loss_tail_fwd", "asd_loss_tail_bwd", "asd_probe_events", "asd_probe_mark",
]


def lib() -> C.CDLL:
    """Load libasd_hip.so (once). Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AsdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C scaledreamer_amd/csrc). The HIP path has no fallback."
            )
        l = C.CDLL(LIB_PATH)
        l.asd_last_error.restype = C.c_char_p
        l.asd_version.restype = C.c_char_p
        l.asd_grid_meta_init.restype = C.c_uint32
        l.asd_grid_meta_init.argtypes = [C.POINTER(GridMeta), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double]
        for fn in ("asd_gemm_workspace_bytes", "asd_unet_workspace_bytes", "asd_unet_workspace_bytes_shared", "asd_vae_enc_workspace_bytes",
                   "asd_conv3d_workspace_bytes", "asd_tx_linear_workspace", "asd_tx_wgrad_workspace", "asd_tx_attention_workspace",
                   "asd_tritx_packed_floats", "asd_tritx_save_floats", "asd_tritx_workspace_floats"):
            getattr(l, fn).restype = C.c_int64
        l.asd_unet_destroy.restype = None
        l.asd_vae_enc_destroy.restype = None
        l.asd_unet_destroy.argtypes = [C.c_void_p]
        l.asd_vae_enc_destroy.argtypes = [C.c_void_p]
        _lib = l
    return _lib


def check(status: int) -> None:
    if status != 0:
        raise AsdError(f"libasd_hip error {status}: {lib().asd_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_contiguous():
        raise AsdError("non-contiguous tensor passed to the C ABI")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(v: float) -> C.c_float:
    return C.c_float(v)


def i32(v: int) -> C.c_int32:
    return C.c_int32(v)


def make_grid_meta(n_levels: int, n_features: int, log2_hashmap_size: int, base_resolution: int,
                   per_level_scale: float) -> GridMeta:
    m = GridMeta()
    n = lib().asd_grid_meta_init(C.byref(m), n_levels, n_features, log2_hashmap_size, base_resolution,
                                 float(per_level_scale))
    if n == 0:
        raise AsdError(lib().asd_last_error().decode())
    return m
