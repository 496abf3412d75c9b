/*
 * asd_hip.h — C ABI of libasd_hip.so: the MI355X (gfx950) implementation of ScaleDreamer's
 * Asynchronous-Score-Distillation inner loop.
 *
 * The reference (theEricMa/ScaleDreamer) defines no C symbols of its own: every native call on this
 * path goes into a pip-installed third-party CUDA package.  Each entry point below therefore cites the
 * *reference call site* whose native callee it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers + sizes only, no torch types; every pointer is DEVICE memory unless marked [host]
 *   - the CALLER owns all memory (outputs, workspaces); nothing is allocated or freed in here
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); no call synchronises
 *   - return value: 0 = ok, non-zero = error; text via asd_last_error() (thread local)
 *   - sample arrays are "packed": samples of ray r occupy [ray_offset[r], ray_offset[r]+ray_count[r])
 *   - sizes that only exist on the device (number of marched / kept samples) are passed as
 *     `const int32_t* n_dev`; the launch is sized by the host-known upper bound `n_max`
 */
#ifndef ASD_HIP_H
#define ASD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASD_MAX_LEVELS 16
#define ASD_OK 0
#define ASD_ERR_ARG 1
#define ASD_ERR_LAUNCH 2
#define ASD_ERR_UNSUPPORTED 3

/* ------------------------------------------------------------------------------------------------
 * Multiresolution hash grid (replaces tinycudann `tcnn.Encoding(n_in, {"otype":"HashGrid",...})`,
 * threestudio/models/networks.py:55-64; behaviour spec SURVEY.md Appendix B.1)
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_grid_meta {
    uint32_t n_levels;            /* L */
    uint32_t n_features;          /* F, must be 2 */
    uint32_t n_params;            /* total floats = sum(size[l]) * F */
    uint32_t reserved;
    float    scale[ASD_MAX_LEVELS];       /* 2^(l*log2(per_level_scale)) * base_res - 1 */
    uint32_t resolution[ASD_MAX_LEVELS];  /* ceil(scale)+1 */
    uint32_t offset[ASD_MAX_LEVELS];      /* first entry (in F-tuples) of the level's table */
    uint32_t size[ASD_MAX_LEVELS];        /* entries in the level's table (hashmap size) */
    uint32_t dense[ASD_MAX_LEVELS];       /* 1: res^3 <= size (direct index), 0: hashed */
} asd_grid_meta;

/* [host] fill `meta`; returns the number of fp32 parameters (12 599 920 for the asd_sd_nerf geometry). */
uint32_t asd_grid_meta_init(asd_grid_meta* meta, uint32_t n_levels, uint32_t n_features,
                            uint32_t log2_hashmap_size, uint32_t base_resolution, double per_level_scale);

/* out[n, L*F] = encode(x[n,3] in [0,1]);  tcnn.Encoding.forward (networks.py:64). */
int asd_hashgrid_fwd(const asd_grid_meta* meta, const float* params, const float* x, int32_t n,
                     float* out, void* stream);
/* dparams += scatter(dout[n, L*F]);  autograd backward of the call above (tcnn kernel_grid_backward). */
int asd_hashgrid_bwd(const asd_grid_meta* meta, const float* x, const float* dout, int32_t n,
                     float* dparams, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused field: contract -> hash grid -> VanillaMLP(32->H->1) + density bias + activation,
 * VanillaMLP(32->H->C) features, finite-difference normal.
 * Replaces ImplicitVolume.forward / forward_density (threestudio/models/geometry/implicit_volume.py:
 * 109-207) = TCNNEncoding + 2x VanillaMLP (networks.py:214-251) + get_activated_density (:80-107).
 * ---------------------------------------------------------------------------------------------- */
enum { ASD_BIAS_CONST = 0, ASD_BIAS_BLOB_MAGIC3D = 1, ASD_BIAS_BLOB_DREAMFUSION = 2,
       ASD_BIAS_SPHERE = 3 /* |p| - bias_value: sdf_bias "sphere", custom/amortized/models/geometry/hyper_iNGP.py:222-225 */ };
/* field_mode: DENSITY = ImplicitVolume (normal = -grad sigma / |.|); SDF = the amortized geometries' signed-distance
 * head (Hyper-iNGP, hyper_iNGP.py:263-330): raw MLP output + bias, no activation, sdf_grad = +(sdf(x+eps e_k) - sdf(x))/eps,
 * normal = sdf_grad / |sdf_grad|.  In SDF mode the per-prompt weights of the hypernetwork are passed as the MLP weights. */
enum { ASD_FIELD_DENSITY = 0, ASD_FIELD_SDF = 1 };
enum { ASD_ACT_SOFTPLUS = 0, ASD_ACT_EXP = 1, ASD_ACT_TRUNC_EXP = 2, ASD_ACT_NONE = 3 };

typedef struct asd_field_cfg {
    float   bbox_min[3], bbox_max[3];   /* geometry.bbox (geometry/base.py:72-83) */
    float   radius;                     /* clamp range of finite-difference offsets */
    int32_t bias_mode;                  /* ASD_BIAS_* */
    float   bias_value;                 /* ASD_BIAS_CONST */
    float   blob_scale, blob_std;
    int32_t activation;                 /* ASD_ACT_* */
    float   fd_eps;                     /* finite_difference_normal_eps */
    int32_t n_hidden;                   /* 64 */
    int32_t n_feature_dims;             /* 3 (0: no feature network) */
    int32_t field_mode;                 /* ASD_FIELD_* */
} asd_field_cfg;

/* sigma[n] only (no_grad): geometry.forward_density — used by the marcher's sigma_fn
 * (nerf_volume_renderer.py:153-167) and the occupancy update (:436-439).
 * If n_dev != NULL the live count is read on the device and n is the launch bound. */
int asd_field_density(const asd_grid_meta* meta, const asd_field_cfg* cfg, const float* grid_params,
                      const float* w1_density /*[H,32]*/, const float* w2_density /*[1,H]*/,
                      const float* points /*[n,3]*/, int32_t n, const int32_t* n_dev,
                      float* sigma /*[n]*/, void* stream);

/* Training forward at kept samples: geometry(points, output_normal) (nerf_volume_renderer.py:282-284).
 * Outputs: sigma[n] (the sdf in SDF mode), features[n,C] (pre-activation), normal[n,3] (NULL: skip the 3 offset
 * evaluations), fd_grad[n,3] (NULL or the un-normalised finite-difference gradient: `sdf_grad`, hyper_iNGP.py:305-318).
 * enc_save[n, L*F] keeps the centre encoding for the backward pass. */
int asd_field_fwd(const asd_grid_meta* meta, const asd_field_cfg* cfg, const float* grid_params,
                  const float* w1_density, const float* w2_density,
                  const float* w1_feature /*[H,32]*/, const float* w2_feature /*[C,H]*/,
                  const float* points, int32_t n, const int32_t* n_dev,
                  float* sigma, float* features, float* normal, float* fd_grad, float* enc_save, void* stream);

/* Backward: given dL/dsigma[n], dL/dfeatures[n,C], dL/dnormal[n,3], dL/dfd_grad[n,3] (any may be NULL) accumulate (+=)
 * d_grid_params (atomic scatter) and the MLP weight gradients dw1_density[H,32], dw2_density[1,H],
 * dw1_feature[H,32], dw2_feature[C,H].  `workspace` holds asd_field_bwd_workspace() floats
 * (hidden-layer gradients per sample + per-chunk partial sums, reduced in a fixed order). */
int asd_field_bwd_workspace(const asd_field_cfg* cfg, int32_t n, int32_t with_normal, int64_t* n_floats);
int asd_field_bwd(const asd_grid_meta* meta, const asd_field_cfg* cfg, const float* grid_params,
                  const float* w1_density, const float* w2_density,
                  const float* w1_feature, const float* w2_feature,
                  const float* points, const float* enc_save, const float* sigma /* forward output */,
                  int32_t n, const int32_t* n_dev, const float* d_sigma, const float* d_features, const float* d_normal,
                  const float* d_fd_grad, float* d_grid_params, float* dw1_density, float* dw2_density, float* dw1_feature, float* dw2_feature,
                  float* workspace, void* stream);

/* The same fused field over a SAMPLED feature volume (`3DConv-net`, custom/amortized/models/geometry/stylegan_3dconv_net.py:244-346:
 * contract -> get_trilinear_feature -> VanillaMLP sdf / feature heads -> sdf + bias -> finite-difference sdf_grad): the encoding of a
 * point is the trilinear sample (F.grid_sample, zeros, align_corners = False) of voxel_cl [D][H][W][32] instead of the hash grid, the
 * heads / bias / finite differences are asd_field_fwd's in ASD_FIELD_SDF mode (cfg).  One batch entry per call.  Backward: d_voxel_cl +=
 * (atomics; the scatter is asd_voxel_sample_bwd's), weight gradients += as asd_field_bwd; workspace: asd_voxfield_bwd_workspace floats. */
int asd_voxfield_fwd(const float* voxel_cl, int32_t D, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* w1_sdf,
                     const float* w2_sdf, const float* w1_feature, const float* w2_feature, const float* points, int32_t n, float* sdf,
                     float* features, float* normal, float* fd_grad, float* enc_save, void* stream);
int asd_voxfield_bwd_workspace(const asd_field_cfg* cfg, int32_t n, int32_t with_normal, int64_t* n_floats);
int asd_voxfield_bwd(const float* voxel_cl, int32_t D, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* w1_sdf,
                     const float* w2_sdf, const float* w1_feature, const float* w2_feature, const float* points, const float* enc_save,
                     const float* sdf, int32_t n, const float* d_sdf, const float* d_features, const float* d_normal,
                     const float* d_fd_grad, float* d_voxel_cl, float* dw1_sdf, float* dw2_sdf, float* dw1_feature, float* dw2_feature,
                     float* workspace, void* stream);

/* The fused field of `Triplane-transformer-sdf` (custom/amortized/models/geometry/triplane_transformer.py:139-240): contract -> three
 * bilinear plane lookups (sample_from_planes, geometry/utils.py:81-93; planes_cl [3][H][W][32] of ONE batch entry) -> two VanillaMLP heads
 * 96 -> 64 -> 64 -> 1 | 3 -> sdf + bias -> finite-difference sdf_grad.  cfg: ASD_FIELD_SDF with sphere / constant bias.
 * weights [host array of 6 device pointers]: sdf head W1^T [96][64] (the first layer TRANSPOSED), W2 [64][64], W3 [1][64]; feature head
 * W1^T, W2, W3 [3][64].  Backward: nothing but the points and the sdf is kept from the forward pass; d_planes_cl += (atomics),
 * d_weights [host array of 6 device pointers] += with the FIRST-layer gradients in the NATIVE layout [64][96]; workspace:
 * asd_trifield_bwd_workspace floats (the pass walks the samples in chunks). */
int asd_trifield_fwd_workspace(int32_t H, int32_t W, int64_t* n_floats);   /* scales + split-fp16 fragment images of the two heads + the zero-bordered planes, rebuilt by every call */
int asd_trifield_fwd(const float* planes_cl, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* const* weights,
                     const float* points, int32_t n, float* sdf, float* features /* or NULL */, float* normal, float* fd_grad,
                     float* workspace, void* stream);
int asd_trifield_bwd_workspace(int32_t H, int32_t W, int32_t n, int32_t with_normal, int64_t* n_floats);
int asd_trifield_bwd(const float* planes_cl, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* const* weights,
                     const float* points, const float* sdf, int32_t n, const float* d_sdf, const float* d_features, const float* d_normal,
                     const float* d_fd_grad, float* d_planes_cl, float* const* d_weights, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Background: (d+1)/2 -> hash grid (L levels) -> VanillaMLP(2L -> H -> H -> 3) -> sigmoid.
 * Replaces NeuralEnvironmentMapBackground.forward
 * (threestudio/models/background/neural_environment_map_background.py:46-67).
 * ---------------------------------------------------------------------------------------------- */
int asd_envmap_fwd(const asd_grid_meta* meta, const float* grid_params,
                   const float* w0 /*[H,2L]*/, const float* w1 /*[H,H]*/, const float* w2 /*[3,H]*/,
                   int32_t n_hidden, const float* dirs /*[n,3]*/, int32_t n, float* color /*[n,3]*/,
                   void* stream);
/* backward: d_grid_params / dw0 / dw1 / dw2 are accumulated with atomics (caller zeroes them). */
int asd_envmap_bwd(const asd_grid_meta* meta, const float* grid_params,
                   const float* w0, const float* w1, const float* w2, int32_t n_hidden,
                   const float* dirs, const float* d_color, int32_t n,
                   float* d_grid_params, float* dw0, float* dw1, float* dw2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Amortized (multi-prompt) render path: importance sampling on dense [n_rays, edges] tensors, replacing the
 * nerfacc v0.5.2 calls of ImportanceEstimator.sampling (threestudio/models/estimators.py:72-74,84-88,93-94).
 * Placement convention (nerfacc's jitter stream is unpinned): output edge j of a ray sits at cdf value
 * u_j = (j + jitter[r]) / (n_out + 1) when jitter != NULL (stratified: one jitter per ray) else j / n_out; the edge is
 * linearly interpolated inside the cdf segment [p, p+1], p = last index with cdf[p] <= u.
 * ---------------------------------------------------------------------------------------------- */
int asd_importance_resample(const float* vals /*[n_rays,e_in] sorted*/, const float* cdfs /*[n_rays,e_in]*/, int32_t n_rays,
                            int32_t e_in, int32_t n_out, const float* jitter /*[n_rays] or NULL*/,
                            float* out /*[n_rays,n_out+1]*/, void* stream);
/* cdf[r,j] = 1 - exp(-sum_{k<j} sigma[r,k] (t[r,k+1] - t[r,k])), cdf[r,S] = 1: render_transmittance_from_density +
 * `1 - cat([trans, 0])` (estimators.py:84-86). */
int asd_transmittance_cdf(const float* t_edges /*[n_rays,S+1]*/, const float* sigma /*[n_rays,S]*/, int32_t n_rays,
                          int32_t n_samples, float* cdf /*[n_rays,S+1]*/, void* stream);
/* per-ray merge of two sorted edge lists = sort(cat([a, b], -1)) (estimators.py:93-94); ties keep a first. */
int asd_merge_sorted(const float* a, int32_t na, const float* b, int32_t nb, int32_t n_rays, float* out /*[n_rays,na+nb]*/,
                     void* stream);

/* Feature samplers of the generator-backed geometries (custom/amortized/models/geometry/utils.py):
 * get_trilinear_feature (:95-110) = F.grid_sample(voxel[B,C,D,H,W], points, bilinear, zeros, align_corners=False) and
 * sample_from_planes (:81-93) = 3 bilinear grid_samples on planes[B,3,C,H,W] with the projections (x,y), (x,z), (z,y).
 * Features are stored channel-LAST ([B,D,H,W,C] / [B,3,H,W,C], C in {4,8,16,32,64}); asd_relayout_f32 converts
 * [batch, rows, cols] -> [batch, cols, rows] (NCDHW <-> NDHWC with rows = C, cols = D*H*W).  points [B,M,3] in [-1,1]:
 * x -> W, y -> H, z -> D.  Backward accumulates (+=, atomics) into the zero-initialised feature gradient. */
int asd_voxel_sample_fwd(const float* voxel_cl, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, const float* points,
                         int32_t M, float* out /*[B,M,C]*/, void* stream);
int asd_voxel_sample_bwd(const float* d_out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, const float* points,
                         int32_t M, float* d_voxel_cl, void* stream);
/* the same scatter for rows in ray order with `run` consecutive rows per lane group (the fused voxel field's backward pass) */
int asd_voxel_sample_bwd_rows(const float* d_out, int32_t D, int32_t H, int32_t W, int32_t C, const float* points, int32_t rows,
                              float* d_voxel_cl, int32_t run, void* stream);
int asd_triplane_sample_fwd(const float* planes_cl, int32_t B, int32_t H, int32_t W, int32_t C, const float* points, int32_t M,
                            float coord_scale /* 2 / box_warp */, float* out /*[B,M,3C]*/, void* stream);
int asd_triplane_sample_bwd(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t C, const float* points, int32_t M,
                            float coord_scale, float* d_planes_cl, void* stream);
/* the same scatter for rows in ray order with `run` consecutive rows per lane group (the fused tri-plane field's backward pass) */
int asd_triplane_sample_bwd_rows(const float* d_out, int32_t H, int32_t W, int32_t C, const float* points, int32_t rows, float* d_planes_cl,
                                 int32_t run, void* stream);
int asd_relayout_f32(const float* x, int32_t batch, int32_t rows, int32_t cols, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Occupancy-grid ray marching (replaces nerfacc.OccGridEstimator.sampling -> CUDA traverse_grids /
 * ray_aabb_intersect / render_visibility_from_density; call site nerf_volume_renderer.py:139-180).
 * Sample placement convention (nerfacc's is unpinned): per ray the samples lie on the
 * lattice t_k = t_begin + k*step, t_begin = max(t_aabb_enter, near + jitter*step); interval k is emitted
 * iff its midpoint is inside the aabb, t_mid <= min(t_aabb_exit, far) and its grid cell is occupied.
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_march_cfg {
    float   aabb[6];        /* xmin ymin zmin xmax ymax zmax */
    int32_t resolution;     /* 32 */
    float   near_plane, far_plane;
    float   step;           /* render_step_size */
    int32_t max_steps;      /* lattice points examined per ray (>= ceil(diag/step)+1) */
} asd_march_cfg;

/* pass 1: count[r] = number of lattice intervals of ray r in occupied cells.
 * `occ_bits`: resolution^3 bits, cell (ix,iy,iz) -> bit ix*res*res + iy*res + iz (nerfacc binaries layout),
 * jitter[n_rays] in [0,1) or NULL (stratified=False). */
int asd_march_count(const asd_march_cfg* cfg, const float* rays_o, const float* rays_d, int32_t n_rays,
                    const uint32_t* occ_bits, const float* jitter, int32_t* count, void* stream);
/* exclusive scan of count[n] -> offset[n], total[0]; single launch. */
int asd_scan_i32(const int32_t* count, int32_t n, int32_t* offset, int32_t* total, void* stream);
/* pass 2: write the packed candidates. */
int asd_march_write(const asd_march_cfg* cfg, const float* rays_o, const float* rays_d, int32_t n_rays,
                    const uint32_t* occ_bits, const float* jitter, const int32_t* offset,
                    int32_t* ray_idx, float* t_start, float* t_end, float* points /*[.,3] midpoints*/,
                    void* stream);
/* visibility pruning (nerfacc render_visibility_from_density): keep[i] = T_i >= early_stop_eps &&
 * alpha_i >= alpha_thre, T exclusive per ray; kept_count[r]. */
int asd_prune_count(const float* sigma, const float* t_start, const float* t_end,
                    const int32_t* offset, const int32_t* count, int32_t n_rays,
                    float early_stop_eps, float alpha_thre, uint8_t* keep, int32_t* kept_count,
                    void* stream);
/* compaction of kept samples into a second packed set (+ per-sample positions and directions). */
int asd_compact(const float* rays_o, const float* rays_d, int32_t n_rays,
                const int32_t* offset, const int32_t* count, const uint8_t* keep,
                const float* t_start, const float* t_end, const int32_t* kept_offset,
                int64_t* ray_idx_out, float* t_start_out, float* t_end_out,
                float* points_out /*[.,3]*/, float* dirs_out /*[.,3]*/, void* stream);

/* Camera rays on the device (replaces the CPU tensor code of get_ray_directions / get_rays, threestudio/utils/ops.py:183-269, called
 * from RandomCameraIterableDataset.collate, threestudio/data/uncond.py:326-337, and the H2D copy of the [B,H,W,3] ray tensors):
 * d = ((i + 0.5 - W/2) / focal, -(j + 0.5 - H/2) / focal, -1), rays_d = R d (normalised when `normalize`), rays_o = c2w[:3,3].
 * c2w: fp32 [B,4,4] row-major; focal: fp32 [B] in pixels; outputs fp32 [B,H,W,3]. */
int asd_generate_rays(const float* c2w, const float* focal, int32_t B, int32_t H, int32_t W, int32_t normalize, float* rays_o,
                      float* rays_d, void* stream);

/* occupancy update (nerfacc update_every_n_steps -> _update):
 * occs[c] = max(occs[c]*decay, occ_new[c]) for the listed cells; then
 * threshold = min(mean(occs), occ_thre) and bits = occs > threshold (two launches inside). */
int asd_occgrid_update(float* occs, int32_t n_cells, const int32_t* cell_idx, const float* occ_new,
                       int32_t n_update, float decay, float occ_thre, uint32_t* occ_bits,
                       uint8_t* binaries /*[n_cells] bool mirror*/, float* scratch /*[2]*/, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Compositing (replaces nerfacc.render_weight_from_density + 5x accumulate_along_rays and the glue in
 * nerf_volume_renderer.py:312-364).
 * ---------------------------------------------------------------------------------------------- */
/* per ray r: T,alpha,w over its packed samples;
 *   opacity=sum w, depth=sum w t, rgb_fg=sum w sigmoid?(c) (colors are passed already activated),
 *   z_mean = sum (w/max(op,1e-5)) t, z_var = [op>0.5] * sum (w/max(op,1e-5)) (t-z_mean)^2,
 *   comp_rgb = rgb_fg + bg*(1-opacity).   t = (t_start+t_end)/2.
 * mode 0: sigma is a density (alpha = 1-exp(-sigma*dt)); mode 1: `sigma` already holds alpha
 *   (nerfacc.render_weight_from_alpha, generative_space_volsdf_volume_renderer.py:362-366);
 * mode 2: alpha in, and z_var is the VolSDF renderer's plain second moment sum_i w_i (t_i - depth)^2 — not normalised by the
 *   opacity, not masked (generative_space_volsdf_volume_renderer.py:380-385) — so that renderer is ONE pass as well. */
int asd_composite_fwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end,
                      const float* rgb /*[n,3]*/, const int32_t* offset, const int32_t* count,
                      int32_t n_rays, const float* bg /*[n_rays,3]*/,
                      float* weights /*[n]*/, float* opacity, float* depth, float* rgb_fg /*[.,3]*/,
                      float* z_var, float* comp_rgb /*[.,3]*/, void* stream);
/* gradients w.r.t. sigma (or alpha), rgb and bg given upstream grads of the per-ray outputs and of
 * the per-sample weights (any upstream pointer may be NULL). */
int asd_composite_bwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end,
                      const float* rgb, const int32_t* offset, const int32_t* count, int32_t n_rays,
                      const float* bg, const float* weights, const float* opacity, const float* depth,
                      const float* d_comp_rgb, const float* d_rgb_fg, const float* d_opacity,
                      const float* d_depth, const float* d_z_var, const float* d_weights,
                      float* d_sigma, float* d_rgb, float* d_bg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The renderer's training pass as one entry point each way: NeRFVolumeRenderer.forward (nerf_volume_renderer.py:118-428) with the
 * occupancy-grid estimator (:139-180), a density field that runs as the fused kernels above (ImplicitVolume, 3 feature dims), a material
 * that is colour = activation(features) (NoMaterial, no_material.py:41-54) and per-ray background colours — march, candidate densities,
 * visibility pruning, compaction, field, compositing — and its backward (compositing gradient -> field gradients).  Every launch is
 * enqueued from the library, every count stays on the device; one caller-owned workspace holds all buffers at `capacity` samples
 * (>= n_rays * march.max_steps), asd_render_layout_init() gives their byte offsets (per-sample buffers are valid up to *n_kept).
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_render_params {
    asd_march_cfg march;
    const asd_grid_meta* meta;          /* [host] */
    const asd_field_cfg* field;         /* [host] */
    const float* rays_o; const float* rays_d; int32_t n_rays;
    const uint32_t* occ_bits;
    const float* jitter;                /* [n_rays] or NULL */
    const float* grid; const float* w1d; const float* w2d; const float* w1f; const float* w2f;
    const float* bg;                    /* [n_rays, 3] */
    float early_stop_eps, alpha_thre;   /* visibility pruning (asd_prune_count) */
    int32_t prune;                      /* 0: every candidate is kept */
    int32_t color_act;                  /* 0: features are colours; 1: sigmoid */
    int32_t capacity;
} asd_render_params;
typedef struct asd_render_layout {      /* byte offsets into the workspace */
    int64_t total_bytes;
    int64_t count, offset, total;                                   /* candidates per ray [n_rays] int32, exclusive scan, sum [1] */
    int64_t c_ray_idx, c_t0, c_t1, c_pts, c_sigma, keep;            /* candidates [capacity] */
    int64_t kept, koff, n_kept;                                     /* kept per ray, exclusive scan, sum [1] */
    int64_t ray_idx /* int64 */, t0, t1, pts, dirs, sigma, feats /* raw */, enc, weights;   /* kept samples [capacity] */
    int64_t opacity, depth, z_var, rgb_fg, comp_rgb;                /* per ray */
    int64_t c_feats, c_enc;                                         /* candidates [capacity]: the field's outputs before the compaction */
} asd_render_layout;
int asd_render_layout_init(int32_t n_rays, int32_t capacity, asd_render_layout* layout);   /* [host] */
int asd_render_fwd(const asd_render_params* p, void* workspace, void* stream);
int asd_render_bwd_workspace(const asd_render_params* p, int64_t* n_floats);
/* upstream gradients of the per-ray outputs (any may be NULL) -> d_grid / dw* (+=, as asd_field_bwd), d_bg [n_rays, 3] (or NULL) */
int asd_render_bwd(const asd_render_params* p, void* workspace, const float* d_comp_rgb, const float* d_rgb_fg, const float* d_opacity,
                   const float* d_depth, const float* d_z_var, float* d_grid, float* dw1d, float* dw2d, float* dw1f, float* dw2f,
                   float* d_bg, float* bwd_workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Frozen diffusion prior (SD-2.1 UNet eps-prediction): replaces the cuDNN / cuBLAS / SDPA kernels behind
 * diffusers' UNet2DConditionModel called at threestudio/models/guidance/stable_diffusion_asd_guidance.py:
 * 319-331 (forward_unet) — layer inventory SURVEY.md Appendix A.1.  Activations are NHWC fp16, accumulation
 * fp32 (the reference runs the UNet in fp16, :38,57-59).
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_gemm_args {
    const void* A;          /* fp16: [M,lda] row-major, or NHWC image [B,Hin,Win,Cin] when conv=1 */
    const void* W;          /* fp16 [N,ldw]: Linear weight, or conv weight packed [Cout][ky][kx][Cin] */
    void*       C;          /* [M,ldc] fp16 (fp32 when out_f32) */
    int32_t M, N, K;
    int32_t lda, ldw, ldc;
    const void* bias;       /* fp16 [N] or NULL */
    const void* row_bias;   /* fp16 [M/rows_per_group, N] or NULL (time-embedding add of a ResBlock) */
    int32_t rows_per_group;
    const void* residual;   /* fp16 [M,ldr] or NULL, added after the activation */
    int32_t ldr;
    int32_t act;            /* 0 none, 1 SiLU, 2 fused GEGLU: W rows (and bias) interleaved in 32-row groups [16 value | 16 gate],
                               C[M, N/2] = (value + b) * gelu(gate + b)  (attention.py:49-56) */
    int32_t out_f32;
    int32_t conv;           /* 0: GEMM, 1: 3x3 convolution (K = 9*Cin) */
    int32_t Hin, Win, Cin, Hout, Wout, stride, pad;
    int32_t upsample;       /* 0: plain, 1: fused nearest-2x upsample, 2: input gradient of a stride-2 conv */
    const void* zero_page;  /* >= 16 B of zeros: source of out-of-range rows / taps */
    int32_t split_k;        /* >= 1; > 1 needs workspace[split_k, M, N] fp32 */
    float*  workspace;
    int32_t tile_cfg;       /* 0: tile chosen by the built-in cost model; 1 + i: tile configuration i (see asd_gemm_force_tile),
                               as found by the caller's autotuner (scaledreamer_amd/diffusion/hip_ops.py) */
    int32_t ld_row_bias;    /* row stride (halfs) of row_bias; 0 = N.  Lets a column slice of a wider matrix be used in place
                               (the UNet keeps every ResBlock's time-embedding projection in one [B, sum Cout] matrix) */
    int32_t group_m, group_n; /* block order: workgroups of one XCD walk group_m x group_n super-tiles so that they share operand
                               tiles through that XCD's L2 (csrc/gemm.hip, asd_xcd_item); 0 = chosen by the library */
    float*  gn_partials;    /* optional: GroupNorm statistics of C computed in the epilogue — one 64-float record {sum_g, sumsq_g} x 32
                               groups per output tile, at index tile_m * tiles_n + tile_n; asd_gemm_gn_records() tells how many
                               records per batch element this launch writes (0: not supported for this shape / plan, the pointer
                               is ignored and the consumer runs its own statistics pass) */
    int32_t gn_cg;          /* channels per group (N / 32) */
    int32_t gn_rows;        /* rows of C per batch element (H*W): a tile must not straddle two of them */
    /* backward form: C is the gradient dy reaching a GroupNorm(+SiLU) whose input x (fp16 [M,N]) and forward sums are given; the
     * records then hold {sum g, sum g*xhat} per group, g = dy * silu'(z) * gamma — the two reductions of the GroupNorm input
     * gradient (asd_groupnorm_bwd_f16's first pass), so the consumer skips its pass over (x, dy) */
    const void* gn_bwd_x;   /* NULL: forward statistics */
    const float* gn_bwd_fstats;            /* [batch*64] forward {sum, sumsq} per group */
    const void* gn_bwd_gamma; const void* gn_bwd_beta;
    float   gn_eps;
    int32_t gn_silu;
    int32_t wide_rows;      /* set by the library: the W tile is staged in a permuted row order so that every lane stores 8 consecutive
                               channels (16 B) per row (csrc/gemm.hip, tile_epilogue); callers leave it 0 */
    /* LayerNorm folded into the GEMM that consumes it (BasicTransformerBlock: attn(norm(x)), ff(norm(x)), attention.py:246-275): with
     * W' = gamma (.) W, s = rowsum(W') and c = W beta prepared at pack time,  LN(x) W^T = rstd * (x W'^T - mean * s) + c,  so the
     * normalised tensor is never written or read — the main loop runs on the RAW rows, only the epilogue changes.
     *   ln_mode 1: A holds the raw rows; their mean / rstd (over all K columns; plain GEMM, split_k is forced to 1) are reduced from
     *              the A fragments in the main loop; ln_sc = {s[N], c[N]} fp32; ln_stats (optional) receives {mean, rstd} per row.
     *   ln_mode 2: the W operand holds the raw rows (C^T form, e.g. V^T = W_v LN(x)^T): their statistics are READ from ln_stats
     *              [N][2] (left there by a mode-1 launch over the same rows); ln_sc = {s[M], c[M]}. */
    int32_t ln_mode;
    float   ln_eps;
    const float* ln_sc;
    float*  ln_stats;
    /* Segmented rows (plain GEMM, K % 64 == 0): row r of A lives at A + a_seg_off[r / a_seg_rows] bytes + (r % a_seg_rows) * lda halfs
     * (likewise W with w_seg_*), instead of A + r * lda.  Lets ONE matrix serve several row blocks that are the same rows read at
     * different (also odd: 2-byte aligned) offsets along K — the 27 taps of the 3-D convolution's weight gradient are the same
     * channel-major plane shifted along the voxel axis (csrc/conv3d.hip).  0 rows = off.  Offsets must be >= 0. */
    int32_t a_seg_rows, w_seg_rows;
    int32_t a_seg_off[9], w_seg_off[6];
    int32_t partials_only;  /* split_k > 1: leave the fp32 slabs workspace[split_k, M, N] unreduced (no epilogue launch; the caller sums them) */
    /* GroupNorm(32)(+SiLU) of C applied by the PRODUCER (ResBlock: GroupNorm32 -> SiLU -> conv, openaimodel.py:206-222, when C has no
     * other reader of its statistics): on a split-K launch the reduction kernel owns whole (batch element, group) blocks, so it stores C,
     * takes the group statistics of the stored values and writes gn_apply_y[M, N] = silu?((C - mean) * rstd * gamma + beta) in the same
     * launch — no records, no apply launch.  asd_gemm_gn_applies() tells whether a launch does (plan-dependent); when it does not,
     * the fields are ignored.  gn_cg / gn_rows as above; gn_apply_stats (optional) receives [batch*64] {sum, sumsq} per group. */
    int32_t gn_apply;
    void*   gn_apply_y;
    const void* gn_apply_gamma; const void* gn_apply_beta;   /* fp16 [N] */
    float   gn_apply_eps;
    int32_t gn_apply_silu;
    float*  gn_apply_stats;
} asd_gemm_args;
/* records per batch element asd_gemm_f16(args) will write to args->gn_partials under the current plan; 0 = none */
int32_t asd_gemm_gn_records(const asd_gemm_args* args);
/* 1 when asd_gemm_f16(args) with args->gn_apply set will write args->gn_apply_y itself under the current plan (pointers may be null) */
int32_t asd_gemm_gn_applies(const asd_gemm_args* args);
int asd_gemm_f16(const asd_gemm_args* args, void* stream);
/* Tuning hook (tools/gemm_sweep.py): force tile configuration `cfg` (index into the table of csrc/gemm.hip: 128x64, 128x128,
 * 256x64, 256x128, 128x320, 256x256, 256x320, 320x128, and for 3x3 stride-1 convolutions the LDS-window kernel with
 * 16x16-pixel patches x 64 / x 128 channels) for all following asd_gemm_f16 calls; -1 restores the cost model. */
int asd_gemm_force_tile(int32_t cfg);

#define ASD_GN_STATS_FLOATS(batch) (64 * (batch) + 64 * (512 + (batch)))
/* GroupNorm(32 groups) [+ SiLU] on NHWC fp16 with fp32 statistics (GroupNorm32, diffusionmodules/util.py:229-231);
 * x may be the channel-concatenation of two tensors (skip connections, openaimodel.py:797-799): x2/c2. */
int asd_groupnorm_f16(const void* x1, int32_t c1, const void* x2, int32_t c2, int32_t batch, int32_t hw,
                      const void* gamma, const void* beta, float eps, int32_t silu, void* y,
                      float* stats /* ASD_GN_STATS_FLOATS(batch): [batch*32*2] sums for the backward pass + per-block partials */,
                      void* stream);
/* the same GroupNorm when the tensor's producer already left its statistics as `records` 64-float records per batch element
 * (asd_gemm_args.gn_partials): no statistics pass over x.  stats: ASD_GN_STATS_FLOATS(batch) floats; its first batch*64 receive the
 * per-(batch, group) sums (kept for a backward pass). */
int asd_groupnorm_apply_f16(const void* x, int32_t c, int32_t batch, int32_t hw, const void* gamma, const void* beta, float eps,
                            int32_t silu, const float* partials, int32_t records, void* y, float* stats, void* stream);
/* Input gradient of GroupNorm(+SiLU) with frozen gamma/beta (VAE encoder backward, the reference keeps the
 * VAE in the autograd graph: stable_diffusion_asd_guidance.py:171-178,225): dx from x, dy and the forward stats. */
int asd_groupnorm_bwd_f16(const void* x, const void* dy, int32_t c, int32_t batch, int32_t hw, const void* gamma,
                          const void* beta, float eps, int32_t silu, const float* fwd_stats,
                          const void* dx_add /* optional [batch, hw, c] fp16 added to the result: the gradient reaching x through
                                                its other consumer (the ResnetBlock shortcut, model.py:141-148) */,
                          void* dx, float* bwd_stats /* workspace: ASD_GN_STATS_FLOATS(batch) - 64*batch floats */, void* stream);
/* the same input gradient when the producer of dy already left the two reductions as `records` 64-float records per batch element
 * (asd_gemm_args.gn_partials with gn_bwd_x set) */
int asd_groupnorm_bwd_apply_f16(const void* x, const void* dy, int32_t c, int32_t batch, int32_t hw, const void* gamma, const void* beta,
                                float eps, int32_t silu, const float* fwd_stats, const float* partials, int32_t records,
                                const void* dx_add, void* dx, float* scratch /* (16 * batch) * 64 floats */, void* stream);
/* y[cols, rows] = x[rows, cols]^T, fp16 (operand layout changes for the attention-backward GEMMs). */
int asd_transpose_f16(const void* x, int32_t rows, int32_t cols, int32_t ldx, void* y, int32_t ldy, void* stream);
/* LayerNorm over the last dim (attention.py:265-267), fp16 in/out, fp32 statistics. */
int asd_layernorm_f16(const void* x, int32_t rows, int32_t c, const void* gamma, const void* beta, float eps,
                      void* y, void* stream);
/* Row softmax y = softmax(scale * x) and its input gradient ds = scale * p o (dp - rowsum(dp o p)), fp16 in/out with
 * fp32 statistics: the single-head 512-wide attention of the VAE encoder's mid block (AttnBlock,
 * extern/mvdream/ldm/modules/diffusionmodules/model.py:170-227: bmm(q,k) * c^-0.5 -> softmax -> bmm(v, w)) runs as
 * asd_gemm_f16 -> asd_softmax_f16 -> asd_gemm_f16, and likewise backwards. */
int asd_softmax_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, float scale, void* y, int32_t ldy, void* stream);
int asd_softmax_bwd_f16(const void* p, const void* dp, int32_t ld, int32_t rows, int32_t cols, float scale, void* ds,
                        void* stream);
/* GEGLU (attention.py:49-56): y[r, j] = h[r, j] * gelu(h[r, c + j]), h = [rows, 2c]. */
int asd_geglu_f16(const void* h, int32_t rows, int32_t c, void* y, void* stream);
/* y = silu(x) elementwise (emb_layers, openaimodel.py:213-219). */
int asd_silu_f16(const void* x, int64_t n, void* y, void* stream);
/* sinusoidal timestep embedding [n, dim] fp16 (diffusionmodules/util.py:165-186). */
int asd_timestep_embedding_f16(const float* t, int32_t n, int32_t dim, void* y, void* stream);
/* channel concat of two NHWC tensors: y[r, :c1] = x1[r], y[r, c1:] = x2[r]. */
int asd_concat_f16(const void* x1, int32_t c1, const void* x2, int32_t c2, int64_t rows, void* y, void* stream);
/* Multi-head attention (CrossAttention.forward, attention.py:168-194), head_dim 64, flash-style:
 * q [batch, lq, heads*64] (row stride ldq), k [batch, lk_stride, heads*64] (ldk), vT [heads*64, batch*lk_stride]
 * (V stored transposed: row = channel, ld = ldv), o [batch, lq, heads*64] (ldo).  Keys >= lk of every batch
 * entry are masked; lk_stride (>= lk, multiple of 8) is the padded number of key rows per batch entry. */
int asd_attention_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vT, int32_t ldv,
                      void* o, int32_t ldo, int32_t batch, int32_t heads, int32_t lq, int32_t lk,
                      int32_t lk_stride, float scale, const void* zero_page, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM plan table + autotuner (csrc/gemm.hip).  asd_gemm_f16 with split_k == 0 ("auto") uses the tuned (tile, split-K) of the
 * problem's shape when one is recorded, else the built-in cost model.  Shape signature: conv == 0: s0 = lda (negated for the
 * GEGLU epilogue), s1..s4 = 0; conv == 1: s0..s4 = Hin, Cin, stride, upsample, pad.
 * ---------------------------------------------------------------------------------------------- */
int asd_gemm_plan_set(int32_t M, int32_t N, int32_t K, int32_t conv, int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t s4,
                      int32_t tile_cfg, int32_t split_k);
/* plan of args' shape -> (*tile_cfg, *split_k); returns 0 when tuned, 1 when the defaults (cost model, heuristic split) are given */
int asd_gemm_plan_get(const asd_gemm_args* args, int32_t* tile_cfg, int32_t* split_k);
int asd_gemm_plan_count(void);
/* changes whenever the plan table (or the forced tile) changes: lets a caller cache what it derived from the plans (workspace layouts) */
uint64_t asd_gemm_plan_generation(void);
int asd_gemm_plan_entry(int32_t i, int32_t* out11 /* {M,N,K,conv,s0..s4,tile_cfg,split_k} */);
/* bytes of split-K workspace asd_gemm_f16 needs for args (split_k == 0: under the current plan) */
int64_t asd_gemm_workspace_bytes(const asd_gemm_args* args);
/* time every valid (tile, split-K) candidate for args on `stream` (not capturing) and record the winner */
int asd_gemm_tune(const asd_gemm_args* args, void* scratch, int64_t scratch_bytes, void* stream);
/* y[rows, c_pad] fp16 = x[rows, c] fp32, zero padded */
int asd_pad_cast_f16(const float* x, int32_t rows, int32_t c, void* y, int32_t c_pad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Frozen-prior networks as C handles: the layer schedule is enqueued from C++ (csrc/net.hip), one chain of launches on the
 * caller's stream, no host synchronisation => capturable into a HIP graph from any host language.
 *   asd_unet_fwd     replaces forward_unet  (stable_diffusion_asd_guidance.py:319-331: diffusers UNet2DConditionModel) and
 *                    model.apply_model       (mvdream_asd_guidance.py:261-265: MultiViewUNetModel, openaimodel.py:1175-1213)
 *   asd_vae_enc_fwd  replaces encode_images (stable_diffusion_asd_guidance.py:171-178 / mvdream_asd_guidance.py:105-112: the
 *                    encoder + quant_conv up to the posterior moments; sampling and the 0.18215 scale stay in asd_latents_*)
 *   asd_vae_enc_bwd  its input gradient: the reference keeps the VAE in the autograd graph (:225), weights frozen (:101-102)
 * Protocol: create(desc) -> the handle publishes the PACKED weight table it reads (num_weights / weight_info: name, rows, cols
 * of a row-major fp16 matrix; scaledreamer_amd/diffusion/weights.py packs a name-keyed LDM state dict into it) -> bind_weights
 * (device pointers in table order; the caller keeps them alive) -> workspace_bytes(shape) -> fwd.  `tune` != 0: GEMM shapes
 * without a recorded plan are timed first (asd_gemm_tune; needs the larger workspace workspace_bytes(..., tune) reports; must
 * not be used under stream capture).
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_weight_info { const char* name; int32_t rows, cols; } asd_weight_info;

typedef struct asd_unet_desc {          /* extern/mvdream/configs/sd-v2-base.yaml:13-27 */
    int32_t in_channels, out_channels, model_channels, num_res_blocks;
    int32_t n_levels, channel_mult[8];
    int32_t attention_ds_mask;           /* bit k: transformer blocks at downsample factor 2^k (attention_resolutions [4,2,1] -> 0b111) */
    int32_t num_head_channels;           /* 64 */
    int32_t transformer_depth, context_dim;
    int32_t camera_dim;                  /* 0: SD-2.1; 16: MVDream (camera MLP added to the time embedding, self-attention over
                                            the num_frames views of a group) */
} asd_unet_desc;
typedef struct asd_unet asd_unet;
int asd_unet_create(const asd_unet_desc* desc, asd_unet** out);
void asd_unet_destroy(asd_unet* h);
int32_t asd_unet_num_weights(const asd_unet* h);
int asd_unet_weight_info(const asd_unet* h, int32_t i, asd_weight_info* info);
int asd_unet_bind_weights(asd_unet* h, const void* const* device_ptrs, int32_t count);
int64_t asd_unet_workspace_bytes(asd_unet* h, int32_t batch, int32_t H, int32_t W, int32_t n_ctx, int32_t num_frames, int32_t tune);
/* x_nhwc: fp16 [batch,H,W,32] (latent channels first, rest zero); t: fp32 [batch]; context: fp16 [batch*ctx_stride, context_dim]
 * with ctx_stride = n_ctx rounded up to 8, padding rows zero; camera: fp16 [batch,16] or NULL; eps_nhwc: fp32 [batch,H,W,out_channels] */
int asd_unet_fwd(asd_unet* h, const void* x_nhwc, const float* t, const void* context, const void* camera, int32_t batch, int32_t H,
                 int32_t W, int32_t n_ctx, int32_t num_frames, void* workspace, int64_t workspace_bytes, float* eps_nhwc, int32_t tune,
                 void* stream);

/* The same forward when several batch entries are known to carry the SAME (x, t, camera) and differ only in their text context —
 * the ASD step evaluates one noised latent under 4 prompts (CFG + Perp-Neg, stable_diffusion_asd_guidance.py:333-428): everything
 * in front of the first cross-attention (conv_in, the first ResBlock, proj_in and the first self-attention, the 4096-token one) is
 * computed once per distinct input and broadcast.  uniq_src_dev[n_uniq]: batch index of each distinct input (ascending; for the
 * camera-conditioned UNet whole groups of num_frames); expand_dev[batch]: index into that list for every batch entry.  Device
 * int32 arrays owned by the caller.  n_uniq == batch (or NULL arrays) is asd_unet_fwd. */
int64_t asd_unet_workspace_bytes_shared(asd_unet* h, int32_t batch, int32_t H, int32_t W, int32_t n_ctx, int32_t num_frames, int32_t tune,
                                        int32_t n_uniq);
int asd_unet_fwd_shared(asd_unet* h, const void* x_nhwc, const float* t, const void* context, const void* camera, int32_t batch, int32_t H,
                        int32_t W, int32_t n_ctx, int32_t num_frames, const int32_t* uniq_src_dev, const int32_t* expand_dev, int32_t n_uniq,
                        void* workspace, int64_t workspace_bytes, float* eps_nhwc, int32_t tune, void* stream);
/* dst[i][:] = src[idx_dev[i]][:], rows of row_halfs fp16 values (multiple of 8) */
int asd_gather_rows_f16(const void* src, const int32_t* idx_dev, int32_t n_out, int64_t row_halfs, void* dst, void* stream);

typedef struct asd_vae_desc {           /* first_stage_config.ddconfig of sd-v2-base.yaml:33-50 */
    int32_t in_channels, ch, n_levels, ch_mult[8], num_res_blocks, z_channels, embed_dim;
} asd_vae_desc;
typedef struct asd_vae_enc asd_vae_enc;
int asd_vae_enc_create(const asd_vae_desc* desc, asd_vae_enc** out);
void asd_vae_enc_destroy(asd_vae_enc* h);
int32_t asd_vae_enc_num_weights(const asd_vae_enc* h);
int asd_vae_enc_weight_info(const asd_vae_enc* h, int32_t i, asd_weight_info* info);
int asd_vae_enc_bind_weights(asd_vae_enc* h, const void* const* device_ptrs, int32_t count);
/* one workspace serves a forward and the backward that follows it (the forward leaves the activations the gradient needs there) */
int64_t asd_vae_enc_workspace_bytes(asd_vae_enc* h, int32_t batch, int32_t H, int32_t W, int32_t tune);
/* x_nhwc32: fp16 [batch,H,W,32] image in [-1,1], channels >= in_channels zero; moments_nhwc: fp32 [batch,H/8,W/8,2*embed_dim] */
int asd_vae_enc_fwd(asd_vae_enc* h, const void* x_nhwc32, int32_t batch, int32_t H, int32_t W, void* workspace, int64_t workspace_bytes,
                    float* moments_nhwc, int32_t tune, void* stream);
/* d_moments_nhwc: fp32 like moments; dx_nhwc32: fp16 [batch,H,W,32], channels < in_channels hold the image gradient */
int asd_vae_enc_bwd(asd_vae_enc* h, const float* d_moments_nhwc, int32_t batch, int32_t H, int32_t W, void* workspace,
                    int64_t workspace_bytes, void* dx_nhwc32, int32_t tune, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ASD latent-space glue (csrc/asd_glue.hip): everything between the rendered image and the scalar loss that is not a network.
 * Replaces the torch elementwise code of SDTimestepShiftedScoreDistillationGuidance.__call__ / get_latents / encode_images /
 * get_eps (threestudio/models/guidance/stable_diffusion_asd_guidance.py:171-178,196-209,242-283,404-428; perpendicular_component
 * threestudio/utils/ops.py:501-511) and of the MVDream variant (mvdream_asd_guidance.py:105-139,181-304).
 * ---------------------------------------------------------------------------------------------- */
/* x[b,Y,X,0:3] = 2 * bilinear(rgb)[b,Y,X,:] - 1 (F.interpolate(mode="bilinear", align_corners=False) semantics), x[...,3:32] = 0.
 * rgb: fp32 [B,h,w,3]; x_nhwc32: fp16 [B,H,W,32] = the VAE encoder's input layout. */
int asd_image_prep_fwd(const float* rgb, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, void* x_nhwc32, void* stream);
/* its adjoint: d_rgb fp32 [B,h,w,3] from dx fp16 [B,H,W,32] (channels 0..2) */
int asd_image_prep_bwd(const void* dx_nhwc32, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, float* d_rgb, void* stream);
/* latents = (mean + exp(0.5 clamp(logvar,-30,20)) * post_noise) * scaling (fp32 NCHW [B,C,hl,wl]); x_t = sqrt(a_t) z + sqrt(1-a_t) noise at
 * t and at t_plus written as fp16 NHWC-32 rows of the UNet input in the order [x_t] * n_rep, then [x_t+]; unet_t = [t] * n_rep + [t_plus].
 * moments_nhwc: fp32 [B,hl,wl,2C]; post_noise, noise: fp32 NCHW; t, t_plus: int64 [B]; alphas_cumprod: fp32 [>= max t + 1]. */
int asd_latents_fwd(const float* moments_nhwc, const float* post_noise, const float* noise, const int64_t* t, const int64_t* t_plus,
                    const float* alphas_cumprod, int32_t B, int32_t C, int32_t hl, int32_t wl, float scaling, int32_t n_rep,
                    float* latents, void* unet_x, float* unet_t, void* stream);
/* eps_nhwc: fp32 [(2 + n_neg + 1) * B, hw, C] in the order text | uncond | negatives (n_neg per sample, interleaved) | second.
 * eps_p = uncond + s * (pos + sum_k w_k perp(neg_k - uncond, pos)), pos = text - uncond; grad = nan_to_num(w(t) (eps_p - second)) clamped
 * to +-grad_clip when > 0; weighting 0: sds (1 - a_t), 1: uniform, 2: fantasia3d (sqrt(a_t) (1 - a_t)).
 * grad: fp32 NCHW [B,C,hw]; sumsq: fp32 [B] workspace; loss_and_norm: fp32 [2] = {0.5 * sum(grad^2) / B, ||grad||}. */
int asd_score_fwd(const float* eps_nhwc, int32_t B, int32_t C, int32_t hw, int32_t n_neg, const float* neg_w, float guidance_scale,
                  const int64_t* t, const float* alphas_cumprod, int32_t weighting, float grad_clip, float* grad, float* sumsq,
                  float* loss_and_norm, void* stream);
/* View-dependent prompt selection written into the UNet's context buffer (replaces prompt_processors/base.py:82-167
 * get_text_embeddings_perp_neg / :169-199 get_text_embeddings and the torch.cat of stable_diffusion_asd_guidance.py:377-394).
 * text_vd, uncond_vd: fp32 [n_dir, n_tok, dim] in the order side / front / back / overhead (n_dir = 4) or one embedding (n_dir = 1, no
 * view dependence); elevation, azimuth: fp32 [batch] degrees.  layout 0: context rows of [text | uncond | text] (3 * batch samples);
 * layout 1 (Perp-Neg): [pos | uncond | neg1_0, neg2_0, neg1_1, ... | pos] (5 * batch) and neg_w[batch, 2] = neg_scale * the reference's
 * weights.  params15 (host): overhead / front / back thresholds, then perp_neg_f_sb, f_fsb, f_fs, f_sf (3 floats each).
 * context_f16: fp16 [samples * ctx_stride, dim]; rows n_tok..ctx_stride-1 of a sample are not touched. */
int asd_prompt_context(const float* text_vd, const float* uncond_vd, int32_t n_dir, int32_t n_tok, int32_t dim, const float* elevation,
                       const float* azimuth, int32_t batch, int32_t layout, const float* params15, float neg_scale, void* context_f16,
                       int32_t ctx_stride, float* neg_w, void* stream);
/* d loss / d moments (fp32 [B,hl,wl,2C]) given grad: d z = upstream * grad / B (upstream: device scalar or NULL = 1) */
int asd_latents_bwd(const float* grad, const float* moments_nhwc, const float* post_noise, const float* upstream, int32_t B, int32_t C,
                    int32_t hl, int32_t wl, float scaling, float* d_moments_nhwc, void* stream);
/* t_plus = clamp(t + (int64)(u * clamp(plus_ratio * (t - min_step), 0, T - 1 - t)), 1, T - 1) in the reference's float32 arithmetic
 * (stable_diffusion_asd_guidance.py:294-316 get_t_plus); u: fp32 [n] uniform draws, or NULL (plus_random off: u = 1) */
int asd_timestep_plus(const int64_t* t, const float* u, int32_t n, int64_t min_step, int64_t num_train_timesteps, float plus_ratio,
                      int64_t* t_plus, void* stream);
/* loss assembly of training_step (scaledreamer.py:62-126) in one launch each way:
 *   total = sum_j weights[j] * terms[j][0] + lambda_sparsity * mean(sqrt(opacity^2 + 0.01))
 *         + lambda_opaque * mean(-(x log x + (1 - x) log(1 - x))), x = clamp(opacity, 1e-3, 1 - 1e-3)   (threestudio/utils/ops.py:365-369)
 *         + lambda_z_variance * mean(z_variance[opacity > 0.5])
 * terms: HOST array of n_terms (<= 8) device scalars, weights: HOST floats; opacity, z_variance: fp32 [n_rays]; a term whose lambda is
 * <= 0 is skipped (z_variance may then be NULL).  out5 (device) = {total, sparsity, opaque, z_variance, #rays above 0.5}.
 * bwd: upstream = d total (device scalar, NULL = 1); d_terms [n_terms], d_opacity [n_rays], d_z_variance [n_rays] or NULL. */
int asd_loss_tail_fwd(const float* const* terms, const float* weights, int32_t n_terms, const float* opacity, const float* z_variance,
                      int64_t n_rays, float lambda_sparsity, float lambda_opaque, float lambda_z_variance, float* out5, void* stream);
int asd_loss_tail_bwd(const float* upstream, const float* weights, int32_t n_terms, const float* opacity, int64_t n_rays, float lambda_sparsity,
                      float lambda_opaque, float lambda_z_variance, const float* out5, float* d_terms, float* d_opacity, float* d_z_variance,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step (csrc/optim.hip): multi-tensor fused fp32 kernels for the reference's two optimizers — torch.optim.AdamW / Adam as
 * parsed by threestudio/systems/utils.py:25-53 (per-group lr through param groups) and Adan, threestudio/systems/optimizers.py:200-315.
 * The caller owns parameters, gradients and state; bias corrections are passed per tensor (groups keep their own step counts).
 * ---------------------------------------------------------------------------------------------- */
#define ASD_OPT_MAX_TENSORS 24
typedef struct asd_opt_tensor {
    float* p; float* g; float* m; float* v;     /* parameter, gradient, exp_avg, exp_avg_sq (AdamW) | exp_avg_diff (Adan) */
    float* n2; float* prev;                      /* Adan only: exp_avg_sq, neg_pre_grad; NULL for AdamW */
    int64_t n;                                   /* elements */
    float lr, weight_decay;
    float bias_correction1;                      /* 1 - beta1^t */
    float bias_correction2;                      /* Adan: 1 - beta2^t */
    float bias_correction2_sqrt;                 /* AdamW: sqrt(1 - beta2^t); Adan: sqrt(1 - beta3^t) */
} asd_opt_tensor;
/* AdamW: p *= 1 - lr wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2; p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps).
 * adam_l2 != 0: Adam with L2 regularisation instead (g += wd p, no decoupled decay). */
int asd_adamw_f32(const asd_opt_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float eps, int32_t adam_l2, void* stream);
/* Adan step with the gradient scaled by `clip` first (global-norm clipping factor computed by the caller) */
int asd_adan_f32(const asd_opt_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float beta3, float eps, float clip,
                 int32_t no_prox, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generator backbone of the multi-prompt configs: the StyleGAN-3D synthesis network's 3x3x3 convolutions
 * (custom/amortized/extern/stylegan_3dconv_modules.py:64-82 modulated_conv3d -> F.conv3d(groups = batch), :117-171) with forward,
 * input gradient and weight gradient, and the layer tail around them (:56-62 SmoothUpsample, :301-313 noise + bias + lrelu + clamp).
 * Every volume is fp32 CHANNEL-LAST [N][D][H][W][C]; weights are fp32 in torch's layout [Cout][Cin][3][3][3], one set per sample
 * (stride w_sample_stride floats; 0 = shared).  Arithmetic: each fp32 operand is split into two fp16 planes and a product is three
 * fp16 MFMA products accumulated in fp32 (csrc/conv3d.hip) - ~22 significant bits per operand.  H, W multiples of 16 (pad smaller
 * volumes), Cin, Cout multiples of 64.  Workspace: asd_conv3d_workspace_bytes(desc, pass), pass 0 forward / 1 input gradient /
 * 2 weight gradient; the caller owns it.
 * ---------------------------------------------------------------------------------------------- */
#define ASD_CONV3D_WGRAD_MAX_SPLIT 64
typedef struct asd_conv3d_desc {
    int32_t N, D, H, W, Cin, Cout;
    /* optional: bit pattern of max|x| / max|dy| over the whole tensor when its producer already left it (asd_conv3d_epilogue.amax_out,
     * asd_layer_act_bwd); NULL = one extra pass over the tensor in here.  Only their binary exponent is used (the split scale). */
    const uint32_t* amax_x; const uint32_t* amax_dy;
} asd_conv3d_desc;
typedef struct asd_conv3d_epilogue {      /* y = act(conv + noise[voxel] * *noise_strength + bias[c]) */
    const float* bias;            /* [C] or NULL */
    const float* noise;           /* [N*D*H*W] or NULL */
    const float* noise_strength;  /* device scalar (the layer's parameter); required with noise */
    int32_t act;                  /* 0: none; 1: clamp(leaky_relu(v, 0.2) * gain, -clamp, clamp)  (clamp_gain, :5-7) */
    float gain, clamp;
    uint32_t* amax_out;           /* optional: receives max|y| as a bit pattern (atomicMax: zero it first) for the consumer's asd_conv3d_desc.amax_x */
} asd_conv3d_epilogue;
int64_t asd_conv3d_workspace_bytes(const asd_conv3d_desc* desc, int32_t pass);
int asd_conv3d_fwd(const asd_conv3d_desc* desc, const float* x, const float* w, int64_t w_sample_stride, float* y,
                   const asd_conv3d_epilogue* ep /* or NULL */, void* ws, int64_t ws_bytes, void* stream);
int asd_conv3d_dgrad(const asd_conv3d_desc* desc, const float* dy, const float* w, int64_t w_sample_stride, float* dx, void* ws,
                     int64_t ws_bytes, void* stream);
int asd_conv3d_wgrad(const asd_conv3d_desc* desc, const float* x, const float* dy, float* dw /* [N][Cout][Cin][27] */,
                     int64_t dw_sample_stride, void* ws, int64_t ws_bytes, void* zero_page /* >= 16 B of zeros */, void* stream);
/* gradient through the layer tail, read off the OUTPUT y (minus `sub` when the layer output was act(.) + sub): dz = dy * act'(y - sub) — or,
 * with act_mask (asd_upsample3d_fwd), off the branch bits the forward pass recorded (y may then be NULL);
 * d_bias[C] = sum over rows of dz (optional); d_rowsum[rows] = sum over channels of dz (optional: the gradient of the per-voxel noise
 * term); amax_out (optional, zeroed by the caller): max|dz| for asd_conv3d_desc.amax_dy */
int asd_layer_act_bwd(const float* dy, const float* y, const float* sub, const uint8_t* act_mask, int64_t rows, int32_t C, float gain, float clamp,
                      float* dz, float* d_bias, float* d_rowsum, uint32_t* amax_out, void* stream);
/* the per-sample weights of a modulated convolution (custom/amortized/extern/stylegan_3dconv_modules.py:64-82 modulated_conv3d):
 *   wm[n][co][ci][k] = weight[co][ci][k] * styles[n][ci] * gain * dcoef[n][co],
 *   dcoef = rsqrt(sum_{ci,k} (weight styles gain)^2 + 1e-8) when demodulate, 1 otherwise (toRGB: gain = its weight_gain).  1 <= N <= 8.
 * bwd: d_weight [Cout][Cin][K] and d_styles [N][Cin] from d_wm; K = 1 or 27. */
int asd_modulated_weights_fwd(const float* weight, const float* styles, int32_t N, int32_t Cout, int32_t Cin, int32_t K, float gain,
                              int32_t demodulate, float* wm, float* dcoef, void* stream);
int asd_modulated_weights_bwd(const float* d_wm, const float* wm, const float* weight, const float* styles, const float* dcoef, int32_t N,
                              int32_t Cout, int32_t Cin, int32_t K, float gain, int32_t demodulate, float* d_weight, float* d_styles,
                              void* stream);
/* *amax_out = max(*amax_out, bit pattern of max|x|) over n floats (n % 4 == 0): for tensors whose producer left no such word */
int asd_absmax_f32(const float* x, int64_t n, uint32_t* amax_out, void* stream);
/* toRGB (ToRGBLayer, :283-296: 1x1x1 modulated convolution onto the 32-channel skip volume) on channel-last rows, exact fp32:
 * y[rows][32] = x[rows][Cin] w[32][Cin]^T + bias + add (add optional: the upsampled skip volume); amax_out as above.
 * bwd: dx = dy w (+ dx_add), dw[32][Cin] = dy^T x, d_bias[32] = column sums of dy — each optional (NULL skips it). */
int asd_torgb_fwd(const float* x, int64_t rows, int32_t Cin, const float* w, const float* bias, const float* add, float* y,
                  uint32_t* amax_out, void* stream);
int asd_torgb_bwd(const float* x, const float* dy, int64_t rows, int32_t Cin, const float* w, const float* dx_add, float* dx, float* dw,
                  float* d_bias, void* stream);
/* y[N][2r][2r][2r][C] = act(trilinear_2x(x[N][r][r][r][C], align_corners) + noise * ns + bias) + add   (ep and add optional) */
int asd_upsample3d_fwd(const float* x, int32_t N, int32_t r, int32_t C, const asd_conv3d_epilogue* ep, const float* add, float* y,
                       uint8_t* act_mask /* optional, [N (2r)^3 C / 4] bytes: with ep->act, two bits per element (slope 1 | clamped) for
                                            asd_layer_act_bwd — exact where y = act(.) + add does not give the branch back */,
                       void* stream);
/* dx = transpose of the upsampling applied to dy; ws: 6 * N * r^3 * C floats */
int asd_upsample3d_bwd(const float* dy, int32_t N, int32_t r, int32_t C, float* dx, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tri-plane transformer generator (custom/amortized/extern/triplane_transformer_modules.py:34-187; the space generator of
 * `Triplane-transformer-sdf`, custom/amortized/models/geometry/triplane_transformer.py:97-116), trained in fp32 by the reference.
 * Everything below works on fp32 tensors at fp32-class accuracy: operands are split into two fp16 planes (x * 2^e = hi + lo, one
 * power-of-two scale per operand row) and a product is hi.hi + hi.lo + lo.hi on the fp16 matrix pipe, accumulated in fp32
 * (csrc/tritx.hip).  All tensors are row-major fp32 device memory owned by the caller; `ws` are caller-owned workspaces.
 * ---------------------------------------------------------------------------------------------- */
/* nn.Linear weight w [N, K] -> operand planes: plane_w [N, 3K] fp16 + inv_w [N] for y = x w^T, and / or plane_wt [K, 3 * round64(N)] fp16 +
 * inv_wt [K] for the input gradient dx = dy w (either pair may be NULL); ws: >= K floats */
int asd_tx_pack_weight(const float* w, int32_t N, int32_t K, void* plane_w, float* inv_w, void* plane_wt, float* inv_wt, float* ws, void* stream);
/* y [M, N] (ldy) = f(x [M, K] (ldx) . W^T + bias) + residual (ldr); W as packed by asd_tx_pack_weight (its plane_wt / inv_wt for an input
 * gradient: then `K` is the weight's N and must be a multiple of 64).  mode 0: f = identity; 1: f = GELU (erf form, nn.GELU()), the
 * pre-activation is saved to aux [M, N]; 2: f(v) = v * GELU'(aux) — the gradient through that GELU.  bias / residual may be NULL. */
int64_t asd_tx_linear_workspace(int32_t M, int32_t N, int32_t K);    /* floats */
int asd_tx_linear(const float* x, int32_t M, int32_t K, int32_t ldx, const void* plane_w, const float* inv_w, int32_t N, const float* bias, int32_t mode,
                  float* aux, const float* residual, int32_t ldr, float* y, int32_t ldy, float* ws, void* stream);
/* weight gradient of that layer: dw [N, K] = dy [M, N]^T . x [M, K], db [N] = column sums of dy (NULL: skipped) */
int64_t asd_tx_wgrad_workspace(int32_t M, int32_t N, int32_t K);     /* floats */
int asd_tx_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* dw, float* db, float* ws, void* stream);
/* nn.LayerNorm(D, eps) over rows (D % 4 == 0, D <= 1024); stats [M, 2] = (mean, rstd).  Backward: dx = input gradient (+ dres if given),
 * dgamma / dbeta += (the caller zeroes them once per backward pass) */
int asd_tx_layernorm_fwd(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y, float* stats, void* stream);
int asd_tx_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int32_t M, int32_t D, const float* dres, float* dx,
                         float* dgamma, float* dbeta, void* stream);

/* diffusers' Attention core as the transformer uses it (head dim 48): o [Lq, ldo] = softmax(q k^T / sqrt(48)) v per head, head h = columns
 * 48 h .. 48 h + 47 of q / k / v / o (strided views of a fused qkv matrix are fine: ld % 4 == 0).  lse2 [H, Lq] (base-2 log-sum-exp of the
 * scaled scores) is what asd_tx_attention_bwd needs besides q, k, v, o. */
int64_t asd_tx_attention_workspace(int32_t Lq, int32_t Lk, int32_t H);   /* floats; serves forward and backward */
int asd_tx_attention_fwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, int32_t Lq, int32_t Lk, int32_t H,
                         float* o, int32_t ldo, float* lse2, float* ws, void* stream);
/* (dq, dk, dv) from d_o (the gradient of o); o / lse2 as the forward left them.  Written, not accumulated. */
int asd_tx_attention_bwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* o, int32_t ldo,
                         const float* d_o, int32_t lddo, const float* lse2, int32_t Lq, int32_t Lk, int32_t H, float* dq, int32_t lddq, float* dk,
                         int32_t lddk, float* dv, int32_t lddv, float* ws, void* stream);

/* The generator as ONE call each way.  Parameter table `params` (device pointers, fp32), per layer l at 20 l + i:
 *   0 norm1.weight 1 norm1.bias 2 cross_attn.to_q.weight [D,D] 3 cross_attn.to_k.weight [D,Dc] 4 cross_attn.to_v.weight [D,Dc]
 *   5 cross_attn.to_out.0.weight [D,D] 6 cross_attn.to_out.0.bias 7 norm2.weight 8 norm2.bias 9 self_attn.to_q.weight 10 self_attn.to_k.weight
 *   11 self_attn.to_v.weight 12 self_attn.to_out.0.weight 13 self_attn.to_out.0.bias 14 norm3.weight 15 norm3.bias 16 mlp.0.weight [F,D]
 *   17 mlp.0.bias 18 mlp.3.weight [D,F] 19 mlp.3.bias;   then 20 L + {0 pos_embed [3 R R, D], 1 norm.weight, 2 norm.bias, 3 deconv.weight [D, C, 2, 2]}
 * (state-dict names of triplane_transformer_modules.py:115-187).  Gradient table `grads`, per layer at 17 l + i:
 *   0 norm1.w 1 norm1.b 2 cross to_q 3 cross to_k|to_v STACKED [2D,Dc] 4 cross to_out.w 5 cross to_out.b 6 norm2.w 7 norm2.b
 *   8 self to_q|to_k|to_v STACKED [3D,D] 9 self to_out.w 10 self to_out.b 11 norm3.w 12 norm3.b 13 mlp.0.w 14 mlp.0.b 15 mlp.3.w 16 mlp.3.b;
 *   then 17 L + {pos_embed, norm.w, norm.b, deconv.w} — every entry is WRITTEN (sum over the batch).
 * asd_tritx_pack splits every weight into its operand planes (call it when the weights have changed); asd_tritx_fwd maps
 * text_embed [batch, Tc, Dc] to channel-last planes [batch, 3, 2R, 2R, C] (the reference's [batch, 3, C, 2R, 2R] permuted) and keeps the
 * activations the backward needs in `save`; asd_tritx_bwd takes the gradient of those planes. */
typedef struct asd_tritx_desc {
    int32_t n_layers, dim /* D = heads * 48 */, heads, cond_dim /* Dc */, cond_tokens /* Tc */, hidden /* F */, low_res /* R: 3 R R tokens */,
            out_channels /* C */;
    float eps;
    int32_t grads_prezeroed;   /* asd_tritx_bwd: the caller has zeroed every LayerNorm / bias gradient buffer (they are accumulated into) */
} asd_tritx_desc;
int64_t asd_tritx_packed_floats(const asd_tritx_desc* desc);
int64_t asd_tritx_save_floats(const asd_tritx_desc* desc, int32_t batch);
int64_t asd_tritx_workspace_floats(const asd_tritx_desc* desc);
int asd_tritx_pack(const asd_tritx_desc* desc, const float* const* params, float* packed, void* stream);
int asd_tritx_fwd(const asd_tritx_desc* desc, const float* const* params, const float* packed, const float* text_embed, int32_t batch, float* planes_cl,
                  float* save, float* ws, void* stream);
int asd_tritx_bwd(const asd_tritx_desc* desc, const float* const* params, const float* packed, const float* text_embed, int32_t batch,
                  const float* d_planes_cl, const float* save, float* const* grads, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel exchange: what Lightning's DDP does for the reference (launch.py:233-240: the mean of every trainable gradient once per
 * optimizer step, one process per GPU).  A communicator per process over RCCL (xGMI inside a node), resolved at run time from the process
 * image (PyTorch's librccl.so) or dlopen("librccl.so"); rank 0 calls asd_comm_unique_id and hands the 128 bytes to the other ranks by
 * whatever channel the host has (torch.distributed store, MPI, a file), every rank then calls asd_comm_create on ITS device.
 * asd_allreduce_mean_f32 is in place, enqueued on the caller's stream, never synchronises the host.
 * ---------------------------------------------------------------------------------------------- */
typedef struct asd_comm asd_comm;
int asd_comm_unique_id(void* id128 /* [host] 128 bytes out */);
int asd_comm_create(const void* id128, int32_t rank, int32_t world, asd_comm** comm);
int asd_comm_destroy(asd_comm* comm);
int asd_allreduce_mean_f32(asd_comm* comm, float* buf, int64_t n, void* stream);

/* library info */
const char* asd_version(void);
const char* asd_last_error(void);
/* Measurement hook (bench.py's `roofline` leg): two hipEvent_t created by the caller (timing enabled).  While they are set, entry
 * points that launch more than one kernel record `start` / `stop` on their stream immediately around their DOMINANT kernel
 * (asd_field_bwd: field_bwd_sample_kernel, the gradient scatter), so its duration can be read with hipEventElapsedTime without a
 * profiler.  Pass NULL, NULL to clear.  Process-wide; not meant for concurrent callers. */
int asd_probe_events(void* start_event, void* stop_event);
/* one empty launch named asd_trace_mark_kernel on `stream`: brackets a region of a kernel trace */
int asd_probe_mark(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASD_HIP_H */
