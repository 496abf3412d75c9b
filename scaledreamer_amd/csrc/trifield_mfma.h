// trifield_mfma.h — layout of the prepared weights of the matrix-pipe tri-plane field (trifield_mfma.hip) and its host entry points
#pragma once
#include <hip/hip_runtime.h>
#include "trifield_common.h"

// fragment images of one head, in halves (each [blocks][k-steps][64 lanes][8]); hi / lo pairs of the split-fp16 weights
#define TFM_A1 (4 * 3 * 64 * 8)          // W1   [64 x 96]  natural k order (layer 1 reads the lookup)
#define TFM_A2 (4 * 2 * 64 * 8)          // W2   [64 x 64]  k in accumulator order
#define TFM_A1T (6 * 2 * 64 * 8)         // W1^T [96 x 64]  k in accumulator order (feature gradient)
#define TFM_OFF_A1H 0
#define TFM_OFF_A1L (TFM_OFF_A1H + TFM_A1)
#define TFM_OFF_A2H (TFM_OFF_A1L + TFM_A1)
#define TFM_OFF_A2L (TFM_OFF_A2H + TFM_A2)
#define TFM_FWD_HALVES (TFM_OFF_A2L + TFM_A2)                 // 20480: what the forward chain keeps in LDS per head
#define TFM_OFF_A2TH TFM_FWD_HALVES
#define TFM_OFF_A2TL (TFM_OFF_A2TH + TFM_A2)
#define TFM_OFF_A1TH (TFM_OFF_A2TL + TFM_A2)
#define TFM_OFF_A1TL (TFM_OFF_A1TH + TFM_A1T)
#define TFM_HEAD_HALVES (TFM_OFF_A1TL + TFM_A1T)              // 40960
// prep buffer (floats): [0] bits of max|planes| | [16 + 16 head + TFM_S_*] scales | [64 ..] two head images | [TFM_PREP_FIXED ..] padded planes
#define TFM_PREP_FIXED (64 + 2 * TFM_HEAD_HALVES / 2)
// ... followed by the planes with a one-texel ZERO border, [3][H + 2][W + 2][32]: the lookup reads its four taps without range checks
static inline long long tfm_prep_floats(int H, int W) { return TFM_PREP_FIXED + (long long)3 * (H + 2) * (W + 2) * 32; }
enum { TFM_S_E = 0, TFM_S_W1, TFM_S_W2, TFM_S_H1, TFM_S_V2, TFM_S_U1, TFM_S_H2 };

int tfm_prepare(const float* planes_cl, int H, int W, const float* const* w6, float* prep, hipStream_t s);
int tfm_forward(const tf_geom g, const asd_field_cfg* cfg, const float* planes_cl, const float* const* w6, const float* prep, const float* points, int n, float* sdf,
                float* features, float* normal, float* fd_grad, hipStream_t s);
int tfm_backward_chunk(const tf_geom g, const asd_field_cfg* cfg, const float* planes_cl, const float* const* w6, const float* prep, const float* points,
                       const float* sdf, int i0, int nc, int npt, const float* d_sdf, const float* d_features, const float* d_normal, const float* d_fd_grad,
                       float* denc, float* pts, float* const* dw6, hipStream_t s);
// counting sort of the feature-gradient rows by plane cell + segmented reduction (trifield_scatter.hip)
int64_t tfs_work_ints(int rows, int H, int W);
bool tfs_supported(int H, int W);
int tfs_scatter(const float* denc, const float* pts, int R, int H, int W, float* d_planes, int* work, hipStream_t s);
