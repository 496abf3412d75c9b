/*
 * asd_oracle.c — CPU restatement (plain C, fp32) of the renderer half of ScaleDreamer's ASD step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scaledreamer_amd/ may import, link or call this file; it is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / CPU baseline.
 *
 * Parity status: the arithmetic restated here lives in un-vendored third-party packages of the reference
 * (tiny-cuda-nn @ unpinned master, nerfacc v0.5.2 — reference README.md:65-66).  The reference holds no
 * tests or golden vectors for them, so:
 *   - hash grid: pinned by the known-answer tests of SURVEY.md §8c (parameter count 12 599 920,
 *     linear-field reproduction on dense levels, hashed index of (1,1,1)) and by running the reference's
 *     own Python glue (ImplicitVolume / NeRFVolumeRenderer) on top of it (tests/golden/make_goldens.py);
 *   - marching sample placement: "parity unpinned" (nerfacc's lattice phase is not reproducible here);
 *     the convention is stated in include/asd_hip.h and tested through invariants;
 *   - compositing: closed forms (constant sigma) + the reference glue run.
 *
 * Each function cites the reference call site it follows.  All float arithmetic that feeds a discrete
 * decision is written with explicit fmaf()/no contraction so the HIP kernels can match it bit for bit
 * (build with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/asd_hip.h"

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------------------------------- */
/* hash grid — tcnn GridEncoding as configured in configs/single-prompt_benchmark/asd_sd_nerf.yaml:47-53
 * (call site threestudio/models/networks.py:55-64); spec SURVEY.md Appendix B.1                      */
/* ---------------------------------------------------------------------------------------------- */
ORC_API uint32_t orc_grid_meta_init(asd_grid_meta* m, uint32_t n_levels, uint32_t n_features,
                                    uint32_t log2_hashmap_size, uint32_t base_resolution,
                                    double per_level_scale) {
    memset(m, 0, sizeof(*m));
    m->n_levels = n_levels;
    m->n_features = n_features;
    uint32_t offset = 0;
    const float log2_scale = log2f((float)per_level_scale);
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float scale = exp2f((float)l * log2_scale) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        uint64_t dense = (uint64_t)res * res * res;
        uint64_t size = (dense + 7u) / 8u * 8u;
        const uint64_t cap = 1ull << log2_hashmap_size;
        if (size > cap) size = cap;
        m->scale[l] = scale;
        m->resolution[l] = res;
        m->offset[l] = offset;
        m->size[l] = (uint32_t)size;
        m->dense[l] = dense <= size ? 1u : 0u;
        offset += (uint32_t)size;
    }
    m->n_params = offset * n_features;
    return m->n_params;
}

static inline uint32_t orc_grid_index(const asd_grid_meta* m, uint32_t l, uint32_t cx, uint32_t cy, uint32_t cz) {
    const uint32_t res = m->resolution[l];
    uint32_t idx;
    if (m->dense[l]) {
        idx = cx + cy * res + cz * res * res; /* tcnn: index % hashmap_size (x == 1.0 wraps) */
    } else {
        idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
    }
    return idx % m->size[l];
}

/* position -> (cell, fractional weight) for one level */
static inline void orc_pos_fract(float x, float scale, uint32_t* cell, float* w) {
    /* out-of-range rule (SURVEY.md B.1 leaves it to the build): clamp to the unit cube */
    x = fminf(fmaxf(x, 0.f), 1.f);
    const float p = fmaf(scale, x, 0.5f);
    const float f = floorf(p);
    *cell = (uint32_t)(int32_t)f;
    *w = p - f;
}

static void orc_encode_point(const asd_grid_meta* m, const float* params, const float x[3], float* out) {
    for (uint32_t l = 0; l < m->n_levels; ++l) {
        uint32_t c[3];
        float w[3];
        for (int d = 0; d < 3; ++d) orc_pos_fract(x[d], m->scale[l], &c[d], &w[d]);
        float f0 = 0.f, f1 = 0.f;
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const uint32_t bx = corner & 1u, by = (corner >> 1) & 1u, bz = (corner >> 2) & 1u;
            const float wx = bx ? w[0] : 1.f - w[0];
            const float wy = by ? w[1] : 1.f - w[1];
            const float wz = bz ? w[2] : 1.f - w[2];
            const float wt = wx * wy * wz;
            const uint32_t idx = orc_grid_index(m, l, c[0] + bx, c[1] + by, c[2] + bz);
            const float* e = params + 2u * (size_t)(m->offset[l] + idx);
            f0 = fmaf(wt, e[0], f0);
            f1 = fmaf(wt, e[1], f1);
        }
        out[2 * l + 0] = f0;
        out[2 * l + 1] = f1;
    }
}

static void orc_scatter_point(const asd_grid_meta* m, const float x[3], const float* dout, float* dparams) {
    for (uint32_t l = 0; l < m->n_levels; ++l) {
        uint32_t c[3];
        float w[3];
        for (int d = 0; d < 3; ++d) orc_pos_fract(x[d], m->scale[l], &c[d], &w[d]);
        for (uint32_t corner = 0; corner < 8; ++corner) {
            const uint32_t bx = corner & 1u, by = (corner >> 1) & 1u, bz = (corner >> 2) & 1u;
            const float wx = bx ? w[0] : 1.f - w[0];
            const float wy = by ? w[1] : 1.f - w[1];
            const float wz = bz ? w[2] : 1.f - w[2];
            const float wt = wx * wy * wz;
            const uint32_t idx = orc_grid_index(m, l, c[0] + bx, c[1] + by, c[2] + bz);
            float* e = dparams + 2u * (size_t)(m->offset[l] + idx);
            e[0] += wt * dout[2 * l + 0];
            e[1] += wt * dout[2 * l + 1];
        }
    }
}

ORC_API void orc_hashgrid_fwd(const asd_grid_meta* m, const float* params, const float* x, int32_t n, float* out) {
    const int nf = (int)(m->n_levels * 2);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) orc_encode_point(m, params, x + 3 * (size_t)i, out + (size_t)nf * i);
}

ORC_API void orc_hashgrid_bwd(const asd_grid_meta* m, const float* x, const float* dout, int32_t n, float* dparams) {
    const int nf = (int)(m->n_levels * 2);
    for (int32_t i = 0; i < n; ++i) orc_scatter_point(m, x + 3 * (size_t)i, dout + (size_t)nf * i, dparams);
}

/* ---------------------------------------------------------------------------------------------- */
/* field — ImplicitVolume.forward / forward_density / get_activated_density
 * (threestudio/models/geometry/implicit_volume.py:80-107,109-196,198-207), contract_to_unisphere
 * (geometry/base.py:20-32, bounded branch = scale_tensor, utils/ops.py:27-38), VanillaMLP
 * (models/networks.py:214-251: Linear(no bias) -> ReLU -> Linear(no bias))                           */
/* ---------------------------------------------------------------------------------------------- */
static inline float orc_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
static inline float orc_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

static inline float orc_density_bias(const asd_field_cfg* c, const float p[3]) {
    const float r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    if (c->bias_mode == ASD_BIAS_BLOB_MAGIC3D) return c->blob_scale * (1.f - sqrtf(r2) / c->blob_std);
    if (c->bias_mode == ASD_BIAS_BLOB_DREAMFUSION) return c->blob_scale * expf(-0.5f * r2 / (c->blob_std * c->blob_std));
    if (c->bias_mode == ASD_BIAS_SPHERE) return sqrtf(r2) - c->bias_value; /* get_shifted_sdf "sphere", hyper_iNGP.py:222-225 */
    return c->bias_value;
}

static inline float orc_activate(const asd_field_cfg* c, float raw) {
    switch (c->activation) {
        case ASD_ACT_SOFTPLUS: return orc_softplus(raw);
        case ASD_ACT_EXP:
        case ASD_ACT_TRUNC_EXP: return expf(raw);
        default: return raw;
    }
}
/* d activation / d raw */
static inline float orc_activate_grad(const asd_field_cfg* c, float raw) {
    switch (c->activation) {
        case ASD_ACT_SOFTPLUS: return raw > 20.f ? 1.f : orc_sigmoid(raw);
        case ASD_ACT_EXP: return expf(raw);
        case ASD_ACT_TRUNC_EXP: return expf(fminf(raw, 15.f));
        default: return 1.f;
    }
}

static inline void orc_contract(const asd_field_cfg* c, const float p[3], float x[3]) {
    for (int d = 0; d < 3; ++d) x[d] = (p[d] - c->bbox_min[d]) / (c->bbox_max[d] - c->bbox_min[d]);
}

/* hidden[h] = relu(sum_k w1[h][k] enc[k]); out[o] = sum_h w2[o][h] hidden[h] */
static void orc_mlp(const float* w1, const float* w2, int n_in, int n_hidden, int n_out, const float* enc,
                    float* hidden, float* out) {
    for (int h = 0; h < n_hidden; ++h) {
        float a = 0.f;
        for (int k = 0; k < n_in; ++k) a = fmaf(w1[h * n_in + k], enc[k], a);
        hidden[h] = a > 0.f ? a : 0.f;
    }
    for (int o = 0; o < n_out; ++o) {
        float a = 0.f;
        for (int h = 0; h < n_hidden; ++h) a = fmaf(w2[o * n_hidden + h], hidden[h], a);
        out[o] = a;
    }
}

/* raw (pre-activation incl. bias) density at world point p; optionally returns enc/hidden */
static float orc_point_raw(const asd_grid_meta* m, const asd_field_cfg* c, const float* grid, const float* w1d,
                           const float* w2d, const float p[3], float* enc, float* hidden) {
    float x[3];
    orc_contract(c, p, x);
    orc_encode_point(m, grid, x, enc);
    float o;
    orc_mlp(w1d, w2d, (int)m->n_levels * 2, c->n_hidden, 1, enc, hidden, &o);
    return o + orc_density_bias(c, p);
}

ORC_API void orc_field_density(const asd_grid_meta* m, const asd_field_cfg* c, const float* grid, const float* w1d,
                               const float* w2d, const float* points, int32_t n, float* sigma) {
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        float enc[2 * ASD_MAX_LEVELS], hid[256];
        sigma[i] = orc_activate(c, orc_point_raw(m, c, grid, w1d, w2d, points + 3 * (size_t)i, enc, hid));
    }
}

static inline float orc_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

ORC_API void orc_field_fwd(const asd_grid_meta* m, const asd_field_cfg* c, const float* grid, const float* w1d,
                           const float* w2d, const float* w1f, const float* w2f, const float* points, int32_t n,
                           float* sigma, float* features, float* normal, float* fd_grad, float* enc_save) {
    const int nin = (int)m->n_levels * 2;
    /* density field: normal = -grad sigma (implicit_volume.py:177); sdf field: sdf_grad = +grad (hyper_iNGP.py:316) */
    const float fd_sign = c->field_mode == ASD_FIELD_SDF ? 1.f : -1.f;
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        const float* p = points + 3 * (size_t)i;
        float enc[2 * ASD_MAX_LEVELS], hid[256];
        const float raw = orc_point_raw(m, c, grid, w1d, w2d, p, enc, hid);
        const float s = orc_activate(c, raw);
        sigma[i] = s;
        if (enc_save) memcpy(enc_save + (size_t)nin * i, enc, sizeof(float) * nin);
        if (features && c->n_feature_dims > 0)
            orc_mlp(w1f, w2f, nin, c->n_hidden, c->n_feature_dims, enc, hid, features + (size_t)c->n_feature_dims * i);
        if (normal || fd_grad) {
            /* finite_difference branch, implicit_volume.py:162-177 / hyper_iNGP.py:303-318 */
            float nr[3];
            for (int k = 0; k < 3; ++k) {
                float q[3] = {p[0], p[1], p[2]};
                q[k] = orc_clampf(q[k] + c->fd_eps, -c->radius, c->radius);
                for (int d = 0; d < 3; ++d)
                    if (d != k) q[d] = orc_clampf(q[d], -c->radius, c->radius);
                float enc2[2 * ASD_MAX_LEVELS];
                const float sk = orc_activate(c, orc_point_raw(m, c, grid, w1d, w2d, q, enc2, hid));
                nr[k] = fd_sign * (sk - s) / c->fd_eps;
            }
            if (fd_grad)
                for (int k = 0; k < 3; ++k) fd_grad[3 * (size_t)i + k] = nr[k];
            if (normal) {
                const float len = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
                const float inv = 1.f / fmaxf(len, 1e-12f); /* F.normalize eps */
                for (int k = 0; k < 3; ++k) normal[3 * (size_t)i + k] = nr[k] * inv;
            }
        }
    }
}

/* backward of one density-MLP evaluation: given dL/draw, accumulate dW1/dW2 and scatter into the grid */
static void orc_density_point_bwd(const asd_grid_meta* m, const asd_field_cfg* c, const float* w1d, const float* w2d,
                                  const float x01[3], const float* enc, float draw, float* dgrid, float* dw1d,
                                  float* dw2d) {
    const int nin = (int)m->n_levels * 2, H = c->n_hidden;
    float denc[2 * ASD_MAX_LEVELS];
    memset(denc, 0, sizeof(denc));
    for (int h = 0; h < H; ++h) {
        float a = 0.f;
        for (int k = 0; k < nin; ++k) a = fmaf(w1d[h * nin + k], enc[k], a);
        const float hv = a > 0.f ? a : 0.f;
        dw2d[h] += draw * hv;
        if (a > 0.f) {
            const float da = draw * w2d[h];
            for (int k = 0; k < nin; ++k) {
                dw1d[h * nin + k] += da * enc[k];
                denc[k] = fmaf(da, w1d[h * nin + k], denc[k]);
            }
        }
    }
    orc_scatter_point(m, x01, denc, dgrid);
}

ORC_API void orc_field_bwd(const asd_grid_meta* m, const asd_field_cfg* c, const float* grid, const float* w1d,
                           const float* w2d, const float* w1f, const float* w2f, const float* points, int32_t n,
                           const float* d_sigma, const float* d_features, const float* d_normal, const float* d_fd_grad,
                           float* dgrid, float* dw1d, float* dw2d, float* dw1f, float* dw2f) {
    const int nin = (int)m->n_levels * 2, H = c->n_hidden, C = c->n_feature_dims;
    const float fd_sign = c->field_mode == ASD_FIELD_SDF ? 1.f : -1.f;
    for (int32_t i = 0; i < n; ++i) {
        const float* p = points + 3 * (size_t)i;
        float x[3], enc[2 * ASD_MAX_LEVELS], hid[256];
        orc_contract(c, p, x);
        const float raw = orc_point_raw(m, c, grid, w1d, w2d, p, enc, hid);
        const float s = orc_activate(c, raw);
        float ds = d_sigma ? d_sigma[i] : 0.f;
        if (d_normal || d_fd_grad) {
            float q[3][3], sk[3], rawk[3], nr[3];
            for (int k = 0; k < 3; ++k) {
                for (int d = 0; d < 3; ++d) q[k][d] = orc_clampf(p[d] + (d == k ? c->fd_eps : 0.f), -c->radius, c->radius);
                float e2[2 * ASD_MAX_LEVELS];
                rawk[k] = orc_point_raw(m, c, grid, w1d, w2d, q[k], e2, hid);
                sk[k] = orc_activate(c, rawk[k]);
                nr[k] = fd_sign * (sk[k] - s) / c->fd_eps;
            }
            float dnr[3] = {0, 0, 0};
            if (d_normal) {
                const float len = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
                const float* dn = d_normal + 3 * (size_t)i;
                if (len > 1e-12f) {
                    const float inv = 1.f / len;
                    const float nh[3] = {nr[0] * inv, nr[1] * inv, nr[2] * inv};
                    const float dot = nh[0] * dn[0] + nh[1] * dn[1] + nh[2] * dn[2];
                    for (int k = 0; k < 3; ++k) dnr[k] = (dn[k] - nh[k] * dot) * inv;
                } else {
                    for (int k = 0; k < 3; ++k) dnr[k] = dn[k] * 1e12f;
                }
            }
            if (d_fd_grad)
                for (int k = 0; k < 3; ++k) dnr[k] += d_fd_grad[3 * (size_t)i + k];
            for (int k = 0; k < 3; ++k) {
                const float dsk = fd_sign * dnr[k] / c->fd_eps;
                ds -= fd_sign * dnr[k] / c->fd_eps;
                float xk[3], e2[2 * ASD_MAX_LEVELS];
                orc_contract(c, q[k], xk);
                orc_encode_point(m, grid, xk, e2);
                orc_density_point_bwd(m, c, w1d, w2d, xk, e2, dsk * orc_activate_grad(c, rawk[k]), dgrid, dw1d, dw2d);
            }
        }
        /* centre point: density MLP */
        float denc[2 * ASD_MAX_LEVELS];
        memset(denc, 0, sizeof(denc));
        const float draw = ds * orc_activate_grad(c, raw);
        for (int h = 0; h < H; ++h) {
            float a = 0.f;
            for (int k = 0; k < nin; ++k) a = fmaf(w1d[h * nin + k], enc[k], a);
            const float hv = a > 0.f ? a : 0.f;
            dw2d[h] += draw * hv;
            if (a > 0.f) {
                const float da = draw * w2d[h];
                for (int k = 0; k < nin; ++k) {
                    dw1d[h * nin + k] += da * enc[k];
                    denc[k] = fmaf(da, w1d[h * nin + k], denc[k]);
                }
            }
        }
        /* feature MLP */
        if (d_features && C > 0) {
            const float* df = d_features + (size_t)C * i;
            for (int h = 0; h < H; ++h) {
                float a = 0.f;
                for (int k = 0; k < nin; ++k) a = fmaf(w1f[h * nin + k], enc[k], a);
                const float hv = a > 0.f ? a : 0.f;
                float dh = 0.f;
                for (int o = 0; o < C; ++o) {
                    dw2f[o * H + h] += df[o] * hv;
                    dh = fmaf(df[o], w2f[o * H + h], dh);
                }
                if (a > 0.f) {
                    for (int k = 0; k < nin; ++k) {
                        dw1f[h * nin + k] += dh * enc[k];
                        denc[k] = fmaf(dh, w1f[h * nin + k], denc[k]);
                    }
                }
            }
        }
        orc_scatter_point(m, x, denc, dgrid);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* background — NeuralEnvironmentMapBackground.forward
 * (threestudio/models/background/neural_environment_map_background.py:46-67) without the random-colour
 * augmentation (host logic): (d+1)/2 -> HashGrid -> Linear-ReLU-Linear-ReLU-Linear -> sigmoid         */
/* ---------------------------------------------------------------------------------------------- */
ORC_API void orc_envmap_fwd(const asd_grid_meta* m, const float* grid, const float* w0, const float* w1,
                            const float* w2, int32_t H, const float* dirs, int32_t n, float* color) {
    const int nin = (int)m->n_levels * 2;
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        float x[3], enc[2 * ASD_MAX_LEVELS], h0[64], h1[64];
        for (int d = 0; d < 3; ++d) x[d] = (dirs[3 * (size_t)i + d] + 1.f) / 2.f;
        orc_encode_point(m, grid, x, enc);
        for (int h = 0; h < H; ++h) {
            float a = 0.f;
            for (int k = 0; k < nin; ++k) a = fmaf(w0[h * nin + k], enc[k], a);
            h0[h] = a > 0.f ? a : 0.f;
        }
        for (int h = 0; h < H; ++h) {
            float a = 0.f;
            for (int k = 0; k < H; ++k) a = fmaf(w1[h * H + k], h0[k], a);
            h1[h] = a > 0.f ? a : 0.f;
        }
        for (int o = 0; o < 3; ++o) {
            float a = 0.f;
            for (int k = 0; k < H; ++k) a = fmaf(w2[o * H + k], h1[k], a);
            color[3 * (size_t)i + o] = orc_sigmoid(a);
        }
    }
}

ORC_API void orc_envmap_bwd(const asd_grid_meta* m, const float* grid, const float* w0, const float* w1,
                            const float* w2, int32_t H, const float* dirs, const float* d_color, int32_t n,
                            float* dgrid, float* dw0, float* dw1, float* dw2) {
    const int nin = (int)m->n_levels * 2;
    for (int32_t i = 0; i < n; ++i) {
        float x[3], enc[2 * ASD_MAX_LEVELS], a0[64], a1[64], h0[64], h1[64], dout[3], dh1[64], dh0[64];
        float denc[2 * ASD_MAX_LEVELS];
        for (int d = 0; d < 3; ++d) x[d] = (dirs[3 * (size_t)i + d] + 1.f) / 2.f;
        orc_encode_point(m, grid, x, enc);
        for (int h = 0; h < H; ++h) {
            float a = 0.f;
            for (int k = 0; k < nin; ++k) a = fmaf(w0[h * nin + k], enc[k], a);
            a0[h] = a;
            h0[h] = a > 0.f ? a : 0.f;
        }
        for (int h = 0; h < H; ++h) {
            float a = 0.f;
            for (int k = 0; k < H; ++k) a = fmaf(w1[h * H + k], h0[k], a);
            a1[h] = a;
            h1[h] = a > 0.f ? a : 0.f;
        }
        for (int o = 0; o < 3; ++o) {
            float a = 0.f;
            for (int k = 0; k < H; ++k) a = fmaf(w2[o * H + k], h1[k], a);
            const float s = orc_sigmoid(a);
            dout[o] = d_color[3 * (size_t)i + o] * s * (1.f - s);
        }
        for (int k = 0; k < H; ++k) {
            float acc = 0.f;
            for (int o = 0; o < 3; ++o) {
                dw2[o * H + k] += dout[o] * h1[k];
                acc = fmaf(dout[o], w2[o * H + k], acc);
            }
            dh1[k] = a1[k] > 0.f ? acc : 0.f;
        }
        for (int k = 0; k < H; ++k) {
            float acc = 0.f;
            for (int h = 0; h < H; ++h) {
                dw1[h * H + k] += dh1[h] * h0[k];
                acc = fmaf(dh1[h], w1[h * H + k], acc);
            }
            dh0[k] = a0[k] > 0.f ? acc : 0.f;
        }
        for (int k = 0; k < nin; ++k) {
            float acc = 0.f;
            for (int h = 0; h < H; ++h) {
                dw0[h * nin + k] += dh0[h] * enc[k];
                acc = fmaf(dh0[h], w0[h * nin + k], acc);
            }
            denc[k] = acc;
        }
        orc_scatter_point(m, x, denc, dgrid);
    }
}


/* ---------------------------------------------------------------------------------------------- */
/* importance sampling — ImportanceEstimator.sampling (threestudio/models/estimators.py:23-101) calls the
 * un-vendored nerfacc v0.5.2: pdf.importance_sampling (:72-74,88), volrend.render_transmittance_from_density
 * (:84).  Restated from nerfacc's published behaviour (inverse-transform sampling of interval edges with
 * one jitter per ray, linear interpolation inside a cdf segment); the Philox jitter stream itself is not
 * reproducible here, so jitter is an input ("parity unpinned" for sample placement, as for the marcher).
 * Pinned through the reference's own glue run on top of these functions (tests/golden/make_goldens_amortized.py). */
/* ---------------------------------------------------------------------------------------------- */
/* out[r, j], j = 0..n_out: edge at cdf value u_j = (j + jitter[r]) / (n_out + 1)  (stratified)  or  j / n_out */
ORC_API void orc_importance_resample(const float* vals, const float* cdfs, int32_t n_rays, int32_t e_in,
                                     int32_t n_out, const float* jitter, float* out) {
#pragma omp parallel for schedule(static)
    for (int32_t r = 0; r < n_rays; ++r) {
        const float* v = vals + (size_t)r * e_in;
        const float* c = cdfs + (size_t)r * e_in;
        int p = 0;
        for (int j = 0; j <= n_out; ++j) {
            const float u = jitter ? ((float)j + jitter[r]) / (float)(n_out + 1) : (float)j / (float)n_out;
            while (p < e_in - 2 && c[p + 1] <= u) ++p; /* last p with c[p] <= u (searchsorted side="right") */
            const float c0 = c[p], c1 = c[p + 1];
            const float w = c1 > c0 ? fminf(fmaxf((u - c0) / (c1 - c0), 0.f), 1.f) : 0.f;
            out[(size_t)r * (n_out + 1) + j] = fmaf(w, v[p + 1] - v[p], v[p]);
        }
    }
}

/* cdf[r, j] = 1 - T_j, T_j = exp(-sum_{k<j} sigma_k (t_{k+1} - t_k)), cdf[r, S] = 1   (estimators.py:84-86) */
ORC_API void orc_transmittance_cdf(const float* t_edges, const float* sigma, int32_t n_rays, int32_t S, float* cdf) {
#pragma omp parallel for schedule(static)
    for (int32_t r = 0; r < n_rays; ++r) {
        const float* t = t_edges + (size_t)r * (S + 1);
        float acc = 0.f;
        for (int j = 0; j < S; ++j) {
            cdf[(size_t)r * (S + 1) + j] = 1.f - expf(-acc);
            acc = fmaf(sigma[(size_t)r * S + j], t[j + 1] - t[j], acc);
        }
        cdf[(size_t)r * (S + 1) + S] = 1.f;
    }
}

/* per-ray merge of two sorted edge lists = torch.sort(torch.cat([a, b], -1))  (estimators.py:93-94); ties: a first */
ORC_API void orc_merge_sorted(const float* a, int32_t na, const float* b, int32_t nb, int32_t n_rays, float* out) {
#pragma omp parallel for schedule(static)
    for (int32_t r = 0; r < n_rays; ++r) {
        const float* x = a + (size_t)r * na;
        const float* y = b + (size_t)r * nb;
        float* o = out + (size_t)r * (na + nb);
        int i = 0, j = 0;
        while (i < na || j < nb) {
            if (j >= nb || (i < na && x[i] <= y[j])) { *o++ = x[i++]; } else { *o++ = y[j++]; }
        }
    }
}

/* volsdf_density (threestudio/models/renderers/neus_volume_renderer.py:19-23) */
ORC_API void orc_volsdf_density(const float* sdf, int64_t n, float inv_std, float* sigma) {
    const float a = fminf(fmaxf(inv_std, 0.f), 80.f), beta = 1.f / a;
    for (int64_t i = 0; i < n; ++i) {
        const float s = sdf[i];
        const float sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
        sigma[i] = a * (0.5f + 0.5f * sg * expm1f(-fabsf(s) / beta));
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* voxel / tri-plane samplers — get_trilinear_feature and sample_from_planes
 * (custom/amortized/models/geometry/utils.py:67-110): F.grid_sample(mode="bilinear", padding_mode="zeros",
 * align_corners=False).  Channel-LAST feature storage (the HIP kernels' layout); pinned against the
 * reference functions themselves (tests/golden/amortized_samplers.npz).                              */
/* ---------------------------------------------------------------------------------------------- */
static inline void orc_gs_axis(float x, int size, int* i0, float* w1) {
    const float ix = ((x + 1.f) * (float)size - 1.f) * 0.5f; /* unnormalise, align_corners=False */
    const float f = floorf(ix);
    *i0 = (int)f;
    *w1 = ix - f;
}

/* voxel_cl [B, D, H, W, C]; points [B, M, 3] (x -> W, y -> H, z -> D); out [B, M, C] */
ORC_API void orc_voxel_sample_fwd(const float* voxel_cl, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                  const float* points, int32_t M, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)B * M; ++q) {
        const int b = (int)(q / M);
        const float* p = points + 3 * q;
        int x0, y0, z0;
        float fx, fy, fz;
        orc_gs_axis(p[0], W, &x0, &fx);
        orc_gs_axis(p[1], H, &y0, &fy);
        orc_gs_axis(p[2], D, &z0, &fz);
        float* o = out + q * C;
        for (int c = 0; c < C; ++c) o[c] = 0.f;
        for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
            if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
            const float* v = voxel_cl + ((((int64_t)b * D + z) * H + y) * W + x) * C;
            for (int c = 0; c < C; ++c) o[c] = fmaf(w, v[c], o[c]);
        }
    }
}

ORC_API void orc_voxel_sample_bwd(const float* d_out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                  const float* points, int32_t M, float* d_voxel_cl) {
    for (int64_t q = 0; q < (int64_t)B * M; ++q) {
        const int b = (int)(q / M);
        const float* p = points + 3 * q;
        int x0, y0, z0;
        float fx, fy, fz;
        orc_gs_axis(p[0], W, &x0, &fx);
        orc_gs_axis(p[1], H, &y0, &fy);
        orc_gs_axis(p[2], D, &z0, &fz);
        const float* g = d_out + q * C;
        for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
            if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
            float* v = d_voxel_cl + ((((int64_t)b * D + z) * H + y) * W + x) * C;
            for (int c = 0; c < C; ++c) v[c] = fmaf(w, g[c], v[c]);
        }
    }
}

/* planes_cl [B, 3, H, W, C]; points [B, M, 3]; out [B, M, 3*C] (plane-major channels).
 * Plane projections (utils.py:30-47 `planes`, project_onto_planes :65-79): (x,y), (x,z), (z,y); the first
 * coordinate indexes W, the second H; coordinates are pre-scaled by coord_scale = 2 / box_warp. */
static inline void orc_plane_uv(const float* p, int plane, float s, float* u, float* v) {
    const float x = p[0] * s, y = p[1] * s, z = p[2] * s;
    if (plane == 0) { *u = x; *v = y; } else if (plane == 1) { *u = x; *v = z; } else { *u = z; *v = y; }
}

ORC_API void orc_triplane_sample_fwd(const float* planes_cl, int32_t B, int32_t H, int32_t W, int32_t C,
                                     const float* points, int32_t M, float coord_scale, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (int64_t)B * M; ++q) {
        const int b = (int)(q / M);
        for (int pl = 0; pl < 3; ++pl) {
            float u, v, fx, fy;
            int x0, y0;
            orc_plane_uv(points + 3 * q, pl, coord_scale, &u, &v);
            orc_gs_axis(u, W, &x0, &fx);
            orc_gs_axis(v, H, &y0, &fy);
            float* o = out + (q * 3 + pl) * C;
            for (int c = 0; c < C; ++c) o[c] = 0.f;
            for (int corner = 0; corner < 4; ++corner) {
                const int dx = corner & 1, dy = corner >> 1;
                const int x = x0 + dx, y = y0 + dy;
                if (x < 0 || x >= W || y < 0 || y >= H) continue;
                const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                const float* src = planes_cl + ((((int64_t)b * 3 + pl) * H + y) * W + x) * C;
                for (int c = 0; c < C; ++c) o[c] = fmaf(w, src[c], o[c]);
            }
        }
    }
}

ORC_API void orc_triplane_sample_bwd(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t C,
                                     const float* points, int32_t M, float coord_scale, float* d_planes_cl) {
    for (int64_t q = 0; q < (int64_t)B * M; ++q) {
        const int b = (int)(q / M);
        for (int pl = 0; pl < 3; ++pl) {
            float u, v, fx, fy;
            int x0, y0;
            orc_plane_uv(points + 3 * q, pl, coord_scale, &u, &v);
            orc_gs_axis(u, W, &x0, &fx);
            orc_gs_axis(v, H, &y0, &fy);
            const float* g = d_out + (q * 3 + pl) * C;
            for (int corner = 0; corner < 4; ++corner) {
                const int dx = corner & 1, dy = corner >> 1;
                const int x = x0 + dx, y = y0 + dy;
                if (x < 0 || x >= W || y < 0 || y >= H) continue;
                const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
                float* dst = d_planes_cl + ((((int64_t)b * 3 + pl) * H + y) * W + x) * C;
                for (int c = 0; c < C; ++c) dst[c] = fmaf(w, g[c], dst[c]);
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* occupancy-grid marching — nerfacc.OccGridEstimator.sampling as called at
 * threestudio/models/renderers/nerf_volume_renderer.py:139-180 (behaviour SURVEY.md Appendix B.2);
 * placement convention in include/asd_hip.h ("parity unpinned" vs nerfacc's own lattice phase)       */
/* ---------------------------------------------------------------------------------------------- */
static int orc_ray_aabb(const float o[3], const float d[3], const float* aabb, float* tmin, float* tmax) {
    float t0 = -INFINITY, t1 = INFINITY;
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.f) {
            if (o[a] < aabb[a] || o[a] > aabb[3 + a]) return 0;
            continue;
        }
        const float inv = 1.f / d[a];
        float ta = (aabb[a] - o[a]) * inv, tb = (aabb[3 + a] - o[a]) * inv;
        if (ta > tb) { const float s = ta; ta = tb; tb = s; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    *tmin = t0;
    *tmax = t1;
    return t1 >= t0;
}

static inline int orc_cell_of(const asd_march_cfg* c, const float p[3]) {
    int idx[3];
    for (int a = 0; a < 3; ++a) {
        const float u = (p[a] - c->aabb[a]) / (c->aabb[3 + a] - c->aabb[a]);
        if (!(u >= 0.f && u <= 1.f)) return -1;
        int ci = (int)floorf(u * (float)c->resolution);
        if (ci > c->resolution - 1) ci = c->resolution - 1;
        idx[a] = ci;
    }
    return (idx[0] * c->resolution + idx[1]) * c->resolution + idx[2];
}

/* returns total number of candidates; any output may be NULL (counting pass) */
ORC_API int32_t orc_march(const asd_march_cfg* c, const float* rays_o, const float* rays_d, int32_t n_rays,
                          const uint32_t* occ_bits, const float* jitter, int32_t* count, int32_t* ray_idx,
                          float* t_start, float* t_end, float* points) {
    int32_t total = 0;
    for (int32_t r = 0; r < n_rays; ++r) {
        const float* o = rays_o + 3 * (size_t)r;
        const float* d = rays_d + 3 * (size_t)r;
        int32_t cnt = 0;
        float tmin, tmax;
        if (orc_ray_aabb(o, d, c->aabb, &tmin, &tmax)) {
            const float near_eff = jitter ? fmaf(jitter[r], c->step, c->near_plane) : c->near_plane;
            const float t_begin = fmaxf(tmin, near_eff);
            const float t_exit = fminf(tmax, c->far_plane);
            for (int k = 0; k < c->max_steps; ++k) {
                const float t0 = fmaf((float)k, c->step, t_begin);
                const float t1 = fmaf((float)(k + 1), c->step, t_begin);
                const float tm = (t0 + t1) / 2.0f;
                if (!(tm <= t_exit)) break;
                float p[3];
                /* positions = t_origins + t_dirs * t_positions (nerf_volume_renderer.py:157-158): mul, then add */
                for (int a = 0; a < 3; ++a) p[a] = o[a] + d[a] * tm;
                const int cell = orc_cell_of(c, p);
                if (cell < 0) continue;
                if (!((occ_bits[cell >> 5] >> (cell & 31)) & 1u)) continue;
                if (ray_idx) {
                    ray_idx[total + cnt] = r;
                    t_start[total + cnt] = t0;
                    t_end[total + cnt] = t1;
                    if (points)
                        for (int a = 0; a < 3; ++a) points[3 * (size_t)(total + cnt) + a] = p[a];
                }
                ++cnt;
            }
        }
        if (count) count[r] = cnt;
        total += cnt;
    }
    return total;
}

/* nerfacc render_visibility_from_density: keep = (T >= early_stop_eps) & (alpha >= alpha_thre) */
ORC_API int32_t orc_prune(const float* sigma, const float* t_start, const float* t_end, const int32_t* offset,
                          const int32_t* count, int32_t n_rays, float early_stop_eps, float alpha_thre,
                          uint8_t* keep, int32_t* kept_count) {
    int32_t total = 0;
    for (int32_t r = 0; r < n_rays; ++r) {
        float acc = 0.f;
        int32_t kc = 0;
        for (int32_t i = offset[r]; i < offset[r] + count[r]; ++i) {
            const float sd = sigma[i] * (t_end[i] - t_start[i]);
            const float T = expf(-acc);
            const float alpha = 1.f - expf(-sd);
            const int k = (T >= early_stop_eps) && (alpha >= alpha_thre);
            keep[i] = (uint8_t)k;
            kc += k;
            acc += sd;
        }
        kept_count[r] = kc;
        total += kc;
    }
    return total;
}

/* ---------------------------------------------------------------------------------------------- */
/* compositing — nerfacc.render_weight_from_density + accumulate_along_rays and the glue at
 * threestudio/models/renderers/nerf_volume_renderer.py:312-364               */
/* ---------------------------------------------------------------------------------------------- */
ORC_API void orc_composite_fwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end,
                               const float* rgb, const int32_t* offset, const int32_t* count, int32_t n_rays,
                               const float* bg, float* weights, float* opacity, float* depth, float* rgb_fg,
                               float* z_var, float* comp_rgb) {
    for (int32_t r = 0; r < n_rays; ++r) {
        float acc = 0.f, Tp = 1.f, op = 0.f, dp = 0.f, c[3] = {0, 0, 0};
        const int32_t b = offset[r], e = offset[r] + count[r];
        for (int32_t i = b; i < e; ++i) {
            float T, alpha;
            if (mode == 0) {
                const float sd = sigma[i] * (t_end[i] - t_start[i]);
                T = expf(-acc);
                alpha = 1.f - expf(-sd);
                acc += sd;
            } else {
                T = Tp;
                alpha = sigma[i];
                Tp *= (1.f - alpha);
            }
            const float w = T * alpha, t = (t_start[i] + t_end[i]) * 0.5f;
            weights[i] = w;
            op += w;
            dp = fmaf(w, t, dp);
            for (int k = 0; k < 3; ++k) c[k] = fmaf(w, rgb[3 * (size_t)i + k], c[k]);
        }
        /* mode 2 = alpha compositing with the VolSDF renderer's z-variance sum_i w_i (t_i - depth)^2, un-normalised and un-masked
         * (custom/amortized/models/renderers/generative_space_volsdf_volume_renderer.py:380-385) */
        const float m = mode == 2 ? 1.f : fmaxf(op, 1e-5f);
        const float zm = dp / m;
        float zv = 0.f;
        for (int32_t i = b; i < e; ++i) {
            const float t = (t_start[i] + t_end[i]) * 0.5f;
            zv = fmaf(weights[i] / m, (t - zm) * (t - zm), zv);
        }
        opacity[r] = op;
        depth[r] = dp;
        z_var[r] = (mode == 2 || op > 0.5f) ? zv : 0.f;
        for (int k = 0; k < 3; ++k) {
            rgb_fg[3 * (size_t)r + k] = c[k];
            comp_rgb[3 * (size_t)r + k] = c[k] + bg[3 * (size_t)r + k] * (1.f - op);
        }
    }
}

ORC_API void orc_composite_bwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end,
                               const float* rgb, const int32_t* offset, const int32_t* count, int32_t n_rays,
                               const float* bg, const float* weights, const float* opacity, const float* depth,
                               const float* d_comp_rgb, const float* d_rgb_fg, const float* d_opacity,
                               const float* d_depth, const float* d_z_var, const float* d_weights,
                               float* d_sigma, float* d_rgb, float* d_bg) {
    for (int32_t r = 0; r < n_rays; ++r) {
        const int32_t b = offset[r], e = offset[r] + count[r];
        const float op = opacity[r], m = mode == 2 ? 1.f : fmaxf(op, 1e-5f), zm = depth[r] / m;
        float G[3], gop = d_opacity ? d_opacity[r] : 0.f;
        for (int k = 0; k < 3; ++k) {
            const float gc = d_comp_rgb ? d_comp_rgb[3 * (size_t)r + k] : 0.f;
            G[k] = gc + (d_rgb_fg ? d_rgb_fg[3 * (size_t)r + k] : 0.f);
            gop -= gc * bg[3 * (size_t)r + k];
            if (d_bg) d_bg[3 * (size_t)r + k] = gc * (1.f - op);
        }
        const float gdp = d_depth ? d_depth[r] : 0.f;
        float gzv = (d_z_var && (mode == 2 || op > 0.5f)) ? d_z_var[r] : 0.f;
        float zvu = 0.f;
        if (gzv != 0.f && mode != 2)
            for (int32_t i = b; i < e; ++i) {
                const float t = (t_start[i] + t_end[i]) * 0.5f;
                zvu = fmaf(weights[i] / m, (t - zm) * (t - zm), zvu);
            }
        /* forward quantities: Tn[i] = transmittance AFTER sample i (mode 0), T[i] before sample i (mode 1) */
        const int32_t cnt = e - b;
        float* Tb = (float*)malloc(sizeof(float) * (size_t)(cnt > 0 ? cnt : 1));
        {
            float acc = 0.f, Tp = 1.f;
            for (int32_t i = b; i < e; ++i) {
                if (mode == 0) {
                    acc += sigma[i] * (t_end[i] - t_start[i]);
                    Tb[i - b] = expf(-acc);
                } else {
                    Tb[i - b] = Tp;
                    Tp *= (1.f - sigma[i]);
                }
            }
        }
        /* suffix sums S_i = sum_{j>i} w_j gw_j, walking the ray backwards */
        float S = 0.f;
        for (int32_t i = e - 1; i >= b; --i) {
            const float t = (t_start[i] + t_end[i]) * 0.5f, dt = t_end[i] - t_start[i];
            float gw = gop + gdp * t + (d_weights ? d_weights[i] : 0.f);
            for (int k = 0; k < 3; ++k) gw = fmaf(G[k], rgb[3 * (size_t)i + k], gw);
            if (gzv != 0.f) gw += mode == 2 ? gzv * ((t - zm) * (t - zm) - 2.f * t * zm * (1.f - op)) : gzv * ((t - zm) * (t - zm) - zvu) / m;
            const float w = weights[i];
            if (mode == 0) {
                d_sigma[i] = dt * (Tb[i - b] * gw - S);
            } else {
                d_sigma[i] = Tb[i - b] * gw - S / fmaxf(1.f - sigma[i], 1e-10f);
            }
            for (int k = 0; k < 3; ++k) d_rgb[3 * (size_t)i + k] = w * G[k];
            S = fmaf(w, gw, S);
        }
        free(Tb);
    }
}

/* nerfacc OccGridEstimator._update tail: EMA-max on the updated cells, then
 * binaries = occs > min(mean(occs), occ_thre)                                                        */
ORC_API void orc_occgrid_update(float* occs, int32_t n_cells, const int32_t* cell_idx, const float* occ_new,
                                int32_t n_update, float decay, float occ_thre, uint32_t* occ_bits,
                                uint8_t* binaries) {
    for (int32_t i = 0; i < n_update; ++i) {
        const int32_t c = cell_idx[i];
        occs[c] = fmaxf(occs[c] * decay, occ_new[i]);
    }
    double sum = 0.0;
    for (int32_t i = 0; i < n_cells; ++i) sum += occs[i];
    const float thre = fminf((float)(sum / n_cells), occ_thre);
    memset(occ_bits, 0, sizeof(uint32_t) * (size_t)((n_cells + 31) / 32));
    for (int32_t i = 0; i < n_cells; ++i) {
        const int b = occs[i] > thre;
        binaries[i] = (uint8_t)b;
        if (b) occ_bits[i >> 5] |= 1u << (i & 31);
    }
}


/* ---- camera rays (threestudio/utils/ops.py:183-269 get_ray_directions + get_rays, as used by RandomCameraIterableDataset.collate,
 * threestudio/data/uncond.py:326-337): pixel-centre directions of an OpenGL camera at unit focal length divided by the focal
 * length, rotated by c2w[:3,:3], optionally normalised (F.normalize, eps 1e-12); the origin is c2w[:3,3] for every pixel. -------- */
ORC_API void orc_generate_rays(const float* c2w /*[B,4,4]*/, const float* focal /*[B]*/, int32_t B, int32_t H, int32_t W, int32_t normalize,
                               float* rays_o /*[B,H,W,3]*/, float* rays_d) {
    const float cx = (float)W / 2.0f, cy = (float)H / 2.0f;
    for (int32_t b = 0; b < B; ++b) {
        const float* m = c2w + (size_t)b * 16;
        for (int32_t j = 0; j < H; ++j)
            for (int32_t i = 0; i < W; ++i) {
                const float d0 = (((float)i + 0.5f) - cx) / focal[b], d1 = -(((float)j + 0.5f) - cy) / focal[b], d2 = -1.0f;
                float v[3];
                for (int k = 0; k < 3; ++k) v[k] = d0 * m[4 * k] + d1 * m[4 * k + 1] + d2 * m[4 * k + 2];
                if (normalize) {
                    float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                    len = len > 1e-12f ? len : 1e-12f;
                    for (int k = 0; k < 3; ++k) v[k] = v[k] / len;
                }
                const size_t o = (((size_t)b * H + j) * W + i) * 3;
                for (int k = 0; k < 3; ++k) { rays_d[o + k] = v[k]; rays_o[o + k] = m[4 * k + 3]; }
            }
    }
}
