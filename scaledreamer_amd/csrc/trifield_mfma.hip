// trifield_mfma.hip — the tri-plane field of `Triplane-transformer-sdf` (custom/amortized/models/geometry/triplane_transformer.py:139-240) with
// the two VanillaMLP heads (threestudio/models/networks.py:139-191: 96 -> 64 -> 64 -> 1 | 3, ReLU, no biases) on the MATRIX pipe.
// trifield.hip evaluates one sample per thread with fp32 multiply-adds (one v_fmac per weight and sample); here a wave owns a tile of
// 64 rows (16 samples x {centre, +x, +y, +z} with a finite-difference normal, 64 samples without) and every layer is a product
//        Z^T [units x rows] = W [units x k] * X^T [k x rows]           (v_mfma_f32_16x16x32_f16, A = weight fragment, B = activation fragment)
// evaluated in split-fp16 arithmetic: x * s = hi + lo (s a power of two), W * sW = hi + lo, product = hi*hi + hi*lo + lo*hi accumulated
// in fp32 (~22 bits; the scheme of conv3d.hip).  Scales are powers of two taken on the device from max|planes|, max|W| and from norm
// bounds of the weights (|h1| <= max_j sum_c |W1[j][c]| * max|planes|: interpolation is convex), so nothing overflows fp16 by
// construction and no pass over activations is needed.
//
// Register-resident chain.  The B fragment of a 16x16x32 product holds, in lane (n = l & 15, g = l >> 4), eight k-values of row n; the
// accumulator holds, in the same lane, units 4g .. 4g+3 of row n for every 16-unit block.  Two unit blocks (2t, 2t+1) therefore ARE the
// B fragment of k-step t of the next layer once ReLU-ed, scaled and split — in the k ORDER (e < 4: unit 32t + 4g + e, e >= 4: unit
// 32t + 16 + 4g + e - 4); the weight image of the next layer is packed in that order (tfm_prep_kernel).  Layer 1's B fragment comes from
// the lookup itself: lane (n, g) gathers channels 8g .. 8g+7 of plane ks for its row — k-step = plane.  The last layer (1 | 3 outputs)
// is a dot product on the vector pipe in the accumulator layout plus a sum over the four lane groups.  Nothing goes through LDS but the
// weight images (40 KB per head for the forward chain).
#include <stdlib.h>

#include "trifield_common.h"
#include "trifield_mfma.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float tfm_scale_for(float amax) {          // 2^(14 - floor(log2 amax)): amax * scale in [2^14, 2^15)
    const unsigned b = __float_as_uint(amax);
    if (b == 0u || b >= 0x7f800000u) return 1.f;
    int se = 14 - ((int)((b >> 23) & 0xffu) - 127);
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __uint_as_float((unsigned)(se + 127) << 23);
}
// x = hi + lo: hi = fp16(x) (v_cvt_pk_f16_f32, two per instruction), lo = fp16(x - hi) as ONE mixed-precision fma per element (v_fma_mix{lo,hi}_f16:
// fp16 hi * -1 + fp32 x, rounded once to fp16) — the compiler's form is cvt back + subtract + cvt: 2.5 instructions per element instead of 1.5
typedef half_t half2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tfm_split2(float x0, float x1, half2_& hi, half2_& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(x0), "v"(x1));
}
__device__ __forceinline__ void tfm_split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        half2_ h, l;
        tfm_split2(x[e], x[e + 1], h, l);
        hi[e] = h[0]; hi[e + 1] = h[1];
        lo[e] = l[0]; lo[e + 1] = l[1];
    }
}
// three pieces (33 bits: an fp32 number exactly, short of fp16 underflow) for the one product whose rows cancel (dW3 under a finite-difference normal)
__device__ __forceinline__ void tfm_split8x3(const float (&x)[8], half8& p0, half8& p1, half8& p2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const half_t a = (half_t)x[e];
        const float r1 = x[e] - (float)a;
        const half_t b = (half_t)r1;
        p0[e] = a; p1[e] = b; p2[e] = (half_t)(r1 - (float)b);
    }
}
// a * b to ~2^-33: the six leading products of (a0 + a1 + a2)(b0 + b1 + b2), smallest first
__device__ __forceinline__ floatx4 tfm_mma6(const half8 (&a)[3], const half8 (&b)[3], floatx4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}
// max(x, 0) as one instruction (fmaxf() canonicalises operands that come out of an MFMA with an extra v_max x, x)
__device__ __forceinline__ float tfm_relu(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ floatx4 tfm_mma3(const half8 ah, const half8 al, const half8 bh, const half8 bl, floatx4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    return acc;
}
// sum over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48 hold the same row of the tile)
__device__ __forceinline__ float tfm_rows_sum(float v) {
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// k order of a fragment built from two accumulator blocks
__host__ __device__ __forceinline__ int tfm_perm(int t, int g, int e) { return 32 * t + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4); }

// ---- planes -> planes with a one-texel zero border + max|planes| --------------------------------------------------------------------------------
// padded [3][H + 2][W + 2][32]; amax: bit pattern, zeroed by the caller (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void tfm_pad_kernel(const float* __restrict__ planes, int H, int W, float* __restrict__ padded, unsigned* __restrict__ amax) {
    const size_t total = (size_t)3 * (H + 2) * (W + 2) * 8;                  // float4 units
    float m = 0.f;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int c4 = (int)(q & 7);
        size_t v = q >> 3;
        const int x = (int)(v % (W + 2)) - 1; v /= (W + 2);
        const int y = (int)(v % (H + 2)) - 1;
        const int pl = (int)(v / (H + 2));
        floatx4 t = {0.f, 0.f, 0.f, 0.f};
        if (x >= 0 && x < W && y >= 0 && y < H) {
            t = *(const floatx4*)(planes + (((size_t)pl * H + y) * W + x) * 32 + 4 * c4);
            m = fmaxf(fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))), m);
        }
        *(floatx4*)(padded + 4 * q) = t;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, __float_as_uint(m));
}

// ---- weights -> scales + fragment images (one block per head) ----------------------------------------------------------------------------
// prep (floats): [0] bit pattern of max|planes| (written by asd_absmax_f32 before this kernel) | [16 + 16 head ..] scales | [64 ..] images
__global__ __launch_bounds__(256) void tfm_prep_kernel(const float* __restrict__ w1t_s, const float* __restrict__ w2_s, const float* __restrict__ w3_s,
                                                       const float* __restrict__ w1t_f, const float* __restrict__ w2_f, const float* __restrict__ w3_f,
                                                       float* __restrict__ prep) {
    const int head = blockIdx.x, O = head ? 3 : 1, tid = threadIdx.x;
    const float* w1t = head ? w1t_f : w1t_s;       // [96][64] = W1^T
    const float* w2 = head ? w2_f : w2_s;          // [64][64]
    const float* w3 = head ? w3_f : w3_s;          // [O][64]
    __shared__ float red[256], row1[TF_H], row2[TF_H], bj[TF_H], col2[TF_H], sc[8];
    // max|W1|, max|W2|: a wave reduction each, then four partials
    float m1 = 0.f, m2 = 0.f;
    for (int q = tid; q < TF_NIN * TF_H; q += 256) m1 = fmaxf(m1, fabsf(w1t[q]));
    for (int q = tid; q < TF_H * TF_H; q += 256) m2 = fmaxf(m2, fabsf(w2[q]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m1 = fmaxf(m1, __shfl_xor(m1, o, 64)); m2 = fmaxf(m2, __shfl_xor(m2, o, 64)); }
    if ((tid & 63) == 0) { red[tid >> 6] = m1; red[4 + (tid >> 6)] = m2; }
    // norms: four threads per unit, a quarter of the terms each
    {
        const int j = tid >> 2, part = tid & 3;
        float a = 0.f, a2 = 0.f;
        for (int c = part; c < TF_NIN; c += 4) a += fabsf(w1t[c * TF_H + j]);
        for (int i = part; i < TF_H; i += 4) a2 += fabsf(w2[j * TF_H + i]);
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
        a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
        if (part == 0) {
            row1[j] = a;                                                     // sum_c |W1[j][c]|
            row2[j] = a2;                                                    // sum_i |W2[j][i]|
            float b = 0.f;
            for (int o = 0; o < O; ++o) b += fabsf(w3[o * TF_H + j]);
            bj[j] = b;                                                       // |v2[j]| <= sum_o |W3[o][j]|  (output gradient normalised to max 1)
        }
    }
    __syncthreads();
    {
        const int i = tid >> 2, part = tid & 3;
        float a = 0.f;
        for (int j = part; j < TF_H; j += 4) a += fabsf(w2[j * TF_H + i]) * bj[j];
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
        if (part == 0) col2[i] = a;                                          // |u1[i]| <= sum_j |W2[j][i]| |v2[j]|
    }
    __syncthreads();
    if (tid < 64) {
        float r1 = row1[tid], r2 = row2[tid], b = bj[tid], c2 = col2[tid];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            r1 = fmaxf(r1, __shfl_xor(r1, o, 64)); r2 = fmaxf(r2, __shfl_xor(r2, o, 64));
            b = fmaxf(b, __shfl_xor(b, o, 64)); c2 = fmaxf(c2, __shfl_xor(c2, o, 64));
        }
        if (tid == 0) {
            const float aw1 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), aw2 = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
            const float amax_planes = __uint_as_float(((const unsigned*)prep)[0]);
            float* s = prep + 16 + 16 * head;
            s[TFM_S_E] = tfm_scale_for(amax_planes);
            s[TFM_S_W1] = tfm_scale_for(aw1);
            s[TFM_S_W2] = tfm_scale_for(aw2);
            s[TFM_S_H1] = tfm_scale_for(r1 * amax_planes * 1.0001f);
            s[TFM_S_V2] = tfm_scale_for(b * 1.0001f);
            s[TFM_S_U1] = tfm_scale_for(c2 * 1.0001f);
            s[TFM_S_H2] = tfm_scale_for(r2 * r1 * amax_planes * 1.0002f);
            sc[1] = s[TFM_S_W1]; sc[2] = s[TFM_S_W2];
        }
    }
    __syncthreads();
    const float s1 = sc[1], s2 = sc[2];
    half_t* img = (half_t*)(prep + 64) + (size_t)head * TFM_HEAD_HALVES;
    auto put = [&](half_t* hi, half_t* lo, int q, float v) {
        const half_t h = (half_t)v;
        hi[q] = h; lo[q] = (half_t)(v - (float)h);
    };
    for (int q = tid; q < TFM_A1; q += 256) {                    // [4 mb][3 ks][64 lanes][8]: W1[16 mb + (l & 15)][32 ks + 8 (l >> 4) + e]
        const int e = q & 7, l = (q >> 3) & 63, ks = (q >> 9) % 3, mb = q / (3 * 512);
        const int m = 16 * mb + (l & 15), k = 32 * ks + 8 * (l >> 4) + e;
        put(img + TFM_OFF_A1H, img + TFM_OFF_A1L, q, w1t[k * TF_H + m] * s1);
    }
    for (int q = tid; q < TFM_A2; q += 256) {                    // [4 mb][2 t][64][8]: W2[16 mb + (l & 15)][perm(t, l >> 4, e)]; transposed: W2[perm][16 mb + (l & 15)]
        const int e = q & 7, l = (q >> 3) & 63, t = (q >> 9) & 1, mb = q >> 10;
        const int m = 16 * mb + (l & 15), k = tfm_perm(t, l >> 4, e);
        put(img + TFM_OFF_A2H, img + TFM_OFF_A2L, q, w2[m * TF_H + k] * s2);
        put(img + TFM_OFF_A2TH, img + TFM_OFF_A2TL, q, w2[k * TF_H + m] * s2);
    }
    for (int q = tid; q < TFM_A1T; q += 256) {                   // [6 mb][2 t][64][8]: W1^T[16 mb + (l & 15)][perm] = W1[perm][channel]
        const int e = q & 7, l = (q >> 3) & 63, t = (q >> 9) & 1, mb = q >> 10;
        const int m = 16 * mb + (l & 15), k = tfm_perm(t, l >> 4, e);
        put(img + TFM_OFF_A1TH, img + TFM_OFF_A1TL, q, w1t[m * TF_H + k] * s1);
    }
}

// ---- pieces of the chain ---------------------------------------------------------------------------------------------------------------
// Layer 1 as a software pipeline over its twelve (plane, row block) units: the eight 16-byte loads of unit u + 1 are in flight while unit u is
// interpolated, split and multiplied (left to the scheduler the loads of a unit are waited for where they are issued: the tile is latency-bound).
struct tfm_raw { floatx4 v[8]; float w[4]; };
__device__ __forceinline__ void tfm_issue(const tf_geom& g, const float* __restrict__ padded, int plane, const float (&nrm)[3], int ch, float s, tfm_raw& r) {
    // `padded`: the planes with a one-texel zero border ([3][H + 2][W + 2][32], tfm_pad_kernel): grid_sample's zero padding is READ, not tested —
    // the coordinate is clamped to [-1, size] first, so a point anywhere outside lands on border texels (or on an inside texel with weight 0)
    float u, v;
    tf_plane_uv(nrm[0], nrm[1], nrm[2], plane, u, v);
    const float ix = __builtin_amdgcn_fmed3f(((u + 1.f) * (float)g.W - 1.f) * 0.5f, -1.f, (float)g.W);
    const float iy = __builtin_amdgcn_fmed3f(((v + 1.f) * (float)g.H - 1.f) * 0.5f, -1.f, (float)g.H);
    const int x0 = min((int)floorf(ix), g.W - 1), y0 = min((int)floorf(iy), g.H - 1);
    const float fx = ix - (float)x0, fy = iy - (float)y0;
    const float* row0 = padded + ((size_t)((plane * (g.H + 2) + y0 + 1) * (g.W + 2) + x0 + 1)) * 32 + ch;
    const float* row1 = row0 + (size_t)(g.W + 2) * 32;
    r.w[0] = (1.f - fx) * (1.f - fy) * s; r.w[1] = fx * (1.f - fy) * s;       // s a power of two: folding it in here is exact
    r.w[2] = (1.f - fx) * fy * s;         r.w[3] = fx * fy * s;
    r.v[0] = *(const floatx4*)row0;        r.v[1] = *(const floatx4*)(row0 + 4);
    r.v[2] = *(const floatx4*)(row0 + 32); r.v[3] = *(const floatx4*)(row0 + 36);
    r.v[4] = *(const floatx4*)row1;        r.v[5] = *(const floatx4*)(row1 + 4);
    r.v[6] = *(const floatx4*)(row1 + 32); r.v[7] = *(const floatx4*)(row1 + 36);
}
__device__ __forceinline__ void tfm_interp(const tfm_raw& r, float (&e)[8]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = 0.f;
#pragma unroll
    for (int corner = 0; corner < 4; ++corner)
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[k] = fmaf(r.w[corner], r.v[2 * corner][k], e[k]); e[4 + k] = fmaf(r.w[corner], r.v[2 * corner + 1][k], e[4 + k]); }
}
// consume(plane, nb, bh, bl): the unit's B fragment (channels 8 lg .. 8 lg + 7 of `plane` for the lane's row of block nb); plane / nb are constants after unrolling.
//
// FD (row block k = 1..3 of a tile is the probe + eps e_k of the SAME 16 samples as block 0): the probe blocks carry DIFFERENCES to the centre
// through every linear step — enc_k - enc_0 here, formed in fp32 in the lane that holds both — and are made whole (z_k = z_0 + dz_k) only where
// a ReLU needs them.  A finite difference (s_k - s_0) / eps and the backward pass's (g_k a_k (x) b_k - g_k a_0 (x) b_0) cancel 2-3 digits; on
// whole values the 22-bit split products would leave that cancellation 4x the noise of fp32 arithmetic, on differences they leave none.
template <bool FD, bool DEEP = true, typename F, typename Z>
__device__ __forceinline__ void tfm_layer1(const tf_geom& g, const float* __restrict__ planes, const float (&N)[4][3], int lg, float sE, F&& consume, Z&& zero) {
    // one scheduling region per unit: the matrix products of unit u next to the interpolation / split of unit u + 1 (loads issued one unit ago) and
    // the tap setup + loads of unit u + 2 — three independent strands for the scheduler to interleave.
    // FD: the probe + eps e_k has the centre's coordinates in the plane that does not contain axis k (unless the clamp to the box moved them):
    // its difference fragment is exactly zero there and the unit is skipped — units 3, 6, 9 of 12 when every row of the wave agrees
    // (zero(plane, nb) instead of consume, for callers that keep the fragments).
    bool dead[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) dead[u] = false;
    if (FD) {
        auto all_same = [&](int k, int a0, int a1) {
            return __builtin_amdgcn_ballot_w64(__float_as_uint(N[k][a0]) == __float_as_uint(N[0][a0]) && __float_as_uint(N[k][a1]) == __float_as_uint(N[0][a1])) == ~0ull;
        };
        dead[3] = all_same(3, 0, 1);                // plane 0 = (x, y), probe + z
        dead[6] = all_same(2, 0, 2);                // plane 1 = (x, z), probe + y
        dead[9] = all_same(1, 2, 1);                // plane 2 = (z, y), probe + x
    }
    if constexpr (!DEEP) {
        // one unit ahead only (the weight-gradient kernel: 176 accumulator registers leave no room for two units of taps in flight)
        tfm_raw r1;
        half8 bh1, bl1;
        float e0[8];
        tfm_issue(g, planes, 0, N[0], 8 * lg, sE, r1);
        tfm_interp(r1, e0);
        tfm_split8(e0, bh1, bl1);
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const bool nxt = u + 1 < 12 && !dead[u + 1 < 12 ? u + 1 : 0];
            if (nxt) tfm_issue(g, planes, (u + 1) / 4, N[(u + 1) % 4], 8 * lg, sE, r1);
            if (dead[u]) zero(u / 4, u % 4);
            else consume(u / 4, u % 4, bh1, bl1);
            if (nxt) {
                float e[8];
                tfm_interp(r1, e);
                if (FD) {
                    if ((u + 1) % 4 == 0) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) e0[k] = e[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) e[k] -= e0[k];
                    }
                }
                tfm_split8(e, bh1, bl1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    tfm_raw r[2];
    half8 bh[2], bl[2];
    float e0[8];
    tfm_issue(g, planes, 0, N[0], 8 * lg, sE, r[0]);
    tfm_issue(g, planes, 0, N[1], 8 * lg, sE, r[1]);
    tfm_interp(r[0], e0);
    tfm_split8(e0, bh[0], bl[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        if (dead[u]) zero(u / 4, u % 4);
        else consume(u / 4, u % 4, bh[u & 1], bl[u & 1]);
        if (u + 1 < 12 && !dead[u + 1 < 12 ? u + 1 : 0]) {
            float e[8];
            tfm_interp(r[(u + 1) & 1], e);
            if (FD) {
                if ((u + 1) % 4 == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e0[k] = e[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) e[k] -= e0[k];
                }
            }
            tfm_split8(e, bh[(u + 1) & 1], bl[(u + 1) & 1]);
        }
        if (u + 2 < 12 && !dead[u + 2 < 12 ? u + 2 : 0]) tfm_issue(g, planes, (u + 2) / 4, N[(u + 2) % 4], 8 * lg, sE, r[u & 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// FD: probe blocks hold differences after a linear step: make them whole
template <bool FD, int M>
__device__ __forceinline__ void tfm_merge(floatx4 (&z)[M][4]) {
    if (!FD) return;
#pragma unroll
    for (int mb = 0; mb < M; ++mb)
#pragma unroll
        for (int k = 1; k < 4; ++k) z[mb][k] += z[mb][0];
}

// accumulator blocks (2t, 2t+1) of NB row blocks (whole values) -> ReLU, * c -> B fragments of k-step t; FD: the probe blocks as differences to block 0
template <int NB, bool FD = false>
__device__ __forceinline__ void tfm_relu_frags(const floatx4 (&acc)[4][NB], float c, half8 (&bh)[2][NB], half8 (&bl)[2][NB]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float x[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { x[r] = tfm_relu(acc[2 * t][nb][r]) * c; x[4 + r] = tfm_relu(acc[2 * t + 1][nb][r]) * c; }
            if (FD && nb > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { x[r] -= tfm_relu(acc[2 * t][0][r]) * c; x[4 + r] -= tfm_relu(acc[2 * t + 1][0][r]) * c; }
            }
            tfm_split8(x, bh[t][nb], bl[t][nb]);
            __builtin_amdgcn_sched_barrier(0);        // one fragment at a time: left alone the scheduler interleaves all of them (400+ registers)
        }
}
// Z^T = W2 * H1^T for NB row blocks: a2h / a2l = LDS images [4 mb][2 t][64 lanes] of half8
template <int NB>
__device__ __forceinline__ void tfm_layer2(const half8* a2h, const half8* a2l, int lane, const half8 (&bh)[2][NB], const half8 (&bl)[2][NB], floatx4 (&z)[4][NB]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) z[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const half8 ah = a2h[(mb * 2 + t) * 64 + lane], al = a2l[(mb * 2 + t) * 64 + lane];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) z[mb][nb] = tfm_mma3(ah, al, bh[t][nb], bl[t][nb], z[mb][nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// out[nb][o] = c * sum_j W3[o][j] relu(z[j][row]): the lane's 16 units, then the four lane groups; FD: out of a probe block is the DIFFERENCE to block 0
template <int NB, int O, bool FD = false>
__device__ __forceinline__ void tfm_layer3(const floatx4 (&z)[4][NB], const float* __restrict__ w3, int g, float c, float (&out)[NB][O]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int o = 0; o < O; ++o) out[nb][o] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const floatx4 w = *(const floatx4*)(w3 + o * TF_H + 16 * mb + 4 * g);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float h = tfm_relu(z[mb][nb][r]);
                    if (FD && nb > 0) h -= tfm_relu(z[mb][0][r]);
                    out[nb][o] = fmaf(w[r], h, out[nb][o]);
                }
        }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int o = 0; o < O; ++o) out[nb][o] = tfm_rows_sum(out[nb][o]) * c;
}

// the point of row block nb of the lane's sample: FD: the centre and its three clamped probes; else the centre of sample block nb
template <bool FD>
__device__ __forceinline__ void tfm_point(const asd_field_cfg& c, const float* __restrict__ points, int n, int tile, int q16, int nb, int& i, float (&p)[3]) {
    i = FD ? tile * 16 + q16 : tile * 64 + nb * 16 + q16;
    const int ic = i < n ? i : n - 1;
    p[0] = points[3 * (size_t)ic]; p[1] = points[3 * (size_t)ic + 1]; p[2] = points[3 * (size_t)ic + 2];
    if (FD && nb > 0) {                                                      // as trifield.hip: every coordinate of a probe is clamped
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = asd_clampf(p[k] + (k == nb - 1 ? c.fd_eps : 0.f), -c.radius, c.radius);
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------------------------
// MODE 0: finite-difference normal — a tile is 16 samples x {centre, +x, +y, +z}: the sdf head on all four row blocks, the feature head
//         (if wanted) on the centres; the four sdf values of a sample meet in one lane.
// MODE 1: 64 samples per tile, sdf head only.   MODE 2: 64 samples per tile, feature head only (features without a normal: two launches).
template <int MODE>
__global__ __launch_bounds__(256, 2) void tfm_fwd_kernel(const tf_geom g, const asd_field_cfg c, const float* __restrict__ planes, const float* __restrict__ prep,
                                                          const float* __restrict__ w3s, const float* __restrict__ w3f, const float* __restrict__ points, int n,
                                                          float* __restrict__ sdf, float* __restrict__ features, float* __restrict__ normal,
                                                          float* __restrict__ fd_grad) {
    constexpr bool FD = MODE == 0;
    constexpr int OP = MODE == 2 ? 3 : 1;                                    // outputs of the head that runs on all four row blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];             // [sdf head: a1h a1l a2h a2l | feature head: same]  2 x 40 KB
    {
        const uint4* src0 = (const uint4*)((const half_t*)(prep + 64));
        const uint4* src1 = (const uint4*)((const half_t*)(prep + 64) + TFM_HEAD_HALVES);
        uint4* dst = (uint4*)smem;
        constexpr int N16 = TFM_FWD_HALVES * 2 / 16;
        for (int q = threadIdx.x; q < N16; q += 256) {          // MODE 1 / 2: one head, 40 KB
            if (MODE != 2) dst[q] = src0[q];
            if (MODE != 1) dst[(MODE == 2 ? 0 : N16) + q] = src1[q];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q16 = lane & 15, lg = lane >> 4;
    const half8* img_s = (const half8*)smem;
    const half8* img_f = (const half8*)(smem + (MODE == 2 ? 0 : TFM_FWD_HALVES * 2));
    const float* ss = prep + 16;
    const float* sf = prep + 32;
    const float sE = ss[TFM_S_E];
    const float c1s = ss[TFM_S_H1] / (ss[TFM_S_W1] * sE), c2s = 1.f / (ss[TFM_S_W2] * ss[TFM_S_H1]);
    const float c1f = sf[TFM_S_H1] / (sf[TFM_S_W1] * sE), c2f = 1.f / (sf[TFM_S_W2] * sf[TFM_S_H1]);
    const half8* img_p = MODE == 2 ? img_f : img_s;
    const float c1p = MODE == 2 ? c1f : c1s, c2p = MODE == 2 ? c2f : c2s;
    const float* w3p = MODE == 2 ? w3f : w3s;
    const bool feat = FD && features != nullptr;
    const int n_tiles = (n + (FD ? 16 : 64) - 1) / (FD ? 16 : 64);
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        int idx[4];
        float P[4][3], N[4][3];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            tfm_point<FD>(c, points, n, tile, q16, nb, idx[nb], P[nb]);
            tf_norm(c, P[nb][0], P[nb][1], P[nb][2], N[nb][0], N[nb][1], N[nb][2]);
        }
        floatx4 acc[4][4], accf[4][1];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
            accf[mb][0] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- layer 1: k-step = plane
        tfm_layer1<FD>(g, planes, N, lg, sE, [&](int plane, int nb, const half8& bh, const half8& bl) __attribute__((always_inline)) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const half8 ah = img_p[TFM_OFF_A1H / 8 + (mb * 3 + plane) * 64 + lane], al = img_p[TFM_OFF_A1L / 8 + (mb * 3 + plane) * 64 + lane];
                acc[mb][nb] = tfm_mma3(ah, al, bh, bl, acc[mb][nb]);
            }
            if (nb == 0 && feat) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const half8 ah = img_f[TFM_OFF_A1H / 8 + (mb * 3 + plane) * 64 + lane], al = img_f[TFM_OFF_A1L / 8 + (mb * 3 + plane) * 64 + lane];
                    accf[mb][0] = tfm_mma3(ah, al, bh, bl, accf[mb][0]);
                }
            }
        }, [](int, int) {});
        // ---- layers 2, 3 of the head on all row blocks
        float o[4][OP];
        {
            half8 bh[2][4], bl[2][4];
            tfm_merge<FD>(acc);
            tfm_relu_frags<4, FD>(acc, c1p, bh, bl);
            tfm_layer2<4>(img_p + TFM_OFF_A2H / 8, img_p + TFM_OFF_A2L / 8, lane, bh, bl, acc);
            tfm_merge<FD>(acc);
            tfm_layer3<4, OP, FD>(acc, w3p, lg, c2p, o);           // FD: o[k] = out_k - out_0 for the probes
        }
        float of[1][3];
        if (feat) {
            half8 bh[2][1], bl[2][1];
            tfm_relu_frags<1>(accf, c1f, bh, bl);
            tfm_layer2<1>(img_f + TFM_OFF_A2H / 8, img_f + TFM_OFF_A2L / 8, lane, bh, bl, accf);
            tfm_layer3<1, 3>(accf, w3f, lg, c2f, of);
        }
        // ---- outputs: lane group 0 writes its 16 rows
        if (lg == 0) {
            if (FD) {
                const int i = idx[0];
                if (i < n) {
                    const float b0 = tf_bias(c, P[0][0], P[0][1], P[0][2]);
                    sdf[i] = o[0][0] + b0;
                    if (feat) { features[3 * (size_t)i] = of[0][0]; features[3 * (size_t)i + 1] = of[0][1]; features[3 * (size_t)i + 2] = of[0][2]; }
                    float nr[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) nr[k] = (o[k + 1][0] + (tf_bias(c, P[k + 1][0], P[k + 1][1], P[k + 1][2]) - b0)) / c.fd_eps;
                    if (fd_grad) { fd_grad[3 * (size_t)i] = nr[0]; fd_grad[3 * (size_t)i + 1] = nr[1]; fd_grad[3 * (size_t)i + 2] = nr[2]; }
                    if (normal) {
                        const float inv = 1.f / fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
                        normal[3 * (size_t)i] = nr[0] * inv; normal[3 * (size_t)i + 1] = nr[1] * inv; normal[3 * (size_t)i + 2] = nr[2] * inv;
                    }
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int i = idx[nb];
                    if (i >= n) continue;
                    if (MODE == 1) sdf[i] = o[nb][0] + tf_bias(c, P[nb][0], P[nb][1], P[nb][2]);
                    else {
#pragma unroll
                        for (int k = 0; k < OP; ++k) features[OP * (size_t)i + k] = o[nb][k];
                    }
                }
            }
        }
    }
}

// ---- backward, data path ----------------------------------------------------------------------------------------------------------------
// One head per launch: forward chain again (nothing but the points is kept from the forward pass), then the chain of TRANSPOSED weight images
//   v2 = W3^T dn (.) [z2 > 0]  ->  u1 = W2^T v2 (.) [z1 > 0]  ->  denc = W1^T u1
// for the output gradient NORMALISED per row (dn = d_out / G, G a power of two >= max|d_out|; the sdf head has one output: dn = 1), so every
// operand is bounded by a norm of the weights and the static scales of tfm_prep_kernel hold; the row's G multiplies the result.  Leaves the feature
// gradient rows for the scatter (sdf head: =, feature head: += into the centre rows) and the head's dW3.
struct tfm_bwd_args {
    tf_geom g; asd_field_cfg c;
    const float* planes; const float* prep; const float* w3;          // this head's W3 [O][64]
    const float* points; const float* sdf;
    int i0, n_chunk, npt;                                             // row of (sample li of the chunk, point pt) = npt * li + pt
    const float* d_sdf; const float* d_features; const float* d_normal; const float* d_fd_grad;
    float* denc; float* pts;                                          // [rows][96], [rows][3]
    float* dw1; float* dw2; float* dw3;                               // [64][96], [64][64], [O][64]  (+=)
};

// power of two >= |x| (x finite): exact divisor for the row normalisation
__device__ __forceinline__ float tfm_pow2_above(float x) {
    const unsigned b = __float_as_uint(x) & 0x7f800000u;
    return b == 0u ? 0.f : __uint_as_float(b + 0x00800000u);
}

// the per-row output gradient of a tile in the lane that owns the row: G[nb] (multiplier) and dn[nb][o] (|dn| <= 1)
// FD: ds4[k] = sdf(probe k) - sdf(centre); d0 = the centre's own output gradient (G[0] = d0 - sum_k G[k]: the centre row of every probe term)
template <int O, bool FD>
__device__ __forceinline__ void tfm_row_grads(const tfm_bwd_args& a, const int (&li)[4], const float (&ds4)[3], float (&G)[4], float (&dn)[4][O], float& d0) {
    d0 = 0.f;
    if (O == 1 && FD) {
        const bool active = li[0] < a.n_chunk;
        const size_t i = (size_t)a.i0 + (active ? li[0] : 0);
        float ds = (active && a.d_sdf) ? a.d_sdf[i] : 0.f;
        float dsk[3] = {0.f, 0.f, 0.f};
        if (active) {
            float dnr[3] = {0.f, 0.f, 0.f};
            if (a.d_normal) {
                float nr[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) nr[k] = ds4[k] / a.c.fd_eps;
                const float len = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
                const float g0 = a.d_normal[3 * i], g1 = a.d_normal[3 * i + 1], g2 = a.d_normal[3 * i + 2];
                if (len > 1e-12f) {
                    const float inv = 1.f / len;
                    const float n0 = nr[0] * inv, n1 = nr[1] * inv, n2 = nr[2] * inv;
                    const float dot = n0 * g0 + n1 * g1 + n2 * g2;
                    dnr[0] = (g0 - n0 * dot) * inv; dnr[1] = (g1 - n1 * dot) * inv; dnr[2] = (g2 - n2 * dot) * inv;
                } else {
                    dnr[0] = g0 * 1e12f; dnr[1] = g1 * 1e12f; dnr[2] = g2 * 1e12f;
                }
            }
            if (a.d_fd_grad) { dnr[0] += a.d_fd_grad[3 * i]; dnr[1] += a.d_fd_grad[3 * i + 1]; dnr[2] += a.d_fd_grad[3 * i + 2]; }
            d0 = ds;
#pragma unroll
            for (int k = 0; k < 3; ++k) { dsk[k] = dnr[k] / a.c.fd_eps; ds -= dnr[k] / a.c.fd_eps; }
        }
        G[0] = ds; G[1] = dsk[0]; G[2] = dsk[1]; G[3] = dsk[2];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dn[nb][0] = 1.f;
    } else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const bool active = li[nb] < a.n_chunk;
            const size_t i = (size_t)a.i0 + (active ? li[nb] : 0);
            if (O == 1) {
                G[nb] = (active && a.d_sdf) ? a.d_sdf[i] : 0.f;
                dn[nb][0] = 1.f;
            } else {
                float df[O], m = 0.f;
#pragma unroll
                for (int o = 0; o < O; ++o) { df[o] = active ? a.d_features[O * i + o] : 0.f; m = fmaxf(m, fabsf(df[o])); }
                m = tfm_pow2_above(m);
                const float inv = m > 0.f ? 1.f / m : 0.f;
                G[nb] = m;
#pragma unroll
                for (int o = 0; o < O; ++o) dn[nb][o] = df[o] * inv;
            }
        }
    }
}

// W3 of the head at the lane's units of block mb in the chain layout (16 mb + 4 lg + r), from the copy in LDS
template <int O>
__device__ __forceinline__ void tfm_w3_chain(const float* w3s, int mb, int lg, floatx4 (&w)[O]) {
#pragma unroll
    for (int o = 0; o < O; ++o) w[o] = *(const floatx4*)(w3s + o * TF_H + 16 * mb + 4 * lg);
}

template <int O, bool FD>
__global__ __launch_bounds__(256, 2) void tfm_bwd_data_kernel(const tfm_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];             // the head's eight images: 80 KB exactly, two blocks per CU (W3 comes from L1)
    constexpr int head = O == 3;
    {
        const uint4* src = (const uint4*)((const half_t*)(a.prep + 64) + (size_t)head * TFM_HEAD_HALVES);
        uint4* dst = (uint4*)smem;
        for (int q = threadIdx.x; q < TFM_HEAD_HALVES * 2 / 16; q += 256) dst[q] = src[q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q16 = lane & 15, lg = lane >> 4;
    const half8* img = (const half8*)smem;
    const float* w3s = a.w3;
    const float* sc = a.prep + 16 + 16 * head;
    const float sE = sc[TFM_S_E], sV2 = sc[TFM_S_V2];
    const float c1 = sc[TFM_S_H1] / (sc[TFM_S_W1] * sE), c2 = 1.f / (sc[TFM_S_W2] * sc[TFM_S_H1]);
    const float cU = sc[TFM_S_U1] / (sc[TFM_S_W2] * sV2), cD = 1.f / (sc[TFM_S_W1] * sc[TFM_S_U1]);
    const int n_tiles = (a.n_chunk + (FD ? 16 : 64) - 1) / (FD ? 16 : 64);
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        int li[4];
        float N[4][3], bias4[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            int i;
            float P[3];
            li[nb] = FD ? tile * 16 + q16 : tile * 64 + nb * 16 + q16;
            tfm_point<FD>(a.c, a.points + 3 * (size_t)a.i0, a.n_chunk, tile, q16, nb, i, P);
            tf_norm(a.c, P[0], P[1], P[2], N[nb][0], N[nb][1], N[nb][2]);
            bias4[nb] = (O == 1 && FD) ? tf_bias(a.c, P[0], P[1], P[2]) : 0.f;
            if (O == 1 && lg == 0 && li[nb] < a.n_chunk) {                   // the rows' points for the scatter
                const size_t row = (size_t)a.npt * li[nb] + (FD ? nb : 0);
                a.pts[3 * row] = N[nb][0]; a.pts[3 * row + 1] = N[nb][1]; a.pts[3 * row + 2] = N[nb][2];
            }
        }
        floatx4 acc[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
        tfm_layer1<FD>(a.g, a.planes, N, lg, sE, [&](int plane, int nb, const half8& bh, const half8& bl) __attribute__((always_inline)) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const half8 ah = img[TFM_OFF_A1H / 8 + (mb * 3 + plane) * 64 + lane], al = img[TFM_OFF_A1L / 8 + (mb * 3 + plane) * 64 + lane];
                acc[mb][nb] = tfm_mma3(ah, al, bh, bl, acc[mb][nb]);
            }
        }, [](int, int) {});
        tfm_merge<FD>(acc);
        unsigned long long m1 = 0ull;                                        // bit (mb * 4 + nb) * 4 + r: layer 1's pre-activation > 0
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m1 |= (unsigned long long)(acc[mb][nb][r] > 0.f) << ((mb * 4 + nb) * 4 + r);
        half8 vh[2][4], vl[2][4];
        tfm_relu_frags<4, FD>(acc, c1, vh, vl);
        tfm_layer2<4>(img + TFM_OFF_A2H / 8, img + TFM_OFF_A2L / 8, lane, vh, vl, acc);
        tfm_merge<FD>(acc);
        // ---- the rows' output gradients (the probes' sdf differences feed the normalisation of the finite-difference normal)
        float ds4[3] = {0.f, 0.f, 0.f};
        if (O == 1 && FD) {
            float o[4][1];
            tfm_layer3<4, 1, true>(acc, w3s, lg, c2, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) ds4[k] = o[k + 1][0] + (bias4[k + 1] - bias4[0]);
        }
        float G[4], dn[4][O], d0;
        tfm_row_grads<O, FD>(a, li, ds4, G, dn, d0);
        // ---- v2 = W3^T dn under layer 2's ReLU mask: the B fragments of the transposed chain
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx4 w3v[2][O];
            tfm_w3_chain<O>(w3s, 2 * t, lg, w3v[0]);
            tfm_w3_chain<O>(w3s, 2 * t + 1, lg, w3v[1]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float x[8];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = 0.f;
#pragma unroll
                        for (int o = 0; o < O; ++o) v = fmaf(dn[nb][o], w3v[hb][o][r], v);
                        x[4 * hb + r] = acc[2 * t + hb][nb][r] > 0.f ? v * sV2 : 0.f;
                        if (FD && nb > 0) x[4 * hb + r] -= acc[2 * t + hb][0][r] > 0.f ? v * sV2 : 0.f;      // (one output: v is the same for the four points)
                    }
                tfm_split8(x, vh[t][nb], vl[t][nb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- u1 = W2^T v2, masked by layer 1's ReLU
        tfm_layer2<4>(img + TFM_OFF_A2TH / 8, img + TFM_OFF_A2TL / 8, lane, vh, vl, acc);
        tfm_merge<FD>(acc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float x[8];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x[4 * hb + r] = ((m1 >> (((2 * t + hb) * 4 + nb) * 4 + r)) & 1ull) ? acc[2 * t + hb][nb][r] * cU : 0.f;
                        if (FD && nb > 0) x[4 * hb + r] -= ((m1 >> (((2 * t + hb) * 4) * 4 + r)) & 1ull) ? acc[2 * t + hb][0][r] * cU : 0.f;
                    }
                tfm_split8(x, vh[t][nb], vl[t][nb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        // ---- denc = W1^T u1 (six 16-channel blocks), times the row's G; lane (n, g) holds channels 16 mb + 4 g .. + 3 of row n
#pragma unroll
        for (int mb = 0; mb < 6; ++mb) {
            floatx4 d[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) d[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const half8 ah = img[TFM_OFF_A1TH / 8 + (mb * 2 + t) * 64 + lane], al = img[TFM_OFF_A1TL / 8 + (mb * 2 + t) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) d[nb] = tfm_mma3(ah, al, vh[t][nb], vl[t][nb], d[nb]);
            }
            if (FD) { d[1] += d[0]; d[2] += d[0]; d[3] += d[0]; }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                if (li[nb] >= a.n_chunk) continue;
                const size_t row = (size_t)a.npt * li[nb] + (FD ? nb : 0);
                float* dst = a.denc + row * TF_NIN + 16 * mb + 4 * lg;
                const float k = G[nb] * cD;
                floatx4 v = {d[nb][0] * k, d[nb][1] * k, d[nb][2] * k, d[nb][3] * k};
                if (O == 3) { const floatx4 old = *(const floatx4*)dst; v += old; }
                *(floatx4*)dst = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- backward, weight gradients --------------------------------------------------------------------------------------------------------
// dW3 = sum_rows (G dn) (x) h2, dW2 = sum_rows (G v2) (x) h1 and dW1 = sum_rows (G u1) (x) enc contract over ROWS: both operands of those products
// must hold a unit (or channel, or output) per lane and rows along k — the TRANSPOSE of the chain's fragments.  Every product of the chain is
// therefore evaluated a second time with the operands swapped (Z = X W^T instead of Z^T = W X^T: same fragments, same images), whose accumulator
// holds, in lane (unit = l & 15, g), rows 4g .. 4g+3 of every row block: two row blocks are one k-step of the contraction (the k order of
// tfm_perm, the same for both operands).  The lookup is transposed by a product with an identity fragment (exact: 1 * hi and 1 * lo are fp16
// numbers).  The rows' G spans many orders of magnitude across a chunk: a wave keeps a RUNNING power-of-two scale S >= every |G| it has seen
// (operands carry G / S, the accumulators are rescaled when S grows — exact), so the fp16 operands stay in range without a pass over the
// gradients.  One wave per SIMD (176 accumulator registers + the chain); per-wave LDS: the lookup fragments of the tile (24 KB), the rows' G.
#ifndef TFM_W_DEEP
#define TFM_W_DEEP false    // lookup pipeline of this kernel one unit deep (two in the forward / data kernels): same-box A/B of C5 at the 256 x 256 render
#endif                      // 267.7 vs 271.6 ms per step (tools/r5_tfm_ab.sh)
template <int O, bool FD>
__global__ __launch_bounds__(256, 1) void tfm_bwd_weights_kernel(const tfm_bwd_args a) {
    constexpr int head = O == 3;
    constexpr int IMG_HALVES = TFM_OFF_A1TH;                                 // a1 a2 a2t (hi, lo): 56 KB
    constexpr int ENC_BYTES = 3 * 4 * 2 * 64 * 16;                           // per wave: [plane][nb][hi | lo][lane] half8
    constexpr int GBUF_FLOATS = (1 + O) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {
        const uint4* src = (const uint4*)((const half_t*)(a.prep + 64) + (size_t)head * TFM_HEAD_HALVES);
        uint4* dst = (uint4*)smem;
        for (int q = threadIdx.x; q < IMG_HALVES * 2 / 16; q += 256) dst[q] = src[q];
        float* w3d = (float*)(smem + IMG_HALVES * 2 + 4 * ENC_BYTES) + 4 * GBUF_FLOATS;
        for (int q = threadIdx.x; q < O * TF_H; q += 256) w3d[q] = a.w3[q];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q16 = lane & 15, lg = lane >> 4;
    const half8* img = (const half8*)smem;
    half8* encbuf = (half8*)(smem + IMG_HALVES * 2 + wave * ENC_BYTES);
    float* gbuf = (float*)(smem + IMG_HALVES * 2 + 4 * ENC_BYTES) + wave * GBUF_FLOATS;
    const float* w3s = (const float*)(smem + IMG_HALVES * 2 + 4 * ENC_BYTES) + 4 * GBUF_FLOATS;
    const float* sc = a.prep + 16 + 16 * head;
    const float sE = sc[TFM_S_E], sV2 = sc[TFM_S_V2];
    const float c1 = sc[TFM_S_H1] / (sc[TFM_S_W1] * sE), c2 = 1.f / (sc[TFM_S_W2] * sc[TFM_S_H1]);
    const float cU = sc[TFM_S_U1] / (sc[TFM_S_W2] * sV2), cH2 = sc[TFM_S_H2] * c2;
    float w3n[O][4];                                     // W3 at the lane's units of the swapped layout (16 mb + q16)
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) w3n[o][mb] = w3s[o * TF_H + 16 * mb + q16];
    // identity fragments: B operand with a one at k == 16 half + (l & 15)
    half8 ident[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) ident[hf][e] = (half_t)((8 * lg + e == 16 * hf + q16) ? 1.f : 0.f);
    floatx4 dw3[4], dw2[4][4], dw1[4][6];                // [block of the gradient operand][block of the activation operand]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dw3[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) dw2[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 6; ++j) dw1[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    float S = 0.f;                                       // running scale of G (power of two, wave-uniform)
    const int n_tiles = (a.n_chunk + (FD ? 16 : 64) - 1) / (FD ? 16 : 64);
    for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
        int li[4];
        float bias4[4];
        floatx4 zt[4][4];
        {
            float N[4][3];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int i;
                float P[3];
                li[nb] = FD ? tile * 16 + q16 : tile * 64 + nb * 16 + q16;
                tfm_point<FD>(a.c, a.points + 3 * (size_t)a.i0, a.n_chunk, tile, q16, nb, i, P);
                tf_norm(a.c, P[0], P[1], P[2], N[nb][0], N[nb][1], N[nb][2]);
                bias4[nb] = (O == 1 && FD) ? tf_bias(a.c, P[0], P[1], P[2]) : 0.f;
            }
            // ---- layer 1 (chain orientation); the lookup fragments go to LDS: the swapped products and the last step read them back
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) zt[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
            tfm_layer1<FD, TFM_W_DEEP>(a.g, a.planes, N, lg, sE, [&](int plane, int nb, const half8& bh, const half8& bl) __attribute__((always_inline)) {
                encbuf[((plane * 4 + nb) * 2 + 0) * 64 + lane] = bh;
                encbuf[((plane * 4 + nb) * 2 + 1) * 64 + lane] = bl;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const half8 ah = img[TFM_OFF_A1H / 8 + (mb * 3 + plane) * 64 + lane], al = img[TFM_OFF_A1L / 8 + (mb * 3 + plane) * 64 + lane];
                    zt[mb][nb] = tfm_mma3(ah, al, bh, bl, zt[mb][nb]);
                }
            }, [&](int plane, int nb) __attribute__((always_inline)) {
                const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                encbuf[((plane * 4 + nb) * 2 + 0) * 64 + lane] = z;
                encbuf[((plane * 4 + nb) * 2 + 1) * 64 + lane] = z;
            });
        }
        // ---- layer 2 (chain orientation): zt <- Z2^T; h1's fragments stay for the swapped product below
        half8 h1h[2][4], h1l[2][4];
        tfm_merge<FD>(zt);
        tfm_relu_frags<4, FD>(zt, c1, h1h, h1l);             // FD: [h1_0 | h1_k - h1_0]
        tfm_layer2<4>(img + TFM_OFF_A2H / 8, img + TFM_OFF_A2L / 8, lane, h1h, h1l, zt);
        tfm_merge<FD>(zt);
        // ---- the rows' output gradients, in the lanes that own the rows, and their exchange to the swapped layout
        float ds4[3] = {0.f, 0.f, 0.f};
        if (O == 1 && FD) {
            float o[4][1];
            tfm_layer3<4, 1, true>(zt, w3s, lg, c2, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) ds4[k] = o[k + 1][0] + (bias4[k + 1] - bias4[0]);
        }
        float G[4], dn[4][O], d0;
        tfm_row_grads<O, FD>(a, li, ds4, G, dn, d0);
        // FD: the weight-gradient sums are regrouped around the centre — sum_pt G_pt a_pt (x) b_pt = (d0 a_0 + sum_k G_k (a_k - a_0)) (x) b_0
        // + sum_k G_k a_k (x) (b_k - b_0) — so the row multipliers are [d0 | G_1 | G_2 | G_3] and every b operand is [b_0 | b_k - b_0]
        if (FD) G[0] = d0;
        unsigned long long m2t = 0ull;                   // layer 2's ReLU mask in the chain layout (all that is kept of zt)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m2t |= (unsigned long long)(zt[mb][nb][r] > 0.f) << ((mb * 4 + nb) * 4 + r);
        float gm = fmaxf(fmaxf(fabsf(G[0]), fabsf(G[1])), fmaxf(fabsf(G[2]), fabsf(G[3])));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
        gm = tfm_pow2_above(gm) * (FD ? 8.f : 1.f);      // FD: the centre block's operand sums up to seven terms
        if (gm > S) {                                    // wave-uniform, rare: the accumulators move to the new scale
            const float f = S / gm;                      // 0 for the first tile (the accumulators are zero)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dw3[i] *= f;
#pragma unroll
                for (int j = 0; j < 4; ++j) dw2[i][j] *= f;
#pragma unroll
                for (int j = 0; j < 6; ++j) dw1[i][j] *= f;
            }
            S = gm;
        }
        const float invS = S > 0.f ? 1.f / S : 0.f;
        if (lg == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                gbuf[nb * 16 + q16] = G[nb] * invS;
#pragma unroll
                for (int o = 0; o < O; ++o) gbuf[(1 + o) * 64 + nb * 16 + q16] = dn[nb][o];
            }
        }
        floatx4 gq[4];                                   // G / S of rows 4 lg .. 4 lg + 3 of every row block
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) gq[nb] = *(const floatx4*)(gbuf + nb * 16 + 4 * lg);
        // (G / S) dn with an OUTPUT per lane (lanes >= O carry zeros): the gradient operand of dW3, scaled by 2^14
        half8 gd[2][3];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float x[8];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                floatx4 d = {1.f, 1.f, 1.f, 1.f};
                if (O > 1) d = *(const floatx4*)(gbuf + (1 + (q16 < O ? q16 : 0)) * 64 + (2 * kk + hb) * 16 + 4 * lg);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[4 * hb + r] = q16 < O ? gq[2 * kk + hb][r] * d[r] * 16384.f : 0.f;
            }
            tfm_split8x3(x, gd[kk][0], gd[kk][1], gd[kk][2]);
        }
        // ---- Z2 in the swapped orientation, one unit block at a time -> h2 and (G / S) v2 with a unit per lane: dW3, and the gradient operand of dW2
        half8 gvh[4][2], gvl[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            floatx4 zn[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) zn[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const half8 ah = img[TFM_OFF_A2H / 8 + (mb * 2 + t) * 64 + lane], al = img[TFM_OFF_A2L / 8 + (mb * 2 + t) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) zn[nb] = tfm_mma3(h1h[t][nb], h1l[t][nb], ah, al, zn[nb]);
            }
            if (FD) { zn[1] += zn[0]; zn[2] += zn[0]; zn[3] += zn[0]; }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float x[8], y[8];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    const int nb = 2 * kk + hb;
                    floatx4 v = {w3n[0][mb], w3n[0][mb], w3n[0][mb], w3n[0][mb]};
                    if (O > 1) {
                        v = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int o = 0; o < O; ++o) {
                            const floatx4 d = *(const floatx4*)(gbuf + (1 + o) * 64 + nb * 16 + 4 * lg);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(d[r], w3n[o][mb], v[r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (FD) {
                            const float v0 = zn[0][r] > 0.f ? v[r] : 0.f;
                            if (nb == 0) {
                                float t = gq[0][r] * v0;
#pragma unroll
                                for (int k2 = 1; k2 < 4; ++k2) t = fmaf(gq[k2][r], (zn[k2][r] > 0.f ? v[r] : 0.f) - v0, t);
                                x[4 * hb + r] = t * sV2;
                                y[4 * hb + r] = tfm_relu(zn[0][r]) * cH2;
                            } else {
                                x[4 * hb + r] = zn[nb][r] > 0.f ? gq[nb][r] * v[r] * sV2 : 0.f;
                                y[4 * hb + r] = (tfm_relu(zn[nb][r]) - tfm_relu(zn[0][r])) * cH2;
                            }
                        } else {
                            x[4 * hb + r] = zn[nb][r] > 0.f ? gq[nb][r] * v[r] * sV2 : 0.f;
                            y[4 * hb + r] = tfm_relu(zn[nb][r]) * cH2;
                        }
                    }
                }
                tfm_split8(x, gvh[mb][kk], gvl[mb][kk]);
                half8 h2[3];
                tfm_split8x3(y, h2[0], h2[1], h2[2]);
                dw3[mb] = tfm_mma6(gd[kk], h2, dw3[mb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- h1 with a unit per lane: the chain's own fragments [h1_0 | h1_k - h1_0], transposed by identity products in the chain's k order
        //      (exact: the pieces are fp16 numbers) — a quarter of the products of Z1 = enc W1^T over again and no LDS traffic; dW2 += (G v2)^T h1.
        //      Layer 1's ReLU mask in this layout is read off the transposed values (hi + lo > 0: equal to z1 > 0 unless 0 < h1 < 2^-40 of its bound)
        unsigned long long m1n = 0ull;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            half8 idp;                                   // B operand with a one at the k position of unit 16 mi + q16: lane group q16 / 4, element 4 (mi & 1) + q16 % 4
#pragma unroll
            for (int e = 0; e < 8; ++e) idp[e] = (half_t)((lg == (q16 >> 2) && e == 4 * (mi & 1) + (q16 & 3)) ? 1.f : 0.f);
            floatx4 th[4], tl[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                th[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1h[mi >> 1][nb], idp, z, 0, 0, 0);
                tl[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1l[mi >> 1][nb], idp, z, 0, 0, 0);
            }
            half8 hh[2], hl[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nb = 2 * kk + hb;
                        float whole = th[nb][r] + tl[nb][r];
                        if (FD && nb > 0) whole += th[0][r] + tl[0][r];
                        m1n |= (unsigned long long)(whole > 0.f) << ((mi * 4 + nb) * 4 + r);
                        hh[kk][4 * hb + r] = (half_t)th[nb][r];
                        hl[kk][4 * hb + r] = (half_t)tl[nb][r];
                    }
#pragma unroll
            for (int mj = 0; mj < 4; ++mj)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) dw2[mj][mi] = tfm_mma3(gvh[mj][kk], gvl[mj][kk], hh[kk], hl[kk], dw2[mj][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- v2 in the chain layout (from the mask): the row operand of u1 = v2 W2 in the swapped orientation
        half8 vh[2][4], vl[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            floatx4 w3v[2][O];
            tfm_w3_chain<O>(w3s, 2 * t, lg, w3v[0]);
            tfm_w3_chain<O>(w3s, 2 * t + 1, lg, w3v[1]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float x[8];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = 0.f;
#pragma unroll
                        for (int o = 0; o < O; ++o) v = fmaf(dn[nb][o], w3v[hb][o][r], v);
                        x[4 * hb + r] = ((m2t >> (((2 * t + hb) * 4 + nb) * 4 + r)) & 1ull) ? v * sV2 : 0.f;
                        if (FD && nb > 0) x[4 * hb + r] -= ((m2t >> (((2 * t + hb) * 4) * 4 + r)) & 1ull) ? v * sV2 : 0.f;
                    }
                tfm_split8(x, vh[t][nb], vl[t][nb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- u1 = v2 W2 in the swapped orientation, masked by layer 1's ReLU, times G / S: the gradient operand of dW1
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            floatx4 zn[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) zn[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const half8 ah = img[TFM_OFF_A2TH / 8 + (mb * 2 + t) * 64 + lane], al = img[TFM_OFF_A2TL / 8 + (mb * 2 + t) * 64 + lane];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) zn[nb] = tfm_mma3(vh[t][nb], vl[t][nb], ah, al, zn[nb]);
            }
            if (FD) { zn[1] += zn[0]; zn[2] += zn[0]; zn[3] += zn[0]; }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float x[8];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nb = 2 * kk + hb;
                        if (FD && nb == 0) {
                            const float u0 = ((m1n >> ((mb * 4) * 4 + r)) & 1ull) ? zn[0][r] : 0.f;
                            float t = gq[0][r] * u0;
#pragma unroll
                            for (int k2 = 1; k2 < 4; ++k2) t = fmaf(gq[k2][r], (((m1n >> ((mb * 4 + k2) * 4 + r)) & 1ull) ? zn[k2][r] : 0.f) - u0, t);
                            x[4 * hb + r] = t * cU;
                        } else {
                            x[4 * hb + r] = ((m1n >> ((mb * 4 + nb) * 4 + r)) & 1ull) ? gq[nb][r] * zn[nb][r] * cU : 0.f;
                        }
                    }
                tfm_split8(x, gvh[mb][kk], gvl[mb][kk]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- dW1 += (G u1)^T enc: the lookup fragments come back from LDS and are transposed by identity products
#pragma unroll
        for (int plane = 0; plane < 3; ++plane) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                half8 eh[2], el[2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    float xh[8], xl[8];
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        const int nb = 2 * kk + hb;
                        const half8 bh = encbuf[((plane * 4 + nb) * 2 + 0) * 64 + lane], bl = encbuf[((plane * 4 + nb) * 2 + 1) * 64 + lane];
                        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                        const floatx4 th = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ident[hf], z, 0, 0, 0);
                        const floatx4 tl = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ident[hf], z, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { xh[4 * hb + r] = th[r]; xl[4 * hb + r] = tl[r]; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) { eh[kk][e] = (half_t)xh[e]; el[kk][e] = (half_t)xl[e]; }
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) dw1[mi][2 * plane + hf] = tfm_mma3(gvh[mi][kk], gvl[mi][kk], eh[kk], el[kk], dw1[mi][2 * plane + hf]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- accumulators -> global: lane (q16, lg) of dw2[mj][mi] holds dW2[16 mj + 4 lg + r][16 mi + q16]; dw1[mi][cb]: dW1[16 mi + 4 lg + r][16 cb + q16];
    //      dw3[mb]: dW3[o = 4 lg + r][16 mb + q16] (outputs >= O are zero)
    const float k3 = S / (16384.f * sc[TFM_S_H2]), k2 = S / (sV2 * sc[TFM_S_H1]), k1 = S / (sc[TFM_S_U1] * sE);
#pragma unroll
    for (int mj = 0; mj < 4; ++mj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (lg == 0 && r < O) {
                const float v = dw3[mj][r] * k3;
                if (v != 0.f) atomicAdd(a.dw3 + r * TF_H + 16 * mj + q16, v);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const float v = dw2[mj][mi][r] * k2;
                if (v != 0.f) atomicAdd(a.dw2 + (16 * mj + 4 * lg + r) * TF_H + 16 * mi + q16, v);
            }
#pragma unroll
            for (int cb = 0; cb < 6; ++cb) {
                const float v = dw1[mj][cb][r] * k1;
                if (v != 0.f) atomicAdd(a.dw1 + (16 * mj + 4 * lg + r) * TF_NIN + 16 * cb + q16, v);
            }
        }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
int tfm_prepare(const float* planes_cl, int H, int W, const float* const* w6, float* prep, hipStream_t s) {
    (void)hipMemsetAsync(prep, 0, 64 * sizeof(float), s);
    const size_t units = (size_t)3 * (H + 2) * (W + 2) * 8;
    hipLaunchKernelGGL(tfm_pad_kernel, dim3(asd_grid_for((int64_t)units, 256)), dim3(256), 0, s, planes_cl, H, W, prep + TFM_PREP_FIXED, (unsigned*)prep);
    hipLaunchKernelGGL(tfm_prep_kernel, dim3(2), dim3(256), 0, s, w6[0], w6[1], w6[2], w6[3], w6[4], w6[5], prep);
    return ASD_OK;
}

int tfm_forward(const tf_geom g, const asd_field_cfg* cfg, const float* planes_cl, const float* const* w6, const float* prep, const float* points, int n, float* sdf,
                float* features, float* normal, float* fd_grad, hipStream_t s) {
    static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
    const size_t lds = (size_t)2 * TFM_FWD_HALVES * 2;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)tfm_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)tfm_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)tfm_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const bool fd = normal || fd_grad;
    const int tiles = asd_div_up(n, fd ? 16 : 64);
    int blocks = asd_div_up(tiles, 4);
    if (blocks > 512) blocks = 512;
    if (fd) {
        hipLaunchKernelGGL(tfm_fwd_kernel<0>, dim3(blocks), dim3(256), lds, s, g, *cfg, prep + TFM_PREP_FIXED, prep, w6[2], w6[5], points, n, sdf, features, normal, fd_grad);
    } else {
        hipLaunchKernelGGL(tfm_fwd_kernel<1>, dim3(blocks), dim3(256), lds / 2, s, g, *cfg, prep + TFM_PREP_FIXED, prep, w6[2], w6[5], points, n, sdf, features, normal, fd_grad);
        if (features)
            hipLaunchKernelGGL(tfm_fwd_kernel<2>, dim3(blocks), dim3(256), lds / 2, s, g, *cfg, prep + TFM_PREP_FIXED, prep, w6[2], w6[5], points, n, sdf, features, normal, fd_grad);
    }
    return ASD_OK;
}

// one chunk of the backward pass: feature-gradient rows (denc, pts) for the scatter and the heads' weight gradients
int tfm_backward_chunk(const tf_geom g, const asd_field_cfg* cfg, const float* planes_cl, const float* const* w6, const float* prep, const float* points,
                       const float* sdf, int i0, int nc, int npt, const float* d_sdf, const float* d_features, const float* d_normal, const float* d_fd_grad,
                       float* denc, float* pts, float* const* dw6, hipStream_t s) {
    static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
    auto ldsd = [](int) { return (size_t)TFM_HEAD_HALVES * 2; };
    auto ldsw = [](int O) { return (size_t)TFM_OFF_A1TH * 2 + 4 * (3 * 4 * 2 * 64 * 16) + 4 * (1 + O) * 64 * sizeof(float) + (size_t)O * TF_H * sizeof(float); };
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)tfm_bwd_weights_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw(1));
        (void)hipFuncSetAttribute((const void*)tfm_bwd_weights_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw(1));
        (void)hipFuncSetAttribute((const void*)tfm_bwd_weights_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw(3));
        (void)hipFuncSetAttribute((const void*)tfm_bwd_data_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd(1));
        (void)hipFuncSetAttribute((const void*)tfm_bwd_data_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd(1));
        (void)hipFuncSetAttribute((const void*)tfm_bwd_data_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd(3));
        attr = true;
    }
    tfm_bwd_args a;
    a.g = g; a.c = *cfg; a.planes = prep + TFM_PREP_FIXED; a.prep = prep; a.points = points; a.sdf = sdf; a.i0 = i0; a.n_chunk = nc; a.npt = npt;
    a.d_sdf = d_sdf; a.d_features = d_features; a.d_normal = d_normal; a.d_fd_grad = d_fd_grad; a.denc = denc; a.pts = pts;
    auto grid = [&](int per_tile) { int b = asd_div_up(asd_div_up(nc, per_tile), 4); return b > 512 ? 512 : b; };
    auto gridw = [&](int per_tile) { int b = asd_div_up(asd_div_up(nc, per_tile), 4); return b > 256 ? 256 : b; };
    a.w3 = w6[2]; a.dw1 = dw6[0]; a.dw2 = dw6[1]; a.dw3 = dw6[2];
    if (npt == 4) {
        hipLaunchKernelGGL((tfm_bwd_data_kernel<1, true>), dim3(grid(16)), dim3(256), ldsd(1), s, a);
        hipLaunchKernelGGL((tfm_bwd_weights_kernel<1, true>), dim3(gridw(16)), dim3(256), ldsw(1), s, a);
    } else {
        hipLaunchKernelGGL((tfm_bwd_data_kernel<1, false>), dim3(grid(64)), dim3(256), ldsd(1), s, a);
        hipLaunchKernelGGL((tfm_bwd_weights_kernel<1, false>), dim3(gridw(64)), dim3(256), ldsw(1), s, a);
    }
    if (d_features) {
        a.w3 = w6[5]; a.dw1 = dw6[3]; a.dw2 = dw6[4]; a.dw3 = dw6[5];
        hipLaunchKernelGGL((tfm_bwd_data_kernel<3, false>), dim3(grid(64)), dim3(256), ldsd(3), s, a);
        hipLaunchKernelGGL((tfm_bwd_weights_kernel<3, false>), dim3(gridw(64)), dim3(256), ldsw(3), s, a);
    }
    return ASD_OK;
}
