// tritx.hip — the tri-plane transformer generator at fp32-class accuracy on the fp16 matrix pipe (gfx950).
//
// Replaces, behind the C ABI of include/asd_hip.h, the library kernels under
//   custom/amortized/extern/triplane_transformer_modules.py:34-187  (ConditionModulationBlock x 12: LayerNorm -> cross-attention to the
//   77 text tokens -> LayerNorm -> self-attention over 3 x 32 x 32 tokens -> LayerNorm -> Linear/GELU/Linear; final LayerNorm;
//   ConvTranspose2d(768 -> 32, 2, 2)), trained in fp32 (`precision: 32`, asd_mv_triplane_transformer_10k.yaml:127).
//
// gfx950 has no reduced-precision fp32 matrix path (v_mfma_f32_*_f32 runs at the vector rate, 1/16 of fp16), so every fp32 operand is
// split into two fp16 planes, x * s = hi + lo with s a power of two PER ROW of the operand as the product sees it (the contraction runs
// along the row, so the scales factor out), and a product is hi.hi + hi.lo + lo.hi accumulated in fp32: 22 bits per operand at a third
// of the fp16 rate (csrc/conv3d.hip's arithmetic).  Linear layers run the three products as ONE K-concatenated fp16 GEMM
// ([hi | hi | lo] x [hi | lo | hi]^T, gemm_f16_kernel with an fp32 result) followed by an fp32 epilogue (row and column scales, bias,
// GELU, residual); attention is its own flash kernel on 32x32x16 MFMAs (attention part below).
//
// Roofline: MFMA for the GEMMs / attention (3 fp16 products per fp32-equivalent multiply-add), HBM for the split / epilogue / LayerNorm
// passes (4-10 B per element).
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_fp16.h>

#include "asd_common.h"

namespace {

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;

// power-of-two scale that puts `amax` into [2^14, 2^15); 1 for an all-zero row
__device__ __forceinline__ float tx_scale_for(float amax) {
    if (!(amax > 0.f)) return 1.f;
    int e;
    (void)frexpf(amax, &e);                 // amax = m * 2^e, m in [0.5, 1)
    e = 15 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.f, e);
}
__device__ __forceinline__ float tx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float tx_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void tx_split(float xs, h16& hi, h16& lo) {
    hi = (h16)xs;
    lo = (h16)(xs - (float)hi);
}

// ---- operand planes -------------------------------------------------------------------------------------------------------------------
// rows of X [R, C] (ld) -> plane [R, 3 C] fp16: LAYOUT 0 (the A side of a product) [hi | hi | lo], 1 (the W side) [hi | lo | hi];
// inv[r] = 1 / scale of row r.  One wave per row.
template <int LAYOUT>
__global__ __launch_bounds__(256) void tx_split_rows_kernel(const float* __restrict__ x, int R, int C, int ld, h16* __restrict__ plane,
                                                            float* __restrict__ inv) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* row = x + (size_t)r * ld;
    float amax = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    const float s = tx_scale_for(tx_wave_max(amax));
    if (lane == 0) inv[r] = 1.f / s;
    h16* dst = plane + (size_t)r * 3 * C;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        h16 h0, h1, h2, h3, l0, l1, l2, l3;
        tx_split(v.x * s, h0, l0); tx_split(v.y * s, h1, l1); tx_split(v.z * s, h2, l2); tx_split(v.w * s, h3, l3);
        const h16x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
        *reinterpret_cast<h16x4*>(dst + c) = hi;
        *reinterpret_cast<h16x4*>(dst + C + c) = LAYOUT == 0 ? hi : lo;
        *reinterpret_cast<h16x4*>(dst + 2 * C + c) = LAYOUT == 0 ? lo : hi;
    }
}

// column max |x| (as uint bits: monotone for non-negative floats) and column sums of X [R, C] (C % 4 == 0): a block covers 64 columns (16
// float4 lanes) x `rows_per_block` rows (16 row lanes); one atomic per column and block
__global__ __launch_bounds__(256) void tx_colstat_kernel(const float* __restrict__ x, int R, int C, int ld, int rows_per_block,
                                                         unsigned* __restrict__ colmax, float* __restrict__ colsum) {
    __shared__ float4 smax[16][16], ssum[16][16];
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s = m;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 16) {
            const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + c);
            m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    smax[rl][cg] = m; ssum[rl][cg] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int col = threadIdx.x, g = col >> 2, e = col & 3;
        float mm = 0.f, sm = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 a = smax[q][g], b = ssum[q][g];
            mm = fmaxf(mm, e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : a.w);
            sm += e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w;
        }
        const int cc = blockIdx.x * 64 + col;
        if (cc < C) {
            if (colmax) atomicMax(colmax + cc, __float_as_uint(mm));
            if (colsum) atomicAdd(colsum + cc, sm);
        }
    }
}

// X [R, C] -> plane of X^T: [C, 3 Rp] fp16 (Rp >= R, zero beyond R), per-row (= column of X) scales from colmax; 64 x 64 tiles through LDS
template <int LAYOUT>
__global__ __launch_bounds__(256) void tx_split_cols_kernel(const float* __restrict__ x, int R, int C, int ld, int Rp,
                                                            const unsigned* __restrict__ colmax, h16* __restrict__ plane,
                                                            float* __restrict__ inv) {
    __shared__ h16 thi[64][66], tlo[64][66];      // [col][row], padded
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = c0 + cl;
    const float s = c < C ? tx_scale_for(__uint_as_float(colmax[c])) : 1.f;
    if (blockIdx.y == 0 && q == 0 && c < C) inv[c] = 1.f / s;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int rl = q * 16 + i, r = r0 + rl;
        const float v = (r < R && c < C) ? x[(size_t)r * ld + c] : 0.f;
        h16 hi, lo;
        tx_split(v * s, hi, lo);
        thi[cl][rl] = hi; tlo[cl][rl] = lo;
    }
    __syncthreads();
    // write: row (c0 + col) of the plane, columns r0 .. r0 + 63 of each of the three K segments
    const int rl = threadIdx.x & 63;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int col = q * 16 + i;
        if (c0 + col >= C) continue;
        h16* dst = plane + (size_t)(c0 + col) * 3 * Rp + r0 + rl;
        const h16 hi = thi[col][rl], lo = tlo[col][rl];
        dst[0] = hi;
        dst[Rp] = LAYOUT == 0 ? hi : lo;
        dst[2 * (size_t)Rp] = LAYOUT == 0 ? lo : hi;
    }
}

// ---- fp32 epilogue of a split product -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tx_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float tx_gelu_grad(float u) {
    return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
}
// y[i][j] = f(acc[i][j] * ia[i] * iw[j] + bias[j]) + residual[i][j]
//   mode 0: f = identity;  1: f = GELU, the pre-activation goes to `aux`;  2: f(v) = v * GELU'(aux[i][j])  (gradient through a GELU)
// nslab > 1: `acc` holds the split-K slabs [nslab][M][N] of the product (asd_gemm_f16 with partials_only): they are summed here instead of
// by a reduction launch of their own
__global__ __launch_bounds__(256) void tx_epilogue_kernel(const float* __restrict__ acc, int nslab, int M, int N, const float* __restrict__ ia,
                                                          const float* __restrict__ iw, const float* __restrict__ bias, int mode,
                                                          float* __restrict__ aux, int ld_aux, const float* __restrict__ residual, int ldr,
                                                          float* __restrict__ y, int ldy) {
    const int n4 = N / 4;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < (size_t)M * n4; t += (size_t)gridDim.x * 256) {
        const int i = (int)(t / n4), j = (int)(t % n4) * 4;
        float4 v = *reinterpret_cast<const float4*>(acc + (size_t)i * N + j);
        for (int z = 1; z < nslab; ++z) {
            const float4 u = *reinterpret_cast<const float4*>(acc + ((size_t)z * M + i) * N + j);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const float a = ia[i];
        const float4 w = *reinterpret_cast<const float4*>(iw + j);
        v.x *= a * w.x; v.y *= a * w.y; v.z *= a * w.z; v.w *= a * w.w;
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + j);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (mode == 1) {
            *reinterpret_cast<float4*>(aux + (size_t)i * ld_aux + j) = v;
            v.x = tx_gelu(v.x); v.y = tx_gelu(v.y); v.z = tx_gelu(v.z); v.w = tx_gelu(v.w);
        } else if (mode == 2) {
            const float4 u = *reinterpret_cast<const float4*>(aux + (size_t)i * ld_aux + j);
            v.x *= tx_gelu_grad(u.x); v.y *= tx_gelu_grad(u.y); v.z *= tx_gelu_grad(u.z); v.w *= tx_gelu_grad(u.w);
        }
        if (residual) {
            const float4 r = *reinterpret_cast<const float4*>(residual + (size_t)i * ldr + j);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(y + (size_t)i * ldy + j) = v;
    }
}

// the A-operand planes of a row that a wave holds in registers (float4 v[i] = columns 4 lane + 256 i): what tx_split_rows_kernel<0> would
// write, without the row going through memory again
__device__ __forceinline__ void tx_row_planes(const float4 (&v)[4], int D, int lane, float amax_lane, h16* __restrict__ dst, float* __restrict__ inv) {
    const float s = tx_scale_for(tx_wave_max(amax_lane));
    if (lane == 0) *inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            h16 h0, h1, h2, h3, l0, l1, l2, l3;
            tx_split(v[i].x * s, h0, l0); tx_split(v[i].y * s, h1, l1); tx_split(v[i].z * s, h2, l2); tx_split(v[i].w * s, h3, l3);
            const h16x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
            *reinterpret_cast<h16x4*>(dst + c) = hi;
            *reinterpret_cast<h16x4*>(dst + D + c) = hi;
            *reinterpret_cast<h16x4*>(dst + 2 * D + c) = lo;
        }
    }
}

// ---- LayerNorm (fp32, one wave per row, D % 4 == 0, D <= 1024) -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tx_layernorm_fwd_kernel(const float* __restrict__ x, int M, int D, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                               float* __restrict__ stats /*[M,2] mean, rstd*/,
                                                               h16* __restrict__ plane /* optional: the row's A-operand planes [M, 3D] (hi | hi | lo) */,
                                                               float* __restrict__ inv /* ... and 1 / its scale */) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float* row = x + (size_t)r * D;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        v[i] = c < D ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = tx_wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(tx_wave_sum(q) / (float)D + eps);
    if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4*>(y + (size_t)r * D + c) = o;
            v[i] = o;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (plane) tx_row_planes(v, D, lane, amax, plane + (size_t)r * 3 * D, inv + r);
}

// dx = rstd * (g - mean(g) - xhat * mean(g xhat)), g = dy * gamma, (+ residual gradient `dres`); dgamma += sum dy xhat, dbeta += sum dy.
// A wave walks rows r = wave, wave + n_waves, ... and keeps the dgamma / dbeta terms of ITS columns in registers: one atomic per column
// and wave at the end.
__global__ __launch_bounds__(256) void tx_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, int M, int D, const float* __restrict__ dres,
                                                               float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               h16* __restrict__ plane /* optional: A-operand planes of the dx rows [M, 3D] */,
                                                               float* __restrict__ inv) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    float4 g4[4], ag[4], ab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        g4[i] = c < D ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int r = wave; r < M; r += n_waves) {
        const float mean = stats[2 * r], rstd = stats[2 * r + 1];
        float4 xh[4], gg[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * D + c), d = *reinterpret_cast<const float4*>(dy + (size_t)r * D + c);
                xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                gg[i] = make_float4(d.x * g4[i].x, d.y * g4[i].y, d.z * g4[i].z, d.w * g4[i].w);
                s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
                s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
                ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
            } else {
                xh[i] = gg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float c1 = tx_wave_sum(s1) / (float)D, c2 = tx_wave_sum(s2) / (float)D;
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < D) {
                float4 o = make_float4(rstd * (gg[i].x - c1 - xh[i].x * c2), rstd * (gg[i].y - c1 - xh[i].y * c2),
                                       rstd * (gg[i].z - c1 - xh[i].z * c2), rstd * (gg[i].w - c1 - xh[i].w * c2));
                if (dres) {
                    const float4 e = *reinterpret_cast<const float4*>(dres + (size_t)r * D + c);
                    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                }
                *reinterpret_cast<float4*>(dx + (size_t)r * D + c) = o;
                gg[i] = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
        if (plane) tx_row_planes(gg, D, lane, amax, plane + (size_t)r * 3 * D, inv + r);
    }
    // the block's four waves meet in LDS: one atomic per column and block
    __shared__ float4 sg[4][256], sb[4][256];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) { sg[w][lane + 64 * i] = ag[i]; sb[w][lane + 64 * i] = ab[i]; }
    __syncthreads();
    {
        const int t = threadIdx.x;          // float4 slot t <-> columns (t & 63) * 4 + 256 * (t >> 6)
        const int c = (t & 63) * 4 + 256 * (t >> 6);
        if (c < D) {
            float4 a = sg[0][t], b = sb[0][t];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                a.x += sg[q][t].x; a.y += sg[q][t].y; a.z += sg[q][t].z; a.w += sg[q][t].w;
                b.x += sb[q][t].x; b.y += sb[q][t].y; b.z += sb[q][t].z; b.w += sb[q][t].w;
            }
            atomicAdd(dgamma + c, a.x); atomicAdd(dgamma + c + 1, a.y); atomicAdd(dgamma + c + 2, a.z); atomicAdd(dgamma + c + 3, a.w);
            atomicAdd(dbeta + c, b.x); atomicAdd(dbeta + c + 1, b.y); atomicAdd(dbeta + c + 2, b.z); atomicAdd(dbeta + c + 3, b.w);
        }
    }
}

// ---- attention (head dim 48, fp32-class) ------------------------------------------------------------------------------------------------
// Operand planes (tx_attn_prep_kernel): Q, K row-major fp16 hi / lo [H][L][48] (96-byte rows), V TRANSPOSED [H][48][Lp]; one power-of-two
// scale per (tensor, head) from tx_attn_absmax_kernel.  Key rows beyond Lk are zero and masked in the kernels.
//
// The flash kernels run S^T = K Q^T on v_mfma_f32_32x32x16_f16: a lane then owns ONE query (column l & 31) and 16 of the 32 keys of a block
// (rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)), so the row maximum / sum of the softmax are 16 in-lane operations plus one exchange with lane
// l ^ 32, and — after conversion to fp16 hi / lo — those same registers ARE the B operand of O^T = V^T P^T in a permuted key order
// (k-step u, position p of half h is key 16 u + (p & 3) + 8 (p >> 2) + 4 h), in which the V^T fragments are loaded: P never moves.
#define TX_HD 48
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct tx_absmax_args { const float* x[4]; int L[4], ld[4]; };
// max |x| per (tensor, head): grid (blocks of 256 rows, H, tensors); 12 float4 lanes per row, one atomic per block
__global__ __launch_bounds__(256) void tx_attn_absmax_kernel(const tx_absmax_args a, int H, unsigned* __restrict__ amax /*[tensors][H]*/) {
    __shared__ float part[4];
    const int h = blockIdx.y, z = blockIdx.z, L = a.L[z], ld = a.ld[z];
    const float* x = a.x[z];
    const int r0 = blockIdx.x * 256;
    if (r0 >= L) return;
    float m = 0.f;
    for (int t = threadIdx.x; t < 256 * (TX_HD / 4); t += 256) {
        const int r = r0 + t / (TX_HD / 4), c = (t % (TX_HD / 4)) * 4;
        if (r < L) {
            const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + h * TX_HD + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    m = tx_wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax + z * H + h, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

// row-major planes [H][Lp][48] (rows >= L zero): hi, lo
__global__ __launch_bounds__(256) void tx_attn_prep_rows_kernel(const float* __restrict__ x, int L, int Lp, int ld, int H, const unsigned* __restrict__ amax,
                                                                h16* __restrict__ hi, h16* __restrict__ lo) {
    const int h = blockIdx.y;
    const float s = tx_scale_for(__uint_as_float(amax[h]));
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;      // one thread per 4 elements
    const int r = (int)(t / (TX_HD / 4)), c = (int)(t % (TX_HD / 4)) * 4;
    if (r >= Lp) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + h * TX_HD + c);
    h16 h0, h1, h2, h3, l0, l1, l2, l3;
    tx_split(v.x * s, h0, l0); tx_split(v.y * s, h1, l1); tx_split(v.z * s, h2, l2); tx_split(v.w * s, h3, l3);
    const size_t o = ((size_t)h * Lp + r) * TX_HD + c;
    *reinterpret_cast<h16x4*>(hi + o) = h16x4{h0, h1, h2, h3};
    *reinterpret_cast<h16x4*>(lo + o) = h16x4{l0, l1, l2, l3};
}

// transposed planes [H][48][Lp] (columns >= L zero): 64 rows of x per block through LDS
__global__ __launch_bounds__(256) void tx_attn_prep_cols_kernel(const float* __restrict__ x, int L, int Lp, int ld, int H, const unsigned* __restrict__ amax,
                                                                h16* __restrict__ hi, h16* __restrict__ lo) {
    __shared__ h16 thi[TX_HD][66], tlo[TX_HD][66];
    const int h = blockIdx.y, r0 = blockIdx.x * 64;
    const float s = tx_scale_for(__uint_as_float(amax[h]));
    for (int t = threadIdx.x; t < 64 * TX_HD; t += 256) {
        const int rl = t / TX_HD, c = t % TX_HD;
        const float v = r0 + rl < L ? x[(size_t)(r0 + rl) * ld + h * TX_HD + c] : 0.f;
        h16 a, b;
        tx_split(v * s, a, b);
        thi[c][rl] = a; tlo[c][rl] = b;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 64 * TX_HD; t += 256) {
        const int c = t / 64, rl = t % 64;
        if (r0 + rl < Lp) {
            const size_t o = ((size_t)h * TX_HD + c) * Lp + r0 + rl;
            hi[o] = thi[c][rl]; lo[o] = tlo[c][rl];
        }
    }
}

__device__ __forceinline__ f32x16 tx_mfma(const h16x8 a, const h16x8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// hi.hi + hi.lo + lo.hi
__device__ __forceinline__ f32x16 tx_mfma3(const h16x8 ah, const h16x8 al, const h16x8 bh, const h16x8 bl, f32x16 c) {
    c = tx_mfma(ah, bh, c);
    c = tx_mfma(ah, bl, c);
    c = tx_mfma(al, bh, c);
    return c;
}
__device__ __forceinline__ int tx_row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }     // C / D row of register r

// ---- streamed tiles through LDS: the four waves of a block walk the same sequence of 32-token tiles -----------------------------------
// A row-major plane tile is 32 rows x 48 halfs (contiguous in the plane), kept at a 112-byte row pitch: ds_read_b128 of 32 rows at one
// offset is conflict-free (28 banks per row: rows i and i + 16 would collide, and they never sit in the same 16-lane group).  A transposed
// plane tile is 48 rows (d) x 32 tokens at a 72-byte pitch (18 banks: 32 distinct even banks for the 8-byte reads of 32 rows).
#define TX_RP 56            // halfs per row of a row-major tile in LDS
#define TX_TP 36            // halfs per row of a transposed tile in LDS
#define TX_RT (32 * TX_RP)  // halfs per row-major tile
#define TX_TT (TX_HD * TX_TP)
// plane k of a stage is loaded by threads 0..191, one 16-byte chunk each (192 chunks per tile of either kind)
template <int NR, int NT>
struct TxStage {
    uint4 r[NR + NT];
};
// rsrc[k]: first half of the tile in row-major plane k;  tsrc[k]: element (d = 0, first token) of the tile in transposed plane k (row pitch Lp)
#define TX_STAGE_LOAD(ST, NR_, NT_, RSRC, TSRC, LP)                                                                                   \
    if (threadIdx.x < 192) {                                                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < (NR_); ++k_) (ST).r[k_] = *reinterpret_cast<const uint4*>((RSRC)[k_] + threadIdx.x * 8); \
        _Pragma("unroll") for (int k_ = 0; k_ < (NT_); ++k_)                                                                          \
            (ST).r[(NR_) + k_] = *reinterpret_cast<const uint4*>((TSRC)[k_] + (size_t)(threadIdx.x >> 2) * (LP) + (threadIdx.x & 3) * 8); \
    }
#define TX_STAGE_STORE(ST, NR_, NT_, LDS)                                                                                             \
    if (threadIdx.x < 192) {                                                                                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < (NR_); ++k_)                                                                          \
            *reinterpret_cast<uint4*>((LDS) + k_ * TX_RT + (threadIdx.x / 6) * TX_RP + (threadIdx.x % 6) * 8) = (ST).r[k_];         \
        _Pragma("unroll") for (int k_ = 0; k_ < (NT_); ++k_) {                                                                        \
            h16* d_ = (LDS) + (NR_) * TX_RT + k_ * TX_TT + (threadIdx.x >> 2) * TX_TP + (threadIdx.x & 3) * 8;                       \
            *reinterpret_cast<uint2*>(d_) = make_uint2((ST).r[(NR_) + k_].x, (ST).r[(NR_) + k_].y);                                   \
            *reinterpret_cast<uint2*>(d_ + 4) = make_uint2((ST).r[(NR_) + k_].z, (ST).r[(NR_) + k_].w);                             \
        }                                                                                                                             \
    }
__device__ __forceinline__ h16x8 tx_lds_row(const h16* tile, int row, int t, int half) {       // A / B fragment of k-step t from a row-major tile
    return *reinterpret_cast<const h16x8*>(tile + row * TX_RP + 16 * t + 8 * half);
}
__device__ __forceinline__ h16x8 tx_lds_perm(const h16* tile, int d, int u, int half) {        // permuted-order fragment from a transposed tile
    const h16x4 a = *reinterpret_cast<const h16x4*>(tile + d * TX_TP + 16 * u + 4 * half), b = *reinterpret_cast<const h16x4*>(tile + d * TX_TP + 16 * u + 4 * half + 8);
    return h16x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

// 2^x on v_exp_f32 alone (arguments here are <= 0 or -inf: no range reduction / denormal fix-up needed)
__device__ __forceinline__ float tx_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// the value of lane l ^ 32 (v_permlane32_swap: VALU, no LDS round trip)
__device__ __forceinline__ float tx_other_half(float v) {
    auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(((threadIdx.x >> 5) & 1) ? t[0] : t[1]);
}
#define TX_PSCALE 16384.f       // probabilities (<= 1) are split at 2^14

// O [Lq, ldo] (fp32), lse2 [H][Lq] = log2 sum_j 2^(s2_ij), s2 = (q . k) log2(e) / sqrt(48).  grid (ceil(Lq / 128), H), 4 waves x 32 queries
__global__ __launch_bounds__(256, 3) void tx_attn_fwd_kernel(const h16* __restrict__ qh, const h16* __restrict__ ql, const h16* __restrict__ kh,
                                                          const h16* __restrict__ kl, const h16* __restrict__ vth, const h16* __restrict__ vtl,
                                                          const unsigned* __restrict__ amax /*[3][H]: q, k, v*/, int Lq, int Lqp, int Lk, int Lkp, int H,
                                                          float* __restrict__ o, int ldo, float* __restrict__ lse2,
                                                          float* __restrict__ part /* gridDim.z > 1: [z][Lq][H 48] unnormalised O, then [z][H][Lq] m, [z][H][Lq] l */) {
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 128 + wave * 32;       // (waves past Lq work on zero rows of the padded planes and store nothing: every wave meets the barriers)
    const int j = lane & 31, half = lane >> 5;
    const float c1 = 1.4426950408889634f * 0.14433756729740643f /* log2(e) / sqrt(48) */ /
                     (tx_scale_for(__uint_as_float(amax[h])) * tx_scale_for(__uint_as_float(amax[H + h])));
    const float cv = 1.f / (tx_scale_for(__uint_as_float(amax[2 * H + h])) * TX_PSCALE);
    // Q fragments: B operand of S^T = K Q^T — lane (query j, half): d = 16 t + 8 half .. + 7
    h16x8 bqh[3], bql[3];
    {
        const size_t qrow = ((size_t)h * Lqp + q0 + j) * TX_HD + 8 * half;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            bqh[t] = *reinterpret_cast<const h16x8*>(qh + qrow + 16 * t);
            bql[t] = *reinterpret_cast<const h16x8*>(ql + qrow + 16 * t);
        }
    }
    f32x16 ot[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ot[0][r] = 0.f; ot[1][r] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const h16* kbase_h = kh + (size_t)h * Lkp * TX_HD, * kbase_l = kl + (size_t)h * Lkp * TX_HD;
    const h16* vbase_h = vth + (size_t)h * TX_HD * Lkp, * vbase_l = vtl + (size_t)h * TX_HD * Lkp;
    // streamed per key block: K rows (hi, lo) and V^T (hi, lo), double-buffered in LDS
    constexpr int STAGE = 2 * TX_RT + 2 * TX_TT;
    __shared__ __attribute__((aligned(16))) h16 lds[2 * STAGE];
    TxStage<2, 2> st_regs;
    // the key range is cut over blockIdx.z (more, shorter blocks: 1.5 waves per SIMD otherwise); the pieces meet in tx_attn_fwd_combine_kernel
    const int k_per = ((Lk + (int)gridDim.z - 1) / (int)gridDim.z + 31) & ~31;
    const int k_begin = (int)blockIdx.z * k_per, k_end = min(Lk, k_begin + k_per);
    if (k_begin < k_end) {
        const h16* const rs[2] = {kbase_h + (size_t)k_begin * TX_HD, kbase_l + (size_t)k_begin * TX_HD};
        const h16* const ts[2] = {vbase_h + k_begin, vbase_l + k_begin};
        TX_STAGE_LOAD(st_regs, 2, 2, rs, ts, Lkp);
        TX_STAGE_STORE(st_regs, 2, 2, lds);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 32, buf ^= 1) {
        const bool more = k0 + 32 < k_end;
        if (more) {
            const h16* const rs[2] = {kbase_h + (size_t)(k0 + 32) * TX_HD, kbase_l + (size_t)(k0 + 32) * TX_HD};
            const h16* const ts[2] = {vbase_h + k0 + 32, vbase_l + k0 + 32};
            TX_STAGE_LOAD(st_regs, 2, 2, rs, ts, Lkp);
        }
        const h16* tile = lds + buf * STAGE;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) st = tx_mfma3(tx_lds_row(tile, j, t, half), tx_lds_row(tile + TX_RT, j, t, half), bqh[t], bql[t], st);
        float mb = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s2 = st[r] * c1;
            if (k0 + 32 > Lk && k0 + tx_row_of(r, half) >= Lk) s2 = -INFINITY;
            st[r] = s2;
            mb = fmaxf(mb, s2);
        }
        mb = fmaxf(mb, tx_other_half(mb));
        const float m_new = fmaxf(m, mb);
        const float alpha = tx_exp2(m - m_new);           // m = -inf on the first block: 0
        float lb = 0.f;
        h16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = tx_exp2(st[r] - m_new);
            lb += p;
            h16 a, b;
            tx_split(p * TX_PSCALE, a, b);
            ph[r >> 3][r & 7] = a; pl[r >> 3][r & 7] = b;
        }
        lb += tx_other_half(lb);
        l = l * alpha + lb;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ot[0][r] *= alpha; ot[1][r] *= alpha; }
        // O^T += V^T P^T: A = V^T rows d = 32 mt + j in the permuted key order of P's registers
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int d = 32 * mt + j;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                h16x8 ah, al;
                if (d < TX_HD) {
                    ah = tx_lds_perm(tile + 2 * TX_RT, d, u, half);
                    al = tx_lds_perm(tile + 2 * TX_RT + TX_TT, d, u, half);
                } else {
                    ah = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    al = ah;
                }
                ot[mt] = tx_mfma3(ah, al, ph[u], pl[u], ot[mt]);
            }
        }
        if (more) { TX_STAGE_STORE(st_regs, 2, 2, lds + (buf ^ 1) * STAGE); }
        __syncthreads();
    }
    // O[q][h * 48 + d] = O^T[d][q] * cv / l
    if (q0 + j < Lq) {
        const bool whole = gridDim.z == 1;
        const float f = whole ? cv / l : cv;
        float* orow = whole ? o + (size_t)(q0 + j) * ldo + h * TX_HD : part + ((size_t)blockIdx.z * Lq + q0 + j) * (H * TX_HD) + h * TX_HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * mt + tx_row_of(r, half);
                if (d < TX_HD) orow[d] = ot[mt][r] * f;
            }
        if (half == 0) {
            if (whole) lse2[(size_t)h * Lq + q0 + j] = m + log2f(l);
            else {
                float* ml = part + (size_t)gridDim.z * Lq * H * TX_HD;
                ml[((size_t)blockIdx.z * H + h) * Lq + q0 + j] = m;
                ml[((size_t)(gridDim.z + blockIdx.z) * H + h) * Lq + q0 + j] = l;
            }
        }
    }
}

// pieces of a key-split forward: O = sum_z O_z 2^(m_z - M) / sum_z l_z 2^(m_z - M); one thread per (query, head, 4 channels)
__global__ __launch_bounds__(256) void tx_attn_fwd_combine_kernel(const float* __restrict__ part, int Z, int Lq, int H, float* __restrict__ o, int ldo,
                                                                  float* __restrict__ lse2) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t % (TX_HD / 4)) * 4;
    const size_t qh = t / (TX_HD / 4);
    const int h = (int)(qh % H), q = (int)(qh / H);
    if (q >= Lq) return;
    const float* ml = part + (size_t)Z * Lq * H * TX_HD;
    float M = -INFINITY;
    for (int z = 0; z < Z; ++z) M = fmaxf(M, ml[((size_t)z * H + h) * Lq + q]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < Z; ++z) {
        const float w = tx_exp2(ml[((size_t)z * H + h) * Lq + q] - M);
        L += ml[((size_t)(Z + z) * H + h) * Lq + q] * w;
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)z * Lq + q) * (H * TX_HD) + h * TX_HD + c);
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
    }
    const float inv = 1.f / L;
    *reinterpret_cast<float4*>(o + (size_t)q * ldo + h * TX_HD + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    if (c == 0) lse2[(size_t)h * Lq + q] = M + log2f(L);
}
// out[r][h 48 + c] (ld) = sum_z part[z][r][H 48]
__global__ __launch_bounds__(256) void tx_attn_sum_parts_kernel(const float* __restrict__ part, int Z, int L, int HD_all, float* __restrict__ out, int ld) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(t % (HD_all / 4)) * 4;
    const size_t r = t / (HD_all / 4);
    if (r >= (size_t)L) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < Z; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)z * L + r) * HD_all + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + r * ld + c) = acc;
}

// D[h][q] = sum_d dO[q][h 48 + d] O[q][h 48 + d]: one thread per (query, head)
__global__ __launch_bounds__(256) void tx_attn_rowdot_kernel(const float* __restrict__ d_o, int ldd, const float* __restrict__ o, int ldo, int Lq, int H,
                                                             float* __restrict__ dsum /*[H][Lq]*/) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Lq * H) return;
    const int q = t / H, h = t % H;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < TX_HD; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(d_o + (size_t)q * ldd + h * TX_HD + c), y = *reinterpret_cast<const float4*>(o + (size_t)q * ldo + h * TX_HD + c);
        a += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    dsum[(size_t)h * Lq + q] = a;
}

// 8 halves of a transposed plane row at the permuted positions of k-step u: columns c0 + 16 u + 4 half + {0..3} and + 8
__device__ __forceinline__ h16x8 tx_load_perm(const h16* __restrict__ rowp, int c0, int u, int half) {
    const h16x4 a = *reinterpret_cast<const h16x4*>(rowp + c0 + 16 * u + 4 * half), b = *reinterpret_cast<const h16x4*>(rowp + c0 + 16 * u + 4 * half + 8);
    return h16x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

struct tx_attn_bwd_args {
    const h16 *qh, *ql, *kh, *kl, *vh, *vl, *doh, *dol;     // row-major planes [H][Lp][48]
    const h16 *qth, *qtl, *doth, *dotl, *kth, *ktl;         // transposed planes [H][48][Lp]
    const unsigned* amax;                                   // [4][H]: q, k, v, dO
    const float *lse2, *dsum;                               // [H][Lq]
    int Lq, Lqp, Lk, Lkp, H;
    float *dq, *dk, *dv;
    int lddq, lddk, lddv;
    float* part;        // gridDim.z > 1: partial sums of the split range
};
// static bound of |P (dP - D)| per head: |dP_ij| <= 48 max|dO| max|V|, |D_i| <= 48 max|dO| max|O| and |O| <= max|V| (convex combination)
__device__ __forceinline__ float tx_ds_scale(const unsigned* amax, int H, int h) {
    return tx_scale_for(96.f * __uint_as_float(amax[3 * H + h]) * __uint_as_float(amax[2 * H + h]));
}

// dK, dV: a wave owns 32 keys and walks the query blocks.  S = Q K^T (rows = queries, column = the lane's key), P = 2^(S2 - lse2),
// dV^T += dO^T P, dP = dO V^T, dS = P (dP - D), dK^T += Q^T dS.  grid (ceil(Lk / 128), H)
__global__ __launch_bounds__(256, 2) void tx_attn_bwd_kv_kernel(const tx_attn_bwd_args a) {
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 128 + wave * 32;        // (waves past Lk work on zero rows of the padded planes and store nothing)
    const int j = lane & 31, half = lane >> 5, H = a.H;
    const float sq = tx_scale_for(__uint_as_float(a.amax[h])), sk = tx_scale_for(__uint_as_float(a.amax[H + h]));
    const float sv = tx_scale_for(__uint_as_float(a.amax[2 * H + h])), sdo = tx_scale_for(__uint_as_float(a.amax[3 * H + h]));
    const float sds = tx_ds_scale(a.amax, H, h);
    const float c1 = 1.4426950408889634f * 0.14433756729740643f / (sq * sk), cdp = 1.f / (sdo * sv);
    // resident B operands: K^T and V^T columns = this lane's key, d range 16 t + 8 half .. + 7
    h16x8 bkh[3], bkl[3], bvh[3], bvl[3];
    {
        const size_t krow = ((size_t)h * a.Lkp + k0 + j) * TX_HD + 8 * half;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            bkh[t] = *reinterpret_cast<const h16x8*>(a.kh + krow + 16 * t); bkl[t] = *reinterpret_cast<const h16x8*>(a.kl + krow + 16 * t);
            bvh[t] = *reinterpret_cast<const h16x8*>(a.vh + krow + 16 * t); bvl[t] = *reinterpret_cast<const h16x8*>(a.vl + krow + 16 * t);
        }
    }
    f32x16 dvt[2], dkt[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[0][r] = dvt[1][r] = dkt[0][r] = dkt[1][r] = 0.f; }
    const h16 *qbh = a.qh + (size_t)h * a.Lqp * TX_HD, *qbl = a.ql + (size_t)h * a.Lqp * TX_HD;
    const h16 *dbh = a.doh + (size_t)h * a.Lqp * TX_HD, *dbl = a.dol + (size_t)h * a.Lqp * TX_HD;
    const h16 *qtbh = a.qth + (size_t)h * TX_HD * a.Lqp, *qtbl = a.qtl + (size_t)h * TX_HD * a.Lqp;
    const h16 *dtbh = a.doth + (size_t)h * TX_HD * a.Lqp, *dtbl = a.dotl + (size_t)h * TX_HD * a.Lqp;
    const float *lse = a.lse2 + (size_t)h * a.Lq, *dsm = a.dsum + (size_t)h * a.Lq;
    const bool key_ok = k0 + j < a.Lk;
    // the query range is cut over blockIdx.z (balance: four short blocks per slot instead of 384 long ones on 512 slots; the
    // cross-attention's 77 keys: 16 blocks otherwise); the partial sums land in a_.part and are added up by tx_attn_sum_parts_kernel
    const int q_per = ((a.Lq + (int)gridDim.z - 1) / (int)gridDim.z + 31) & ~31;
    const int q_begin = (int)blockIdx.z * q_per, q_end = min(a.Lq, q_begin + q_per);
    // streamed per query block: Q and dO rows (row-major) and Q^T, dO^T (transposed), hi / lo each, double-buffered in LDS
    constexpr int STAGE = 4 * TX_RT + 4 * TX_TT;
    __shared__ __attribute__((aligned(16))) h16 lds[2 * STAGE];
    __shared__ float row_stats[2][64];            // [stage][lse2 of the tile's 32 queries | D of the same]
    TxStage<4, 4> st_regs;
    float stat_reg = 0.f;
    const int stat_i = (int)threadIdx.x - 192;   // 0..31: lse2, 32..63: D
    auto stat_load = [&](int q0_) {
        if (stat_i >= 0) {
            const int qi = q0_ + (stat_i & 31);
            stat_reg = qi < a.Lq ? (stat_i < 32 ? lse[qi] : dsm[qi]) : 0.f;
        }
    };
    if (q_begin < q_end) {
        stat_load(q_begin);
        if (stat_i >= 0) row_stats[0][stat_i] = stat_reg;
        const h16* const rs[4] = {qbh + (size_t)q_begin * TX_HD, qbl + (size_t)q_begin * TX_HD, dbh + (size_t)q_begin * TX_HD, dbl + (size_t)q_begin * TX_HD};
        const h16* const ts[4] = {qtbh + q_begin, qtbl + q_begin, dtbh + q_begin, dtbl + q_begin};
        TX_STAGE_LOAD(st_regs, 4, 4, rs, ts, a.Lqp);
        TX_STAGE_STORE(st_regs, 4, 4, lds);
    }
    __syncthreads();
    int buf = 0;
    for (int q0 = q_begin; q0 < q_end; q0 += 32, buf ^= 1) {
        const bool more = q0 + 32 < q_end;
        if (more) {
            const size_t ro = (size_t)(q0 + 32) * TX_HD;
            const h16* const rs[4] = {qbh + ro, qbl + ro, dbh + ro, dbl + ro};
            const h16* const ts[4] = {qtbh + q0 + 32, qtbl + q0 + 32, dtbh + q0 + 32, dtbl + q0 + 32};
            TX_STAGE_LOAD(st_regs, 4, 4, rs, ts, a.Lqp);
            stat_load(q0 + 32);
        }
        const h16* tile = lds + buf * STAGE;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            s = tx_mfma3(tx_lds_row(tile, j, t, half), tx_lds_row(tile + TX_RT, j, t, half), bkh[t], bkl[t], s);
            dp = tx_mfma3(tx_lds_row(tile + 2 * TX_RT, j, t, half), tx_lds_row(tile + 3 * TX_RT, j, t, half), bvh[t], bvl[t], dp);
        }
        h16x8 ph[2], pl[2], gh[2], gl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = tx_row_of(r, half), qi = q0 + ql;
            float p = 0.f, g = 0.f;
            if (qi < a.Lq && key_ok) {
                p = tx_exp2(s[r] * c1 - row_stats[buf][ql]);
                g = p * (dp[r] * cdp - row_stats[buf][32 + ql]);
            }
            h16 x, y;
            tx_split(p * TX_PSCALE, x, y);
            ph[r >> 3][r & 7] = x; pl[r >> 3][r & 7] = y;
            tx_split(g * sds, x, y);
            gh[r >> 3][r & 7] = x; gl[r >> 3][r & 7] = y;
        }
        const h16* tt = tile + 4 * TX_RT;        // Q^T hi, lo, dO^T hi, lo
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int d = 32 * mt + j;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                h16x8 ah, al, ch, cl;
                if (d < TX_HD) {
                    ch = tx_lds_perm(tt, d, u, half); cl = tx_lds_perm(tt + TX_TT, d, u, half);
                    ah = tx_lds_perm(tt + 2 * TX_TT, d, u, half); al = tx_lds_perm(tt + 3 * TX_TT, d, u, half);
                } else {
                    ah = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    al = ah; ch = ah; cl = ah;
                }
                dvt[mt] = tx_mfma3(ah, al, ph[u], pl[u], dvt[mt]);
                dkt[mt] = tx_mfma3(ch, cl, gh[u], gl[u], dkt[mt]);
            }
        }
        if (more) {
            TX_STAGE_STORE(st_regs, 4, 4, lds + (buf ^ 1) * STAGE);
            if (stat_i >= 0) row_stats[buf ^ 1][stat_i] = stat_reg;
        }
        __syncthreads();
    }
    if (key_ok) {
        const float fv = 1.f / (sdo * TX_PSCALE), fk = 0.14433756729740643f / (sq * sds);
        const bool whole = gridDim.z == 1;
        const size_t pstride = (size_t)a.Lk * H * TX_HD;       // one [Lk][H 48] partial
        float* dvrow = whole ? a.dv + (size_t)(k0 + j) * a.lddv + h * TX_HD : a.part + (size_t)(gridDim.z + blockIdx.z) * pstride + (size_t)(k0 + j) * (H * TX_HD) + h * TX_HD;
        float* dkrow = whole ? a.dk + (size_t)(k0 + j) * a.lddk + h * TX_HD : a.part + (size_t)blockIdx.z * pstride + (size_t)(k0 + j) * (H * TX_HD) + h * TX_HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * mt + tx_row_of(r, half);
                if (d < TX_HD) { dvrow[d] = dvt[mt][r] * fv; dkrow[d] = dkt[mt][r] * fk; }
            }
    }
}

// dQ: a wave owns 32 queries and walks the key blocks in the forward's orientation: S^T = K Q^T, dP^T = V dO^T (column = the lane's query),
// dS^T = P^T (dP^T - D), dQ^T += K^T dS^T.  grid (ceil(Lq / 128), H)
__global__ __launch_bounds__(256, 2) void tx_attn_bwd_q_kernel(const tx_attn_bwd_args a) {
    const int h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = blockIdx.x * 128 + wave * 32;        // (waves past Lq: zero rows, nothing stored)
    const int j = lane & 31, half = lane >> 5, H = a.H;
    const float sq = tx_scale_for(__uint_as_float(a.amax[h])), sk = tx_scale_for(__uint_as_float(a.amax[H + h]));
    const float sv = tx_scale_for(__uint_as_float(a.amax[2 * H + h])), sdo = tx_scale_for(__uint_as_float(a.amax[3 * H + h]));
    const float sds = tx_ds_scale(a.amax, H, h);
    const float c1 = 1.4426950408889634f * 0.14433756729740643f / (sq * sk), cdp = 1.f / (sdo * sv);
    h16x8 bqh[3], bql[3], bdh[3], bdl[3];
    {
        const size_t qrow = ((size_t)h * a.Lqp + q0 + j) * TX_HD + 8 * half;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            bqh[t] = *reinterpret_cast<const h16x8*>(a.qh + qrow + 16 * t); bql[t] = *reinterpret_cast<const h16x8*>(a.ql + qrow + 16 * t);
            bdh[t] = *reinterpret_cast<const h16x8*>(a.doh + qrow + 16 * t); bdl[t] = *reinterpret_cast<const h16x8*>(a.dol + qrow + 16 * t);
        }
    }
    const bool q_ok = q0 + j < a.Lq;
    const float lse = q_ok ? a.lse2[(size_t)h * a.Lq + q0 + j] : 0.f, dsm = q_ok ? a.dsum[(size_t)h * a.Lq + q0 + j] : 0.f;
    f32x16 dqt[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dqt[0][r] = dqt[1][r] = 0.f; }
    const h16 *kbh = a.kh + (size_t)h * a.Lkp * TX_HD, *kbl = a.kl + (size_t)h * a.Lkp * TX_HD;
    const h16 *vbh = a.vh + (size_t)h * a.Lkp * TX_HD, *vbl = a.vl + (size_t)h * a.Lkp * TX_HD;
    const h16 *ktbh = a.kth + (size_t)h * TX_HD * a.Lkp, *ktbl = a.ktl + (size_t)h * TX_HD * a.Lkp;
    // streamed per key block: K and V rows (row-major) and K^T (transposed), hi / lo each
    constexpr int STAGE = 4 * TX_RT + 2 * TX_TT;
    __shared__ __attribute__((aligned(16))) h16 lds[2 * STAGE];
    TxStage<4, 2> st_regs;
    const int k_per = ((a.Lk + (int)gridDim.z - 1) / (int)gridDim.z + 31) & ~31;
    const int k_begin = (int)blockIdx.z * k_per, k_end = min(a.Lk, k_begin + k_per);
    if (k_begin < k_end) {
        const size_t ro = (size_t)k_begin * TX_HD;
        const h16* const rs[4] = {kbh + ro, kbl + ro, vbh + ro, vbl + ro};
        const h16* const ts[2] = {ktbh + k_begin, ktbl + k_begin};
        TX_STAGE_LOAD(st_regs, 4, 2, rs, ts, a.Lkp);
        TX_STAGE_STORE(st_regs, 4, 2, lds);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 32, buf ^= 1) {
        const bool more = k0 + 32 < k_end;
        if (more) {
            const size_t ro = (size_t)(k0 + 32) * TX_HD;
            const h16* const rs[4] = {kbh + ro, kbl + ro, vbh + ro, vbl + ro};
            const h16* const ts[2] = {ktbh + k0 + 32, ktbl + k0 + 32};
            TX_STAGE_LOAD(st_regs, 4, 2, rs, ts, a.Lkp);
        }
        const h16* tile = lds + buf * STAGE;
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            st = tx_mfma3(tx_lds_row(tile, j, t, half), tx_lds_row(tile + TX_RT, j, t, half), bqh[t], bql[t], st);
            dpt = tx_mfma3(tx_lds_row(tile + 2 * TX_RT, j, t, half), tx_lds_row(tile + 3 * TX_RT, j, t, half), bdh[t], bdl[t], dpt);
        }
        h16x8 gh[2], gl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float g = 0.f;
            if (q_ok && k0 + tx_row_of(r, half) < a.Lk) {
                const float p = tx_exp2(st[r] * c1 - lse);
                g = p * (dpt[r] * cdp - dsm);
            }
            h16 x, y;
            tx_split(g * sds, x, y);
            gh[r >> 3][r & 7] = x; gl[r >> 3][r & 7] = y;
        }
        const h16* tt = tile + 4 * TX_RT;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int d = 32 * mt + j;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                h16x8 ah, al;
                if (d < TX_HD) {
                    ah = tx_lds_perm(tt, d, u, half); al = tx_lds_perm(tt + TX_TT, d, u, half);
                } else {
                    ah = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    al = ah;
                }
                dqt[mt] = tx_mfma3(ah, al, gh[u], gl[u], dqt[mt]);
            }
        }
        if (more) { TX_STAGE_STORE(st_regs, 4, 2, lds + (buf ^ 1) * STAGE); }
        __syncthreads();
    }
    if (q_ok) {
        const float fq = 0.14433756729740643f / (sk * sds);
        float* row = gridDim.z == 1 ? a.dq + (size_t)(q0 + j) * a.lddq + h * TX_HD
                                    : a.part + ((size_t)blockIdx.z * a.Lq + q0 + j) * (H * TX_HD) + h * TX_HD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * mt + tx_row_of(r, half);
                if (d < TX_HD) row[d] = dqt[mt][r] * fq;
            }
    }
}

// per-column bounds (as float bits) of the activations whose size follows from the weights alone:
//   LayerNorm output: |xhat| <= sqrt(D - 1), so |n_c| <= sqrt(D - 1) |gamma_c| + |beta_c|;   GELU(fc1(n3)): |h_j| <= |u_j| <= sum_c |W1[j][c]| B3_c + |b1_j|
// They stand in for the column maxima when those activations are split into fp16 planes (a loose power-of-two scale costs nothing: fp16 is
// a floating format, the second plane still carries the next 11 bits).  grid: 3 blocks for the LayerNorms + F / 4 blocks (one wave per hidden unit)
__global__ __launch_bounds__(256) void tx_bounds_kernel(const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
                                                        const float* __restrict__ g3, const float* __restrict__ b3, const float* __restrict__ w1, const float* __restrict__ fb1,
                                                        int D, int F, unsigned* __restrict__ bn1, unsigned* __restrict__ bn2, unsigned* __restrict__ bn3,
                                                        unsigned* __restrict__ bh) {
    const float rt = 1.01f * sqrtf((float)(D - 1));
    if (blockIdx.x < 3) {
        const float* g = blockIdx.x == 0 ? g1 : blockIdx.x == 1 ? g2 : g3;
        const float* b = blockIdx.x == 0 ? b1 : blockIdx.x == 1 ? b2 : b3;
        unsigned* o = blockIdx.x == 0 ? bn1 : blockIdx.x == 1 ? bn2 : bn3;
        for (int c = threadIdx.x; c < D; c += 256) o[c] = __float_as_uint(rt * fabsf(g[c]) + fabsf(b[c]));
        return;
    }
    const int j = (blockIdx.x - 3) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= F) return;
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a += fabsf(w1[(size_t)j * D + c]) * (rt * fabsf(g3[c]) + fabsf(b3[c]));
    a = tx_wave_sum(a);
    if (lane == 0) bh[j] = __float_as_uint(1.01f * a + fabsf(fb1[j]));
}

// a page of zeros per device for asd_gemm_f16's out-of-range rows
const void* tx_zero_page() {
    static std::mutex mu;
    static void* pages[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        if (hipMalloc(&pages[dev], 256) != hipSuccess || hipMemset(pages[dev], 0, 256) != hipSuccess) return nullptr;
    }
    return pages[dev];
}

// Zeroed counter words (absmax / column-max cells, the cells atomics add into).  Stand-alone entry points clear their own cells with a
// memset; inside asd_tritx_fwd / _bwd ONE memset at the start of the pass clears a pool the ~25 entries per layer then carve up
// (281 memset launches per step otherwise: 1.3 ms of GPU time and as many launch boundaries).
// Zero fill as a KERNEL: inside asd_tritx_fwd / _bwd the passes are captured into HIP graphs (generators._TritxBuffers), and a captured
// hipMemsetAsync node did not survive a second replay on ROCm 7.2 (first replay correct, every later one left NaNs behind it:
// tools/tritx_graph_check.py, round 6); a fill kernel replays like every other launch of the pass.
__global__ __launch_bounds__(256) void tx_fill_zero_kernel(unsigned* __restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline void tx_memset0(void* p, size_t bytes, hipStream_t s) {
    const size_t words = (bytes + 3) / 4;
    if (words == 0) return;
    const unsigned blocks = (unsigned)((words + 255) / 256 > 1024 ? 1024 : (words + 255) / 256);
    hipLaunchKernelGGL(tx_fill_zero_kernel, dim3(blocks), dim3(256), 0, s, (unsigned*)p, words);
}
struct TxPool { unsigned* p; size_t left; bool outputs_zeroed; };
thread_local TxPool tx_pool = {nullptr, 0, false};
unsigned* tx_zeroed(unsigned* own, size_t words, hipStream_t s) {
    const size_t w = (words + 63) & ~(size_t)63;
    if (tx_pool.p && tx_pool.left >= w) {
        unsigned* r = tx_pool.p;
        tx_pool.p += w; tx_pool.left -= w;
        return r;
    }
    tx_memset0(own, words * 4, s);
    return own;
}

inline int64_t tx_al(int64_t floats) { return (floats + 63) & ~(int64_t)63; }
inline int tx_rp(int r) { return (r + 63) & ~63; }      // rows of a transposed operand padded to the GEMM's k-step

// tile configuration (1-based index into csrc/gemm.hip's table) and split-K per product shape, from tools/tritx_gemm_sweep.py on the shipped
// model's shapes (12 layers x 768 wide x 3072 tokens; profiles/r05_tritx_gemm_sweep.txt); other shapes: cost model, no split
struct TxPlan { int M, N, K3, cfg, sk; };
const TxPlan tx_plans[] = {
    {3072, 768, 2304, 1, 1}, {3072, 2304, 2304, 4, 1}, {3072, 3072, 2304, 8, 1}, {3072, 768, 9216, 4, 3}, {3072, 768, 6912, 2, 3}, {77, 1536, 3072, 1, 6},
    {3072, 128, 2304, 1, 4}, {3072, 768, 384, 1, 1}, {768, 768, 9216, 1, 6}, {2304, 768, 9216, 2, 4}, {768, 3072, 9216, 4, 3}, {1536, 1024, 384, 1, 1},
    {768, 128, 9216, 1, 8},
};
void tx_plan(int M, int N, int K3, int* cfg, int* sk) {
    *cfg = 0; *sk = 1;
    for (const TxPlan& p : tx_plans)
        if (p.M == M && p.N == N && p.K3 == K3) { *cfg = p.cfg; *sk = p.sk; return; }
}
inline int64_t tx_gemm_ws_floats(int M, int N, int K3) {
    int cfg, sk;
    tx_plan(M, N, K3, &cfg, &sk);
    return sk > 1 ? tx_al((int64_t)sk * M * N) : 0;
}

// planeA [M, 3K] . planeW [N, 3K]^T (fp32 result of the three fp16 products): *res / *nslab tell the epilogue where the result is — c32
// [M, N] (nslab 1) or the unreduced split-K slabs [nslab][M][N] in `slabs` (tx_gemm_ws_floats(M, N, K3) floats)
int tx_gemm(const h16* pa, const h16* pw, int M, int N, int K3, float* c32, float* slabs, const float** res, int* nslab, hipStream_t s) {
    asd_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = pa; a.W = pw; a.C = c32;
    a.M = M; a.N = N; a.K = K3;
    a.lda = K3; a.ldw = K3; a.ldc = N;
    a.out_f32 = 1;
    int cfg, sk;
    tx_plan(M, N, K3, &cfg, &sk);
    a.split_k = sk; a.tile_cfg = cfg; a.workspace = slabs;
    a.partials_only = sk > 1;
    *res = sk > 1 ? slabs : c32;
    *nslab = sk;
    a.zero_page = tx_zero_page();
    if (!a.zero_page) { asd_set_error("tritx: could not allocate the zero page"); return ASD_ERR_LAUNCH; }
    return asd_gemm_f16(&a, s);
}

}  // namespace

extern "C" {

// ---- C ABI: the building blocks (each one is tested on its own against float64, tests/test_gpu_tritx.py) --------------------------------
int asd_tx_pack_weight(const float* w, int32_t N, int32_t K, void* plane_w, float* inv_w, void* plane_wt, float* inv_wt, float* ws, void* stream) {
    ASD_CHECK_ARG(w && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (plane_w) {
        ASD_CHECK_ARG(inv_w, "inv_w missing");
        hipLaunchKernelGGL((tx_split_rows_kernel<1>), dim3(asd_div_up(N, 4)), dim3(256), 0, s, w, N, K, K, (h16*)plane_w, inv_w);
    }
    if (plane_wt) {      // rows of W^T = columns of W [N, K]: K rows of 3 * Np halfs
        ASD_CHECK_ARG(inv_wt && ws, "inv_wt / ws missing");
        unsigned* colmax = tx_zeroed(reinterpret_cast<unsigned*>(ws), (size_t)K, s);
        hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(K, 64), asd_div_up(N, 256)), dim3(256), 0, s, w, N, K, K, 256, colmax, (float*)nullptr);
        hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(K, 64), tx_rp(N) / 64), dim3(256), 0, s, w, N, K, K, tx_rp(N), colmax, (h16*)plane_wt, inv_wt);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int64_t asd_tx_linear_workspace(int32_t M, int32_t N, int32_t K) {
    // A plane [M, 3K] halfs + row scales + fp32 product [M, N]
    return tx_al((int64_t)M * 3 * K / 2 + 64) + tx_al(M) + tx_al((int64_t)M * N) + tx_gemm_ws_floats(M, N, 3 * K);
}

// y [M, N] (ldy) = f(x [M, K] (ldx) . W^T + bias) + residual, W given as packed plane [N, 3K] + inv_w [N] (asd_tx_pack_weight; pass the
// W^T plane for an input gradient).  mode 0 identity, 1 GELU (pre-activation saved to aux), 2 multiply by GELU'(aux) (aux [M, N], ld N)
// planes_ready: the A-operand planes and row scales of x already sit in ws where this function would put them (written by the LayerNorm
// kernel that produced x: tx_linear_planes_of)
static int tx_linear_core(const float* x, int32_t M, int32_t K, int32_t ldx, const void* plane_w, const float* inv_w, int32_t N, const float* bias, int32_t mode,
                          float* aux, const float* residual, int32_t ldr, float* y, int32_t ldy, float* ws, bool planes_ready, void* stream);
static inline h16* tx_linear_planes_of(float* ws) { return reinterpret_cast<h16*>(ws); }
static inline float* tx_linear_inv_of(float* ws, int M, int K) { return ws + tx_al((int64_t)M * 3 * K / 2 + 64); }
int asd_tx_linear(const float* x, int32_t M, int32_t K, int32_t ldx, const void* plane_w, const float* inv_w, int32_t N, const float* bias, int32_t mode,
                  float* aux, const float* residual, int32_t ldr, float* y, int32_t ldy, float* ws, void* stream) {
    return tx_linear_core(x, M, K, ldx, plane_w, inv_w, N, bias, mode, aux, residual, ldr, y, ldy, ws, false, stream);
}
static int tx_linear_core(const float* x, int32_t M, int32_t K, int32_t ldx, const void* plane_w, const float* inv_w, int32_t N, const float* bias, int32_t mode,
                          float* aux, const float* residual, int32_t ldr, float* y, int32_t ldy, float* ws, bool planes_ready, void* stream) {
    ASD_CHECK_ARG(x && plane_w && inv_w && y && ws && M > 0 && N > 0 && K > 0, "null argument");
    ASD_CHECK_ARG(K % 64 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (mode == 0 || aux), "K % 64, N % 4, leading dimensions % 4; aux for GELU modes");
    hipStream_t s = (hipStream_t)stream;
    h16* pa = reinterpret_cast<h16*>(ws);
    float* ia = ws + tx_al((int64_t)M * 3 * K / 2 + 64);
    float* c32 = ia + tx_al(M);
    if (!planes_ready) hipLaunchKernelGGL((tx_split_rows_kernel<0>), dim3(asd_div_up(M, 4)), dim3(256), 0, s, x, M, K, ldx, pa, ia);
    const float* res;
    int nslab;
    const int rc = tx_gemm(pa, (const h16*)plane_w, M, N, 3 * K, c32, c32 + tx_al((int64_t)M * N), &res, &nslab, s);
    if (rc != ASD_OK) return rc;
    hipLaunchKernelGGL(tx_epilogue_kernel, dim3(asd_grid_for((int64_t)M * N / 4, 256)), dim3(256), 0, s, res, nslab, M, N, ia, inv_w, bias, mode, aux, N, residual, ldr, y, ldy);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int64_t asd_tx_wgrad_workspace(int32_t M, int32_t N, int32_t K) {
    const int64_t Mp = tx_rp(M);
    return tx_al((int64_t)N * 3 * Mp / 2 + 64) + tx_al((int64_t)K * 3 * Mp / 2 + 64) + tx_al(N) + tx_al(K) + tx_al(N) + tx_al(K) + tx_al((int64_t)N * K) +
           tx_gemm_ws_floats(N, K, 3 * (int)Mp);
}

// dw [N, K] = dy [M, N]^T . x [M, K]  (contraction over the M rows), db [N] = column sums of dy (optional)
// x_planes / x_inv: the transposed planes of x made earlier (the text tokens serve all twelve layers); x_bound: per-column bounds of |x|
// known without looking at x (LayerNorm outputs: sqrt(D - 1) |gamma_c| + |beta_c|; ...) — either spares the statistics pass over x
// accumulate: dw += the product (the epilogue reads the old value as its residual: no staging buffer, no add launch), db += its column sums
static int tx_wgrad_core(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* dw, float* db, float* ws,
                         const h16* x_planes, const float* x_inv, const unsigned* x_bound, void* stream, bool accumulate = false) {
    ASD_CHECK_ARG(dy && (x || x_planes) && dw && ws && M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int Mp = tx_rp(M);
    h16* pa = reinterpret_cast<h16*>(ws);                                   // dy^T [N, 3 Mp]
    float* p = ws + tx_al((int64_t)N * 3 * Mp / 2 + 64);
    h16* pw = reinterpret_cast<h16*>(p);                                    // x^T [K, 3 Mp]
    p += tx_al((int64_t)K * 3 * Mp / 2 + 64);
    float* ia = p; p += tx_al(N);
    float* iw = p; p += tx_al(K);
    unsigned* cmax_a = reinterpret_cast<unsigned*>(p); p += tx_al(N);
    p += tx_al(K);
    float* c32 = p;
    cmax_a = tx_zeroed(cmax_a, (size_t)(tx_al(N) + tx_al(K)), s);
    unsigned* cmax_w = cmax_a + tx_al(N);
    if (db && !tx_pool.outputs_zeroed && !accumulate) tx_memset0(db, (size_t)N * 4, s);
    hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(N, 64), asd_div_up(M, 256)), dim3(256), 0, s, dy, M, N, ldy, 256, cmax_a, db);
    hipLaunchKernelGGL((tx_split_cols_kernel<0>), dim3(asd_div_up(N, 64), Mp / 64), dim3(256), 0, s, dy, M, N, ldy, Mp, cmax_a, pa, ia);
    if (!x_planes) {
        if (!x_bound) hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(K, 64), asd_div_up(M, 256)), dim3(256), 0, s, x, M, K, ldx, 256, cmax_w, (float*)nullptr);
        hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(K, 64), Mp / 64), dim3(256), 0, s, x, M, K, ldx, Mp, x_bound ? x_bound : cmax_w, pw, iw);
    }
    const float* res;
    int nslab;
    const int rc = tx_gemm(pa, x_planes ? x_planes : pw, N, K, 3 * Mp, c32, c32 + tx_al((int64_t)N * K), &res, &nslab, s);
    if (rc != ASD_OK) return rc;
    hipLaunchKernelGGL(tx_epilogue_kernel, dim3(asd_grid_for((int64_t)N * K / 4, 256)), dim3(256), 0, s, res, nslab, N, K, ia, x_planes ? x_inv : iw, (const float*)nullptr, 0,
                       (float*)nullptr, 0, accumulate ? (const float*)dw : (const float*)nullptr, K, dw, K);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}
int asd_tx_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* dw, float* db, float* ws, void* stream) {
    return tx_wgrad_core(dy, ldy, x, ldx, M, N, K, dw, db, ws, nullptr, nullptr, nullptr, stream);
}

static int tx_layernorm_fwd_core(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y, float* stats, h16* plane, float* inv,
                                 void* stream) {
    ASD_CHECK_ARG(x && gamma && beta && y && stats && M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "LayerNorm: D % 4 == 0, D <= 1024");
    hipLaunchKernelGGL(tx_layernorm_fwd_kernel, dim3(asd_div_up(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M, D, gamma, beta, eps, y, stats, plane, inv);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}
int asd_tx_layernorm_fwd(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y, float* stats, void* stream) {
    return tx_layernorm_fwd_core(x, M, D, gamma, beta, eps, y, stats, nullptr, nullptr, stream);
}

// dx = LayerNorm input gradient (+ dres); dgamma / dbeta are ACCUMULATED (+=: the caller zeroes them once per backward pass)
static int tx_layernorm_bwd_core(const float* dy, const float* x, const float* stats, const float* gamma, int32_t M, int32_t D, const float* dres, float* dx,
                                 float* dgamma, float* dbeta, h16* plane, float* inv, void* stream);
int asd_tx_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int32_t M, int32_t D, const float* dres, float* dx,
                         float* dgamma, float* dbeta, void* stream) {
    return tx_layernorm_bwd_core(dy, x, stats, gamma, M, D, dres, dx, dgamma, dbeta, nullptr, nullptr, stream);
}
static int tx_layernorm_bwd_core(const float* dy, const float* x, const float* stats, const float* gamma, int32_t M, int32_t D, const float* dres, float* dx,
                                 float* dgamma, float* dbeta, h16* plane, float* inv, void* stream) {
    ASD_CHECK_ARG(dy && x && stats && gamma && dx && dgamma && dbeta && M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "LayerNorm: D % 4 == 0, D <= 1024");
    static const int rows_per_wave = getenv("ASD_TX_LN_ROWS") ? atoi(getenv("ASD_TX_LN_ROWS")) : 8;      // tools/tritx_time.py: 81 / 44 / 29 / 28 / 40 us at 1 / 2 / 4 / 8 / 16 rows per wave (the per-block atomics of dgamma / dbeta dominate)
    int grid = asd_div_up(M, 4 * (rows_per_wave > 0 ? rows_per_wave : 8));
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(tx_layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, M, D, dres, dx, dgamma, dbeta, plane, inv);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}


// ---- attention ---------------------------------------------------------------------------------------------------------------------------
static inline int tx_lp(int L) { return (L + 127) & ~127; }
// range splits: enough blocks for ~3 per CU (forward / dQ: keys; dK, dV: queries), pieces of whole 32-token tiles
static inline int tx_split_for(int blocks, int target, int tiles) {
    int z = blocks > 0 ? target / blocks : 1;
    if (z > tiles / 4) z = tiles / 4;          // at least four tiles per piece
    if (z > 16) z = 16;
    return z < 1 ? 1 : z;
}
static inline int64_t tx_attn_part_floats(int64_t Lq, int64_t Lk, int64_t H) {
    const int64_t a = 16 * (Lq * H * TX_HD + 2 * H * Lq), b = 16 * 2 * Lk * H * TX_HD;     // at most 16 pieces
    const int64_t zq = tx_split_for((int)((Lq + 127) / 128 * H), 768, (int)((Lk + 31) / 32)), zk = tx_split_for((int)((Lk + 127) / 128 * H), 1536, (int)((Lq + 31) / 32));
    const int64_t nq = zq > 1 ? zq * (Lq * H * TX_HD + 2 * H * Lq) : 0, nk = zk > 1 ? zk * 2 * Lk * H * TX_HD : 0;
    (void)a; (void)b;
    return tx_al((nq > nk ? nq : nk) + 64);
}      // plane rows: whole 128-row blocks (32 per wave) so no load leaves the plane
int64_t asd_tx_attention_workspace(int32_t Lq, int32_t Lk, int32_t H) {
    // six planes of halfs (Q, K row-major hi / lo; V^T hi / lo), the backward's extra planes (Q^T, dO, dO^T) and 5 * H scale words
    const int64_t Lqp = tx_lp(Lq), Lkp = tx_lp(Lk);
    return tx_al((int64_t)H * TX_HD * (8 * Lqp + 6 * Lkp) / 2 + 256) + tx_al(8 * H) + tx_al((int64_t)H * Lq) + tx_attn_part_floats(Lq, Lk, H);
}

// o [Lq, ldo] = softmax(q k^T / sqrt(48)) v per head (head h = columns 48 h .. 48 h + 47 of q / k / v / o), lse2 [H, Lq] for the backward
int asd_tx_attention_fwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, int32_t Lq, int32_t Lk, int32_t H,
                         float* o, int32_t ldo, float* lse2, float* ws, void* stream) {
    ASD_CHECK_ARG(q && k && v && o && lse2 && ws && Lq > 0 && Lk > 0 && H > 0 && H <= 64, "null argument");
    ASD_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "leading dimensions must be multiples of 4 floats");
    hipStream_t s = (hipStream_t)stream;
    const int Lqp = tx_lp(Lq), Lkp = tx_lp(Lk);
    h16* p = reinterpret_cast<h16*>(ws);
    h16* qh = p; p += (size_t)H * Lqp * TX_HD;
    h16* ql = p; p += (size_t)H * Lqp * TX_HD;
    h16* kh = p; p += (size_t)H * Lkp * TX_HD;
    h16* kl = p; p += (size_t)H * Lkp * TX_HD;
    h16* vth = p; p += (size_t)H * Lkp * TX_HD;
    h16* vtl = p; p += (size_t)H * Lkp * TX_HD;
    unsigned* amax = tx_zeroed(reinterpret_cast<unsigned*>(ws + tx_al((int64_t)H * TX_HD * (8 * (int64_t)Lqp + 6 * (int64_t)Lkp) / 2 + 256)), (size_t)8 * H, s);
    {
        tx_absmax_args m;
        m.x[0] = q; m.L[0] = Lq; m.ld[0] = ldq; m.x[1] = k; m.L[1] = Lk; m.ld[1] = ldk; m.x[2] = v; m.L[2] = Lk; m.ld[2] = ldv; m.x[3] = nullptr; m.L[3] = 0; m.ld[3] = 0;
        hipLaunchKernelGGL(tx_attn_absmax_kernel, dim3(asd_div_up(Lq > Lk ? Lq : Lk, 256), H, 3), dim3(256), 0, s, m, H, amax);
    }
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, dim3(asd_div_up((int64_t)Lqp * (TX_HD / 4), 256), H), dim3(256), 0, s, q, Lq, Lqp, ldq, H, amax, qh, ql);
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, dim3(asd_div_up((int64_t)Lkp * (TX_HD / 4), 256), H), dim3(256), 0, s, k, Lk, Lkp, ldk, H, amax + H, kh, kl);
    hipLaunchKernelGGL(tx_attn_prep_cols_kernel, dim3(Lkp / 64, H), dim3(256), 0, s, v, Lk, Lkp, ldv, H, amax + 2 * H, vth, vtl);
    float* part = ws + tx_al((int64_t)H * TX_HD * (8 * (int64_t)Lqp + 6 * (int64_t)Lkp) / 2 + 256) + tx_al(8 * H) + tx_al((int64_t)H * Lq);
    const int zk = tx_split_for(asd_div_up(Lq, 128) * H, 768, asd_div_up(Lk, 32));
    hipLaunchKernelGGL(tx_attn_fwd_kernel, dim3(asd_div_up(Lq, 128), H, zk), dim3(256), 0, s, qh, ql, kh, kl, vth, vtl, amax, Lq, Lqp, Lk, Lkp, H, o, ldo, lse2, part);
    if (zk > 1)
        hipLaunchKernelGGL(tx_attn_fwd_combine_kernel, dim3(asd_div_up((int64_t)Lq * H * (TX_HD / 4), 256)), dim3(256), 0, s, part, zk, Lq, H, o, ldo, lse2);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}


// (dq, dk, dv) of asd_tx_attention_fwd from the gradient d_o of its output; o and lse2 are the forward's outputs
int asd_tx_attention_bwd(const float* q, int32_t ldq, const float* k, int32_t ldk, const float* v, int32_t ldv, const float* o, int32_t ldo,
                         const float* d_o, int32_t lddo, const float* lse2, int32_t Lq, int32_t Lk, int32_t H, float* dq, int32_t lddq, float* dk,
                         int32_t lddk, float* dv, int32_t lddv, float* ws, void* stream) {
    ASD_CHECK_ARG(q && k && v && o && d_o && lse2 && dq && dk && dv && ws && Lq > 0 && Lk > 0 && H > 0 && H <= 64, "null argument");
    ASD_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0, "leading dimensions must be multiples of 4 floats");
    hipStream_t s = (hipStream_t)stream;
    const int Lqp = tx_lp(Lq), Lkp = tx_lp(Lk);
    const size_t nq = (size_t)H * Lqp * TX_HD, nk = (size_t)H * Lkp * TX_HD;
    h16* p = reinterpret_cast<h16*>(ws);
    tx_attn_bwd_args a;
    h16 *qh = p, *ql = p + nq, *doh = p + 2 * nq, *dol = p + 3 * nq, *qth = p + 4 * nq, *qtl = p + 5 * nq, *doth = p + 6 * nq, *dotl = p + 7 * nq;
    p += 8 * nq;
    h16 *kh = p, *kl = p + nk, *vh = p + 2 * nk, *vl = p + 3 * nk, *kth = p + 4 * nk, *ktl = p + 5 * nk;
    float* tail = ws + tx_al((int64_t)H * TX_HD * (8 * (int64_t)Lqp + 6 * (int64_t)Lkp) / 2 + 256);
    unsigned* amax = tx_zeroed(reinterpret_cast<unsigned*>(tail), (size_t)8 * H, s);
    float* dsum = tail + tx_al(8 * H);
    {
        tx_absmax_args m;
        m.x[0] = q; m.L[0] = Lq; m.ld[0] = ldq; m.x[1] = k; m.L[1] = Lk; m.ld[1] = ldk; m.x[2] = v; m.L[2] = Lk; m.ld[2] = ldv; m.x[3] = d_o; m.L[3] = Lq; m.ld[3] = lddo;
        hipLaunchKernelGGL(tx_attn_absmax_kernel, dim3(asd_div_up(Lq > Lk ? Lq : Lk, 256), H, 4), dim3(256), 0, s, m, H, amax);
    }
    const dim3 gq(asd_div_up((int64_t)Lqp * (TX_HD / 4), 256), H), gk(asd_div_up((int64_t)Lkp * (TX_HD / 4), 256), H);
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, gq, dim3(256), 0, s, q, Lq, Lqp, ldq, H, amax, qh, ql);
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, gq, dim3(256), 0, s, d_o, Lq, Lqp, lddo, H, amax + 3 * H, doh, dol);
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, gk, dim3(256), 0, s, k, Lk, Lkp, ldk, H, amax + H, kh, kl);
    hipLaunchKernelGGL(tx_attn_prep_rows_kernel, gk, dim3(256), 0, s, v, Lk, Lkp, ldv, H, amax + 2 * H, vh, vl);
    hipLaunchKernelGGL(tx_attn_prep_cols_kernel, dim3(Lqp / 64, H), dim3(256), 0, s, q, Lq, Lqp, ldq, H, amax, qth, qtl);
    hipLaunchKernelGGL(tx_attn_prep_cols_kernel, dim3(Lqp / 64, H), dim3(256), 0, s, d_o, Lq, Lqp, lddo, H, amax + 3 * H, doth, dotl);
    hipLaunchKernelGGL(tx_attn_prep_cols_kernel, dim3(Lkp / 64, H), dim3(256), 0, s, k, Lk, Lkp, ldk, H, amax + H, kth, ktl);
    hipLaunchKernelGGL(tx_attn_rowdot_kernel, dim3(asd_div_up((int64_t)Lq * H, 256)), dim3(256), 0, s, d_o, lddo, o, ldo, Lq, H, dsum);
    a.qh = qh; a.ql = ql; a.kh = kh; a.kl = kl; a.vh = vh; a.vl = vl; a.doh = doh; a.dol = dol;
    a.qth = qth; a.qtl = qtl; a.doth = doth; a.dotl = dotl; a.kth = kth; a.ktl = ktl;
    a.amax = amax; a.lse2 = lse2; a.dsum = dsum;
    a.Lq = Lq; a.Lqp = Lqp; a.Lk = Lk; a.Lkp = Lkp; a.H = H;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.part = dsum + tx_al((int64_t)H * Lq);
    const int qsplit = tx_split_for(asd_div_up(Lk, 128) * H, 1536, asd_div_up(Lq, 32));
    hipLaunchKernelGGL(tx_attn_bwd_kv_kernel, dim3(asd_div_up(Lk, 128), H, qsplit), dim3(256), 0, s, a);
    if (qsplit > 1) {
        const int HDa = H * TX_HD;
        const dim3 g(asd_div_up((int64_t)Lk * (HDa / 4), 256));
        hipLaunchKernelGGL(tx_attn_sum_parts_kernel, g, dim3(256), 0, s, a.part, qsplit, Lk, HDa, dk, lddk);
        hipLaunchKernelGGL(tx_attn_sum_parts_kernel, g, dim3(256), 0, s, a.part + (size_t)qsplit * Lk * HDa, qsplit, Lk, HDa, dv, lddv);
    }
    const int ksplit = tx_split_for(asd_div_up(Lq, 128) * H, 768, asd_div_up(Lk, 32));
    hipLaunchKernelGGL(tx_attn_bwd_q_kernel, dim3(asd_div_up(Lq, 128), H, ksplit), dim3(256), 0, s, a);
    if (ksplit > 1)
        hipLaunchKernelGGL(tx_attn_sum_parts_kernel, dim3(asd_div_up((int64_t)Lq * (H * TX_HD / 4), 256)), dim3(256), 0, s, a.part, ksplit, Lq, H * TX_HD, dq, lddq);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}


// ---- the whole generator: forward / backward schedules -------------------------------------------------------------------------------------
// y [T, 4 C] (token t = (plane p, h, w) of the low-res grid R x R; column co * 4 + i * 2 + j) <-> channel-last planes [3][2R][2R][C]
__global__ __launch_bounds__(256) void tx_shuffle_kernel(const float* __restrict__ y, int R, int Cc, float* __restrict__ out, int inverse) {
    const size_t n = (size_t)3 * R * R * 4 * Cc;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        // t indexes the channel-last side: [p][Y][X][co]
        const int co = (int)(t % Cc);
        size_t r = t / Cc;
        const int X = (int)(r % (2 * R)); r /= 2 * R;
        const int Y = (int)(r % (2 * R));
        const int p = (int)(r / (2 * R));
        const size_t tok = ((size_t)p * R + (Y >> 1)) * R + (X >> 1);
        const size_t yi = tok * 4 * Cc + co * 4 + (Y & 1) * 2 + (X & 1);
        if (inverse) const_cast<float*>(y)[yi] = out[t]; else out[t] = y[yi];
    }
}
__global__ __launch_bounds__(256) void tx_add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n4, int accumulate) {
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n4; t += (size_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4*>(src)[t];
        if (accumulate) { const float4 b = reinterpret_cast<float4*>(dst)[t]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        reinterpret_cast<float4*>(dst)[t] = a;
    }
}

namespace {
struct TxDims {
    int layers, D, H, Dc, T, Tc, F, Cc, R, O;     // O = 4 * Cc: the deconvolution as a Linear
    explicit TxDims(const asd_tritx_desc& d) : layers(d.n_layers), D(d.dim), H(d.heads), Dc(d.cond_dim), T(3 * d.low_res * d.low_res), Tc(d.cond_tokens),
                                               F(d.hidden), Cc(d.out_channels), R(d.low_res), O(4 * d.out_channels) {}
};
inline int64_t plane_floats(int64_t rows, int64_t k) { return tx_al(rows * 3 * k / 2 + 64); }

// packed weights of one layer / of the head, as offsets (floats) into the packed buffer
struct TxPackLayer { int64_t caq_w, caq_iw, caq_t, caq_it, cakv_w, cakv_iw, cao_w, cao_iw, cao_t, cao_it, qkv_w, qkv_iw, qkv_t, qkv_it, sao_w, sao_iw, sao_t, sao_it,
                             fc1_w, fc1_iw, fc1_t, fc1_it, fc2_w, fc2_iw, fc2_t, fc2_it, bn1, bn2, bn3, bh, end; };
TxPackLayer tx_pack_layout(const TxDims& d, int64_t base) {
    TxPackLayer L;
    int64_t o = base;
    auto take = [&](int64_t n) { const int64_t at = o; o += tx_al(n); return at; };
    auto planes = [&](int N, int K, int64_t& w, int64_t& iw, int64_t* t, int64_t* it) {
        w = o; o += plane_floats(N, K); iw = take(N);
        if (t) { *t = o; o += plane_floats(K, tx_rp(N)); *it = take(K); }
    };
    planes(d.D, d.D, L.caq_w, L.caq_iw, &L.caq_t, &L.caq_it);
    planes(2 * d.D, d.Dc, L.cakv_w, L.cakv_iw, nullptr, nullptr);
    planes(d.D, d.D, L.cao_w, L.cao_iw, &L.cao_t, &L.cao_it);
    planes(3 * d.D, d.D, L.qkv_w, L.qkv_iw, &L.qkv_t, &L.qkv_it);
    planes(d.D, d.D, L.sao_w, L.sao_iw, &L.sao_t, &L.sao_it);
    planes(d.F, d.D, L.fc1_w, L.fc1_iw, &L.fc1_t, &L.fc1_it);
    planes(d.D, d.F, L.fc2_w, L.fc2_iw, &L.fc2_t, &L.fc2_it);
    L.bn1 = take(d.D); L.bn2 = take(d.D); L.bn3 = take(d.D); L.bh = take(d.F);        // column bounds of n1, n2, n3, gelu(fc1): tx_bounds_kernel
    L.end = o;
    return L;
}
struct TxPackHead { int64_t dc_w, dc_iw, dc_t, dc_it, end; };
TxPackHead tx_pack_head(const TxDims& d, int64_t base) {
    TxPackHead h;
    int64_t o = base;
    h.dc_w = o; o += plane_floats(d.O, d.D); h.dc_iw = o; o += tx_al(d.O);
    h.dc_t = o; o += plane_floats(d.D, tx_rp(d.O)); h.dc_it = o; o += tx_al(d.D);
    h.end = o;
    return h;
}

// saved activations of one layer (offsets in floats from the layer's base)
struct TxSave { int64_t x_in, st1, n1, q_ca, kv_ca, o_ca, lse_ca, x1, st2, n2, qkv, o_sa, lse_sa, x2, st3, n3, u, hmid, end; };
TxSave tx_save_layout(const TxDims& d) {
    TxSave s;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += tx_al(n); return at; };
    const int64_t TD = (int64_t)d.T * d.D;
    s.x_in = take(TD); s.st1 = take(2 * d.T); s.n1 = take(TD); s.q_ca = take(TD); s.kv_ca = take((int64_t)d.Tc * 2 * d.D); s.o_ca = take(TD);
    s.lse_ca = take((int64_t)d.H * d.T); s.x1 = take(TD); s.st2 = take(2 * d.T); s.n2 = take(TD); s.qkv = take(3 * TD); s.o_sa = take(TD);
    s.lse_sa = take((int64_t)d.H * d.T); s.x2 = take(TD); s.st3 = take(2 * d.T); s.n3 = take(TD); s.u = take((int64_t)d.T * d.F); s.hmid = take((int64_t)d.T * d.F);
    s.end = o;
    return s;
}
// per batch element: layers x TxSave, then x_final, stF, nF, y
inline int64_t tx_save_per_sample(const TxDims& d) {
    return (int64_t)d.layers * tx_save_layout(d).end + tx_al((int64_t)d.T * d.D) * 2 + tx_al(2 * d.T) + tx_al((int64_t)d.T * d.O);
}
inline int64_t tx_max64(int64_t a, int64_t b) { return a > b ? a : b; }
inline int64_t tx_op_ws(const TxDims& d) {
    int64_t w = 0;
    w = tx_max64(w, asd_tx_linear_workspace(d.T, d.F, d.D));
    w = tx_max64(w, asd_tx_linear_workspace(d.T, d.D, d.F));
    w = tx_max64(w, asd_tx_linear_workspace(d.T, 3 * d.D, d.D));
    w = tx_max64(w, asd_tx_linear_workspace(d.T, d.D, 3 * d.D));
    w = tx_max64(w, asd_tx_linear_workspace(d.Tc, 2 * d.D, d.Dc));
    w = tx_max64(w, asd_tx_wgrad_workspace(d.T, d.F, d.D));
    w = tx_max64(w, asd_tx_wgrad_workspace(d.T, d.D, d.F));
    w = tx_max64(w, asd_tx_wgrad_workspace(d.T, 3 * d.D, d.D));
    w = tx_max64(w, asd_tx_wgrad_workspace(d.Tc, 2 * d.D, d.Dc));
    w = tx_max64(w, asd_tx_wgrad_workspace(d.T, d.D, d.O));
    w = tx_max64(w, asd_tx_attention_workspace(d.T, d.T, d.H));
    w = tx_max64(w, asd_tx_attention_workspace(d.T, d.Tc, d.H));
    return tx_al(w);
}
inline int64_t tx_stage_floats(const TxDims& d) {
    int64_t w = (int64_t)d.F * d.D;
    w = tx_max64(w, (int64_t)3 * d.D * d.D);
    w = tx_max64(w, (int64_t)2 * d.D * d.Dc);
    w = tx_max64(w, (int64_t)d.D * d.O);
    return tx_al(w) + tx_al(tx_max64(d.F, 3 * d.D));
}
int tx_check_desc(const asd_tritx_desc* d) {
    ASD_CHECK_ARG(d && d->n_layers > 0 && d->heads > 0 && d->dim == d->heads * TX_HD, "tritx: dim must be heads * 48");
    ASD_CHECK_ARG(d->dim % 64 == 0 && d->dim <= 1024 && d->cond_dim % 64 == 0 && d->hidden % 64 == 0 && (4 * d->out_channels) % 64 == 0 && d->low_res > 0 &&
                  d->cond_tokens > 0, "tritx: dim / cond_dim / hidden / 4 * out_channels must be multiples of 64, dim <= 1024");
    return ASD_OK;
}
#define TXS(call) do { const int rc__ = (call); if (rc__ != ASD_OK) return rc__; } while (0)
}  // namespace

int64_t asd_tritx_packed_floats(const asd_tritx_desc* desc) {
    if (tx_check_desc(desc) != ASD_OK) return -1;
    const TxDims d(*desc);
    const int64_t per_layer = tx_pack_layout(d, 0).end;
    // + staging for the stacked q|k|v and k|v weights
    return tx_pack_head(d, per_layer * d.layers).end + tx_al((int64_t)3 * d.D * d.D) + tx_al((int64_t)2 * d.D * d.Dc) + tx_al(d.F + d.D + d.Dc + 64);
}
int64_t asd_tritx_save_floats(const asd_tritx_desc* desc, int32_t batch) {
    if (tx_check_desc(desc) != ASD_OK) return -1;
    return tx_save_per_sample(TxDims(*desc)) * batch;
}
static int64_t tx_pool_words(const TxDims& d) {
    // per layer: two attention calls (forward or backward) + seven weight gradients (column-max cells of both operands)
    const int64_t wg = tx_al(d.D) * 8 + tx_al(d.F) * 2 + tx_al(3 * d.D) + tx_al(2 * d.D) + tx_al(d.Dc) + 7 * 64;
    return (int64_t)d.layers * (2 * tx_al(8 * d.H) + wg) + tx_al(d.D) + tx_al(d.O) + 4096;
}

int64_t asd_tritx_workspace_floats(const asd_tritx_desc* desc) {
    if (tx_check_desc(desc) != ASD_OK) return -1;
    const TxDims d(*desc);
    // transient gradients of the backward pass: dx ping-pong (2), dn, do, dqkv (3), dkv, du, dy, the weight-gradient staging of batch
    // elements > 0, + the op workspace
    return tx_op_ws(d) + tx_al((int64_t)d.T * d.D) * 7 + tx_al((int64_t)d.Tc * 2 * d.D) + tx_al((int64_t)d.T * d.F) + tx_al((int64_t)d.T * d.O) +
           tx_stage_floats(d) + tx_pool_words(d) + tx_al((int64_t)d.Dc * 3 * tx_rp(d.Tc) / 2 + 64) + 2 * tx_al(d.Dc) + 256;
}

// params: 20 * n_layers + 4 device pointers (order: include/asd_hip.h) -> packed operand planes of every weight
int asd_tritx_pack(const asd_tritx_desc* desc, const float* const* params, float* packed, void* stream) {
    TXS(tx_check_desc(desc));
    ASD_CHECK_ARG(params && packed, "null argument");
    const TxDims d(*desc);
    hipStream_t s = (hipStream_t)stream;
    const int64_t per_layer = tx_pack_layout(d, 0).end;
    const TxPackHead hd = tx_pack_head(d, per_layer * d.layers);
    float* stage_qkv = packed + hd.end;
    float* stage_kv = stage_qkv + tx_al((int64_t)3 * d.D * d.D);
    float* cws = stage_kv + tx_al((int64_t)2 * d.D * d.Dc);
#define PK(w, N, K, WP, IW, TP, IT) TXS(asd_tx_pack_weight(w, N, K, packed + (WP), packed + (IW), (TP) >= 0 ? (void*)(packed + (TP)) : nullptr, (TP) >= 0 ? packed + (IT) : nullptr, cws, stream))
    for (int l = 0; l < d.layers; ++l) {
        const float* const* P = params + 20 * l;
        const TxPackLayer L = tx_pack_layout(d, per_layer * l);
        PK(P[2], d.D, d.D, L.caq_w, L.caq_iw, L.caq_t, L.caq_it);
        (void)hipMemcpyAsync(stage_kv, P[3], (size_t)d.D * d.Dc * 4, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(stage_kv + (size_t)d.D * d.Dc, P[4], (size_t)d.D * d.Dc * 4, hipMemcpyDeviceToDevice, s);
        PK(stage_kv, 2 * d.D, d.Dc, L.cakv_w, L.cakv_iw, (int64_t)-1, (int64_t)-1);
        PK(P[5], d.D, d.D, L.cao_w, L.cao_iw, L.cao_t, L.cao_it);
        for (int q = 0; q < 3; ++q) (void)hipMemcpyAsync(stage_qkv + (size_t)q * d.D * d.D, P[9 + q], (size_t)d.D * d.D * 4, hipMemcpyDeviceToDevice, s);
        PK(stage_qkv, 3 * d.D, d.D, L.qkv_w, L.qkv_iw, L.qkv_t, L.qkv_it);
        PK(P[12], d.D, d.D, L.sao_w, L.sao_iw, L.sao_t, L.sao_it);
        PK(P[16], d.F, d.D, L.fc1_w, L.fc1_iw, L.fc1_t, L.fc1_it);
        PK(P[18], d.D, d.F, L.fc2_w, L.fc2_iw, L.fc2_t, L.fc2_it);
        hipLaunchKernelGGL(tx_bounds_kernel, dim3(3 + asd_div_up(d.F, 4)), dim3(256), 0, s, P[0], P[1], P[7], P[8], P[14], P[15], P[16], P[17], d.D, d.F,
                           reinterpret_cast<unsigned*>(packed + L.bn1), reinterpret_cast<unsigned*>(packed + L.bn2), reinterpret_cast<unsigned*>(packed + L.bn3),
                           reinterpret_cast<unsigned*>(packed + L.bh));
    }
#undef PK
    // ConvTranspose2d(D -> C, 2, 2) weight V [D, 4 C] is the TRANSPOSE of the equivalent Linear's weight [4 C, D]: its plane for y = x W^T is
    // the column split of V, its plane for dx = dy W the row split of V
    const float* V = params[20 * d.layers + 3];
    unsigned* colmax = reinterpret_cast<unsigned*>(cws);
    tx_memset0(colmax, (size_t)d.O * 4, s);
    hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(d.O, 64), asd_div_up(d.D, 256)), dim3(256), 0, s, V, d.D, d.O, d.O, 256, colmax, (float*)nullptr);
    hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(d.O, 64), tx_rp(d.D) / 64), dim3(256), 0, s, V, d.D, d.O, d.O, tx_rp(d.D), colmax,
                       reinterpret_cast<h16*>(packed + hd.dc_w), packed + hd.dc_iw);
    hipLaunchKernelGGL((tx_split_rows_kernel<1>), dim3(asd_div_up(d.D, 4)), dim3(256), 0, s, V, d.D, d.O, d.O, reinterpret_cast<h16*>(packed + hd.dc_t), packed + hd.dc_it);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// text_embed [batch, Tc, Dc] -> planes_cl [batch, 3, 2R, 2R, C] (channel-last); `save` keeps what asd_tritx_bwd needs
int asd_tritx_fwd(const asd_tritx_desc* desc, const float* const* params, const float* packed, const float* text_embed, int32_t batch, float* planes_cl,
                  float* save, float* ws, void* stream) {
    TXS(tx_check_desc(desc));
    ASD_CHECK_ARG(params && packed && text_embed && planes_cl && save && ws && batch > 0, "null argument");
    const TxDims d(*desc);
    hipStream_t s = (hipStream_t)stream;
    const int64_t per_layer = tx_pack_layout(d, 0).end;
    const TxPackHead hd = tx_pack_head(d, per_layer * d.layers);
    const TxSave S = tx_save_layout(d);
    const int64_t TD = (int64_t)d.T * d.D;
    const float eps = desc->eps;
    struct PoolGuard { ~PoolGuard() { tx_pool = {nullptr, 0, false}; } } pool_guard;
    unsigned* pool = reinterpret_cast<unsigned*>(ws + (asd_tritx_workspace_floats(desc) - tx_pool_words(d) - 256));
    for (int n = 0; n < batch; ++n) {
        tx_memset0(pool, (size_t)tx_pool_words(d) * 4, s);
        tx_pool = {pool, (size_t)tx_pool_words(d), false};
        float* sv = save + (int64_t)n * tx_save_per_sample(d);
        const float* cond = text_embed + (int64_t)n * d.Tc * d.Dc;
        const float* x = params[20 * d.layers];        // pos_embed
        for (int l = 0; l < d.layers; ++l) {
            const float* const* P = params + 20 * l;
            const TxPackLayer L = tx_pack_layout(d, per_layer * l);
            float* B = sv + (int64_t)l * S.end;
            if (x != B + S.x_in)     // layer 0: the position embedding; later layers found their input written in place by the layer before
                hipLaunchKernelGGL(tx_add_kernel, dim3(asd_grid_for(TD / 4, 256)), dim3(256), 0, s, B + S.x_in, x, (size_t)(TD / 4), 0);
            x = B + S.x_in;
            // cross-attention
            // (every LayerNorm leaves the operand planes of its output where the Linear that consumes it looks for them)
            TXS(tx_layernorm_fwd_core(x, d.T, d.D, P[0], P[1], eps, B + S.n1, B + S.st1, tx_linear_planes_of(ws), tx_linear_inv_of(ws, d.T, d.D), stream));
            TXS(tx_linear_core(B + S.n1, d.T, d.D, d.D, packed + L.caq_w, packed + L.caq_iw, d.D, nullptr, 0, nullptr, nullptr, 0, B + S.q_ca, d.D, ws, true, stream));
            TXS(asd_tx_linear(cond, d.Tc, d.Dc, d.Dc, packed + L.cakv_w, packed + L.cakv_iw, 2 * d.D, nullptr, 0, nullptr, nullptr, 0, B + S.kv_ca, 2 * d.D, ws, stream));
            TXS(asd_tx_attention_fwd(B + S.q_ca, d.D, B + S.kv_ca, 2 * d.D, B + S.kv_ca + d.D, 2 * d.D, d.T, d.Tc, d.H, B + S.o_ca, d.D, B + S.lse_ca, ws, stream));
            TXS(asd_tx_linear(B + S.o_ca, d.T, d.D, d.D, packed + L.cao_w, packed + L.cao_iw, d.D, P[6], 0, nullptr, x, d.D, B + S.x1, d.D, ws, stream));
            // self-attention
            TXS(tx_layernorm_fwd_core(B + S.x1, d.T, d.D, P[7], P[8], eps, B + S.n2, B + S.st2, tx_linear_planes_of(ws), tx_linear_inv_of(ws, d.T, d.D), stream));
            TXS(tx_linear_core(B + S.n2, d.T, d.D, d.D, packed + L.qkv_w, packed + L.qkv_iw, 3 * d.D, nullptr, 0, nullptr, nullptr, 0, B + S.qkv, 3 * d.D, ws, true, stream));
            TXS(asd_tx_attention_fwd(B + S.qkv, 3 * d.D, B + S.qkv + d.D, 3 * d.D, B + S.qkv + 2 * d.D, 3 * d.D, d.T, d.T, d.H, B + S.o_sa, d.D, B + S.lse_sa, ws, stream));
            TXS(asd_tx_linear(B + S.o_sa, d.T, d.D, d.D, packed + L.sao_w, packed + L.sao_iw, d.D, P[13], 0, nullptr, B + S.x1, d.D, B + S.x2, d.D, ws, stream));
            // MLP
            TXS(tx_layernorm_fwd_core(B + S.x2, d.T, d.D, P[14], P[15], eps, B + S.n3, B + S.st3, tx_linear_planes_of(ws), tx_linear_inv_of(ws, d.T, d.D), stream));
            TXS(tx_linear_core(B + S.n3, d.T, d.D, d.D, packed + L.fc1_w, packed + L.fc1_iw, d.F, P[17], 1, B + S.u, nullptr, 0, B + S.hmid, d.F, ws, true, stream));
            float* xo = l + 1 < d.layers ? sv + (int64_t)(l + 1) * S.end + S.x_in : sv + (int64_t)d.layers * S.end;     // next layer's input slot / x_final
            TXS(asd_tx_linear(B + S.hmid, d.T, d.F, d.F, packed + L.fc2_w, packed + L.fc2_iw, d.D, P[19], 0, nullptr, B + S.x2, d.D, xo, d.D, ws, stream));
            x = xo;
        }
        float* xf = sv + (int64_t)d.layers * S.end;
        float* stF = xf + tx_al(TD);
        float* nF = stF + tx_al(2 * d.T);
        float* y = nF + tx_al(TD);
        TXS(asd_tx_layernorm_fwd(xf, d.T, d.D, params[20 * d.layers + 1], params[20 * d.layers + 2], eps, nF, stF, stream));
        TXS(asd_tx_linear(nF, d.T, d.D, d.D, packed + hd.dc_w, packed + hd.dc_iw, d.O, nullptr, 0, nullptr, nullptr, 0, y, d.O, ws, stream));
        hipLaunchKernelGGL(tx_shuffle_kernel, dim3(asd_grid_for((int64_t)d.T * d.O, 256)), dim3(256), 0, s, y, d.R, d.Cc, planes_cl + (int64_t)n * d.T * d.O, 0);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// grads: 17 * n_layers + 4 device pointers (order: include/asd_hip.h; q|k|v and k|v gradients stacked), WRITTEN (summed over the batch)
int asd_tritx_bwd(const asd_tritx_desc* desc, const float* const* params, const float* packed, const float* text_embed, int32_t batch,
                  const float* d_planes_cl, const float* save, float* const* grads, float* ws, void* stream) {
    TXS(tx_check_desc(desc));
    ASD_CHECK_ARG(params && packed && text_embed && d_planes_cl && save && grads && ws && batch > 0, "null argument");
    const TxDims d(*desc);
    hipStream_t s = (hipStream_t)stream;
    const int64_t per_layer = tx_pack_layout(d, 0).end;
    const TxPackHead hd = tx_pack_head(d, per_layer * d.layers);
    const TxSave S = tx_save_layout(d);
    const int64_t TD = (int64_t)d.T * d.D;
    float* p = ws + tx_op_ws(d);
    float* dxa = p; p += tx_al(TD);
    float* dxb = p; p += tx_al(TD);
    float* dn = p; p += tx_al(TD);
    float* dob = p; p += tx_al(TD);
    float* dqkv = p; p += 3 * tx_al(TD);
    float* dkv = p; p += tx_al((int64_t)d.Tc * 2 * d.D);
    float* du = p; p += tx_al((int64_t)d.T * d.F);
    float* dy = p; p += tx_al((int64_t)d.T * d.O);
    h16* cond_planes = reinterpret_cast<h16*>(p); p += tx_al((int64_t)d.Dc * 3 * tx_rp(d.Tc) / 2 + 64);      // text tokens^T [Dc, 3 Tcp]: one split for all layers
    float* cond_inv = p; p += tx_al(d.Dc);
    unsigned* cond_max = reinterpret_cast<unsigned*>(p); p += tx_al(d.Dc);
    (void)p;      // (a staging area follows in the workspace layout: unused since the weight gradients accumulate in their epilogue)
    // LayerNorm gradients accumulate over layers' rows and the batch, bias gradients are atomic column sums: zero them once — or take the
    // caller's word that it did (desc->grads_prezeroed: the Python side carves all of them out of one zeroed buffer)
    float* const* GH = grads + 17 * d.layers;       // pos_embed, norm.w, norm.b, deconv.w
    if (!desc->grads_prezeroed) {
        for (int l = 0; l < d.layers; ++l) {
            float* const* G = grads + 17 * l;
            for (int q : {0, 1, 6, 7, 11, 12}) tx_memset0(G[q], (size_t)d.D * 4, s);
        }
        tx_memset0(GH[1], (size_t)d.D * 4, s);
        tx_memset0(GH[2], (size_t)d.D * 4, s);
    }
    struct PoolGuard { ~PoolGuard() { tx_pool = {nullptr, 0, false}; } } pool_guard;
    unsigned* pool = reinterpret_cast<unsigned*>(ws + (asd_tritx_workspace_floats(desc) - tx_pool_words(d) - 256));
    // weight gradients of batch element n > 0 are added to those of the elements before it — in the product's own epilogue (residual = dw)
    const h16* xpl = nullptr;          // precomputed transposed planes of x (+ scales) / column bounds of x for the NEXT wgrad call
    const float* xinv = nullptr;
    const unsigned* xbound = nullptr;
    auto wgrad = [&](const float* dyp, int ldy, const float* xp, int ldx, int M, int N, int K, float* dw, float* db, bool acc) -> int {
        const h16* pl = xpl; const float* pi = xinv; const unsigned* pb = xbound;
        xpl = nullptr; xinv = nullptr; xbound = nullptr;
        return tx_wgrad_core(dyp, ldy, xp, ldx, M, N, K, dw, db, ws, pl, pi, pb, stream, acc);
    };
    for (int n = 0; n < batch; ++n) {
        const bool acc = n > 0;
        tx_memset0(pool, (size_t)tx_pool_words(d) * 4, s);
        // bias gradients of the first batch element land in caller-zeroed cells; later elements go through the staging buffer (own memset)
        tx_pool = {pool, (size_t)tx_pool_words(d), desc->grads_prezeroed != 0 && !acc};
        const float* sv = save + (int64_t)n * tx_save_per_sample(d);
        const float* cond = text_embed + (int64_t)n * d.Tc * d.Dc;
        const float* xf = sv + (int64_t)d.layers * S.end;
        const float* stF = xf + tx_al(TD);
        const float* nF = stF + tx_al(2 * d.T);
        {   // the text tokens' transposed planes, once per batch element
            tx_memset0(cond_max, (size_t)d.Dc * 4, s);
            hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(d.Dc, 64), asd_div_up(d.Tc, 256)), dim3(256), 0, s, cond, d.Tc, d.Dc, d.Dc, 256, cond_max, (float*)nullptr);
            hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(d.Dc, 64), tx_rp(d.Tc) / 64), dim3(256), 0, s, cond, d.Tc, d.Dc, d.Dc, tx_rp(d.Tc), cond_max, cond_planes,
                               cond_inv);
        }
        // head: planes -> y -> nF -> x_final
        hipLaunchKernelGGL(tx_shuffle_kernel, dim3(asd_grid_for((int64_t)d.T * d.O, 256)), dim3(256), 0, s, dy, d.R, d.Cc,
                           const_cast<float*>(d_planes_cl) + (int64_t)n * d.T * d.O, 1);
        // dV [D, 4C] = nF^T dy: the "Linear" with the roles of x and dy swapped
        TXS(wgrad(nF, d.D, dy, d.O, d.T, d.D, d.O, GH[3], nullptr, acc));
        TXS(asd_tx_linear(dy, d.T, d.O, d.O, packed + hd.dc_t, packed + hd.dc_it, d.D, nullptr, 0, nullptr, nullptr, 0, dn, d.D, ws, stream));
        float* dx = dxa;
        float* dx_other = dxb;
        // (every LayerNorm backward leaves the operand planes of its dx where the next input-gradient Linear looks for them: that Linear runs
        // BEFORE the weight gradient that shares dx with it, whose workspace would overwrite the planes)
        h16* const lp = tx_linear_planes_of(ws);
        float* const li = tx_linear_inv_of(ws, d.T, d.D);
        TXS(tx_layernorm_bwd_core(dn, xf, stF, params[20 * d.layers + 1], d.T, d.D, nullptr, dx, GH[1], GH[2], lp, li, stream));
        for (int l = d.layers - 1; l >= 0; --l) {
            const float* const* P = params + 20 * l;
            float* const* G = grads + 17 * l;
            const TxPackLayer L = tx_pack_layout(d, per_layer * l);
            const float* B = sv + (int64_t)l * S.end;
            // ---- MLP: x3 = x2 + fc2(gelu(fc1(n3)))
            TXS(tx_linear_core(dx, d.T, d.D, d.D, packed + L.fc2_t, packed + L.fc2_it, d.F, nullptr, 2, const_cast<float*>(B + S.u), nullptr, 0, du, d.F, ws, true, stream));
            xbound = reinterpret_cast<const unsigned*>(packed + L.bh);
            TXS(wgrad(dx, d.D, B + S.hmid, d.F, d.T, d.D, d.F, G[15], G[16], acc));
            TXS(asd_tx_linear(du, d.T, d.F, d.F, packed + L.fc1_t, packed + L.fc1_it, d.D, nullptr, 0, nullptr, nullptr, 0, dn, d.D, ws, stream));
            xbound = reinterpret_cast<const unsigned*>(packed + L.bn3);
            TXS(wgrad(du, d.F, B + S.n3, d.D, d.T, d.F, d.D, G[13], G[14], acc));
            TXS(tx_layernorm_bwd_core(dn, B + S.x2, B + S.st3, P[14], d.T, d.D, dx, dx_other, G[11], G[12], lp, li, stream));
            { float* t = dx; dx = dx_other; dx_other = t; }
            // ---- self-attention: x2 = x1 + o_sa Wo^T + bo
            TXS(tx_linear_core(dx, d.T, d.D, d.D, packed + L.sao_t, packed + L.sao_it, d.D, nullptr, 0, nullptr, nullptr, 0, dob, d.D, ws, true, stream));
            TXS(wgrad(dx, d.D, B + S.o_sa, d.D, d.T, d.D, d.D, G[9], G[10], acc));
            TXS(asd_tx_attention_bwd(B + S.qkv, 3 * d.D, B + S.qkv + d.D, 3 * d.D, B + S.qkv + 2 * d.D, 3 * d.D, B + S.o_sa, d.D, dob, d.D, B + S.lse_sa, d.T, d.T, d.H,
                                     dqkv, 3 * d.D, dqkv + d.D, 3 * d.D, dqkv + 2 * d.D, 3 * d.D, ws, stream));
            xbound = reinterpret_cast<const unsigned*>(packed + L.bn2);
            TXS(wgrad(dqkv, 3 * d.D, B + S.n2, d.D, d.T, 3 * d.D, d.D, G[8], nullptr, acc));
            TXS(asd_tx_linear(dqkv, d.T, 3 * d.D, 3 * d.D, packed + L.qkv_t, packed + L.qkv_it, d.D, nullptr, 0, nullptr, nullptr, 0, dn, d.D, ws, stream));
            TXS(tx_layernorm_bwd_core(dn, B + S.x1, B + S.st2, P[7], d.T, d.D, dx, dx_other, G[6], G[7], lp, li, stream));
            { float* t = dx; dx = dx_other; dx_other = t; }
            // ---- cross-attention: x1 = x0 + o_ca Wo^T + bo
            TXS(tx_linear_core(dx, d.T, d.D, d.D, packed + L.cao_t, packed + L.cao_it, d.D, nullptr, 0, nullptr, nullptr, 0, dob, d.D, ws, true, stream));
            TXS(wgrad(dx, d.D, B + S.o_ca, d.D, d.T, d.D, d.D, G[4], G[5], acc));
            TXS(asd_tx_attention_bwd(B + S.q_ca, d.D, B + S.kv_ca, 2 * d.D, B + S.kv_ca + d.D, 2 * d.D, B + S.o_ca, d.D, dob, d.D, B + S.lse_ca, d.T, d.Tc, d.H,
                                     dqkv, d.D, dkv, 2 * d.D, dkv + d.D, 2 * d.D, ws, stream));
            xbound = reinterpret_cast<const unsigned*>(packed + L.bn1);
            TXS(wgrad(dqkv, d.D, B + S.n1, d.D, d.T, d.D, d.D, G[2], nullptr, acc));
            xpl = cond_planes; xinv = cond_inv;
            TXS(wgrad(dkv, 2 * d.D, cond, d.Dc, d.Tc, 2 * d.D, d.Dc, G[3], nullptr, acc));
            TXS(asd_tx_linear(dqkv, d.T, d.D, d.D, packed + L.caq_t, packed + L.caq_it, d.D, nullptr, 0, nullptr, nullptr, 0, dn, d.D, ws, stream));
            TXS(tx_layernorm_bwd_core(dn, B + S.x_in, B + S.st1, P[0], d.T, d.D, dx, dx_other, G[0], G[1], lp, li, stream));
            { float* t = dx; dx = dx_other; dx_other = t; }
        }
        hipLaunchKernelGGL(tx_add_kernel, dim3(asd_grid_for(TD / 4, 256)), dim3(256), 0, s, GH[0], dx, (size_t)(TD / 4), acc ? 1 : 0);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
