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

// column max |x| (as uint bits: monotone for non-negative floats) and column sums of X [R, C]: blocks of 64 columns x `rows_per_block` rows
__global__ __launch_bounds__(256) void tx_colstat_kernel(const float* __restrict__ x, int R, int C, int ld, int rows_per_block,
                                                         unsigned* __restrict__ colmax, float* __restrict__ colsum) {
    __shared__ float smax[4][64], ssum[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float m = 0.f, s = 0.f;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 4) {
            const float v = x[(size_t)r * ld + c];
            m = fmaxf(m, fabsf(v));
            s += v;
        }
    smax[rl][threadIdx.x & 63] = m; ssum[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        m = fmaxf(fmaxf(smax[0][threadIdx.x], smax[1][threadIdx.x]), fmaxf(smax[2][threadIdx.x], smax[3][threadIdx.x]));
        s = (ssum[0][threadIdx.x] + ssum[1][threadIdx.x]) + (ssum[2][threadIdx.x] + ssum[3][threadIdx.x]);
        if (colmax) atomicMax(colmax + c, __float_as_uint(m));
        if (colsum) atomicAdd(colsum + c, s);
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
__global__ __launch_bounds__(256) void tx_epilogue_kernel(const float* __restrict__ acc, int M, int N, const float* __restrict__ ia,
                                                          const float* __restrict__ iw, const float* __restrict__ bias, int mode,
                                                          float* __restrict__ aux, int ld_aux, const float* __restrict__ residual, int ldr,
                                                          float* __restrict__ y, int ldy) {
    const int n4 = N / 4;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < (size_t)M * n4; t += (size_t)gridDim.x * 256) {
        const int i = (int)(t / n4), j = (int)(t % n4) * 4;
        float4 v = *reinterpret_cast<const float4*>(acc + (size_t)i * N + j);
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

// ---- LayerNorm (fp32, one wave per row, D % 4 == 0, D <= 1024) -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tx_layernorm_fwd_kernel(const float* __restrict__ x, int M, int D, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                               float* __restrict__ stats /*[M,2] mean, rstd*/) {
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<float4*>(y + (size_t)r * D + c) = o;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g xhat)), g = dy * gamma, (+ residual gradient `dres`); dgamma += sum dy xhat, dbeta += sum dy.
// A wave walks rows r = wave, wave + n_waves, ... and keeps the dgamma / dbeta terms of ITS columns in registers: one atomic per column
// and wave at the end.
__global__ __launch_bounds__(256) void tx_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, int M, int D, const float* __restrict__ dres,
                                                               float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
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
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < D) {
            atomicAdd(dgamma + c, ag[i].x); atomicAdd(dgamma + c + 1, ag[i].y); atomicAdd(dgamma + c + 2, ag[i].z); atomicAdd(dgamma + c + 3, ag[i].w);
            atomicAdd(dbeta + c, ab[i].x); atomicAdd(dbeta + c + 1, ab[i].y); atomicAdd(dbeta + c + 2, ab[i].z); atomicAdd(dbeta + c + 3, ab[i].w);
        }
    }
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

inline int64_t tx_al(int64_t floats) { return (floats + 63) & ~(int64_t)63; }
inline int tx_rp(int r) { return (r + 63) & ~63; }      // rows of a transposed operand padded to the GEMM's k-step

// C32 [M, N] = planeA [M, 3K] . planeW [N, 3K]^T  (fp32 result of the three fp16 products)
int tx_gemm(const h16* pa, const h16* pw, int M, int N, int K3, float* c32, hipStream_t s) {
    asd_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = pa; a.W = pw; a.C = c32;
    a.M = M; a.N = N; a.K = K3;
    a.lda = K3; a.ldw = K3; a.ldc = N;
    a.out_f32 = 1;
    a.split_k = 1;
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
        unsigned* colmax = reinterpret_cast<unsigned*>(ws);
        (void)hipMemsetAsync(colmax, 0, (size_t)K * 4, s);
        hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(K, 64), asd_div_up(N, 256)), dim3(256), 0, s, w, N, K, K, 256, colmax, (float*)nullptr);
        hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(K, 64), tx_rp(N) / 64), dim3(256), 0, s, w, N, K, K, tx_rp(N), colmax, (h16*)plane_wt, inv_wt);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int64_t asd_tx_linear_workspace(int32_t M, int32_t N, int32_t K) {
    // A plane [M, 3K] halfs + row scales + fp32 product [M, N]
    return tx_al((int64_t)M * 3 * K / 2 + 64) + tx_al(M) + tx_al((int64_t)M * N);
}

// y [M, N] (ldy) = f(x [M, K] (ldx) . W^T + bias) + residual, W given as packed plane [N, 3K] + inv_w [N] (asd_tx_pack_weight; pass the
// W^T plane for an input gradient).  mode 0 identity, 1 GELU (pre-activation saved to aux), 2 multiply by GELU'(aux) (aux [M, N], ld N)
int asd_tx_linear(const float* x, int32_t M, int32_t K, int32_t ldx, const void* plane_w, const float* inv_w, int32_t N, const float* bias, int32_t mode,
                  float* aux, const float* residual, int32_t ldr, float* y, int32_t ldy, float* ws, void* stream) {
    ASD_CHECK_ARG(x && plane_w && inv_w && y && ws && M > 0 && N > 0 && K > 0, "null argument");
    ASD_CHECK_ARG(K % 64 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (mode == 0 || aux), "K % 64, N % 4, leading dimensions % 4; aux for GELU modes");
    hipStream_t s = (hipStream_t)stream;
    h16* pa = reinterpret_cast<h16*>(ws);
    float* ia = ws + tx_al((int64_t)M * 3 * K / 2 + 64);
    float* c32 = ia + tx_al(M);
    hipLaunchKernelGGL((tx_split_rows_kernel<0>), dim3(asd_div_up(M, 4)), dim3(256), 0, s, x, M, K, ldx, pa, ia);
    const int rc = tx_gemm(pa, (const h16*)plane_w, M, N, 3 * K, c32, s);
    if (rc != ASD_OK) return rc;
    hipLaunchKernelGGL(tx_epilogue_kernel, dim3(asd_grid_for((int64_t)M * N / 4, 256)), dim3(256), 0, s, c32, M, N, ia, inv_w, bias, mode, aux, N, residual, ldr, y, ldy);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int64_t asd_tx_wgrad_workspace(int32_t M, int32_t N, int32_t K) {
    const int64_t Mp = tx_rp(M);
    return tx_al((int64_t)N * 3 * Mp / 2 + 64) + tx_al((int64_t)K * 3 * Mp / 2 + 64) + tx_al(N) + tx_al(K) + tx_al(N) + tx_al(K) + tx_al((int64_t)N * K);
}

// dw [N, K] = dy [M, N]^T . x [M, K]  (contraction over the M rows), db [N] = column sums of dy (optional)
int asd_tx_linear_wgrad(const float* dy, int32_t ldy, const float* x, int32_t ldx, int32_t M, int32_t N, int32_t K, float* dw, float* db, float* ws, void* stream) {
    ASD_CHECK_ARG(dy && x && dw && ws && M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int Mp = tx_rp(M);
    h16* pa = reinterpret_cast<h16*>(ws);                                   // dy^T [N, 3 Mp]
    float* p = ws + tx_al((int64_t)N * 3 * Mp / 2 + 64);
    h16* pw = reinterpret_cast<h16*>(p);                                    // x^T [K, 3 Mp]
    p += tx_al((int64_t)K * 3 * Mp / 2 + 64);
    float* ia = p; p += tx_al(N);
    float* iw = p; p += tx_al(K);
    unsigned* cmax_a = reinterpret_cast<unsigned*>(p); p += tx_al(N);
    unsigned* cmax_w = reinterpret_cast<unsigned*>(p); p += tx_al(K);
    float* c32 = p;
    (void)hipMemsetAsync(cmax_a, 0, (size_t)(tx_al(N) + tx_al(K)) * 4, s);
    if (db) (void)hipMemsetAsync(db, 0, (size_t)N * 4, s);
    hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(N, 64), asd_div_up(M, 256)), dim3(256), 0, s, dy, M, N, ldy, 256, cmax_a, db);
    hipLaunchKernelGGL(tx_colstat_kernel, dim3(asd_div_up(K, 64), asd_div_up(M, 256)), dim3(256), 0, s, x, M, K, ldx, 256, cmax_w, (float*)nullptr);
    hipLaunchKernelGGL((tx_split_cols_kernel<0>), dim3(asd_div_up(N, 64), Mp / 64), dim3(256), 0, s, dy, M, N, ldy, Mp, cmax_a, pa, ia);
    hipLaunchKernelGGL((tx_split_cols_kernel<1>), dim3(asd_div_up(K, 64), Mp / 64), dim3(256), 0, s, x, M, K, ldx, Mp, cmax_w, pw, iw);
    const int rc = tx_gemm(pa, pw, N, K, 3 * Mp, c32, s);
    if (rc != ASD_OK) return rc;
    hipLaunchKernelGGL(tx_epilogue_kernel, dim3(asd_grid_for((int64_t)N * K / 4, 256)), dim3(256), 0, s, c32, N, K, ia, iw, (const float*)nullptr, 0, (float*)nullptr, 0,
                       (const float*)nullptr, 0, dw, K);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_tx_layernorm_fwd(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y, float* stats, void* stream) {
    ASD_CHECK_ARG(x && gamma && beta && y && stats && M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "LayerNorm: D % 4 == 0, D <= 1024");
    hipLaunchKernelGGL(tx_layernorm_fwd_kernel, dim3(asd_div_up(M, 4)), dim3(256), 0, (hipStream_t)stream, x, M, D, gamma, beta, eps, y, stats);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// dx = LayerNorm input gradient (+ dres); dgamma / dbeta are ACCUMULATED (+=: the caller zeroes them once per backward pass)
int asd_tx_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, int32_t M, int32_t D, const float* dres, float* dx,
                         float* dgamma, float* dbeta, void* stream) {
    ASD_CHECK_ARG(dy && x && stats && gamma && dx && dgamma && dbeta && M > 0 && D > 0 && D % 4 == 0 && D <= 1024, "LayerNorm: D % 4 == 0, D <= 1024");
    int grid = asd_div_up(M, 4 * 6);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(tx_layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, x, stats, gamma, M, D, dres, dx, dgamma, dbeta);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
