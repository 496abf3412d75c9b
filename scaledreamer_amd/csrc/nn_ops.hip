// nn_ops.hip — bandwidth-bound layers of the SD-2.1 UNet on NHWC fp16 tensors (gfx950): GroupNorm(+SiLU),
// LayerNorm, GEGLU, SiLU, sinusoidal timestep embedding, channel concat.  All kernels move 16 B per lane
// (8 halfs), keep statistics in fp32 (GroupNorm32, diffusionmodules/util.py:229-231) and are HBM-bound:
// algorithmic bytes = 2 B read + 2 B written per element (GroupNorm reads x twice: stats + apply).
#include "asd_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// ---- GroupNorm statistics: grid (batch, chunks); block 256.  A thread owns a fixed 8-channel slot (two when
// C > 2048) and walks the rows of its chunk with stride 256/slots, accumulating per-channel sums in registers;
// only at the end are they folded into 32 per-group LDS cells and from there into global memory (one atomic
// pair per block and group).
#ifndef GN_UNR
#define GN_UNR 4      // rows in flight per thread (16-B loads)
#endif
#ifndef GN_CAP_A
#define GN_CAP_A 512  // apply blocks per launch
#endif
#ifndef GN_FOLD_RECORDS
#define GN_FOLD_RECORDS 96   // up to this many epilogue records per batch element are summed by every apply block itself (24 KB of L2
                             // reads, four loads in flight per thread); above it a reduce launch (~5 us) runs first
#endif

__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* __restrict__ x1, int c1, const half_t* __restrict__ x2,
                                                       int c2, int hw, int rows_per_block, float* __restrict__ stats) {
    __shared__ float gsum[32], gsq[32];
    const int C = c1 + c2, slots = C / 8, cg = C / 32;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 32) { gsum[tid] = 0.f; gsq[tid] = 0.f; }
    __syncthreads();
    const int r0 = blockIdx.y * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const int rows_in_flight = slots >= 256 ? 1 : 256 / slots;
    const int rsub = slots >= 256 ? 0 : tid / slots;
    const int slot0 = slots >= 256 ? tid : tid % slots;
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // second pass only when more than 256 slots (C = 2560)
        const int slot = slot0 + j * 256;
        if (slot >= slots || rsub >= rows_in_flight || (j == 1 && slots <= 256)) continue;
        float s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
        const int c = slot * 8;
        // 4 rows per trip: four independent 16-byte loads in flight per thread (a single dependent load per trip left the
        // kernel latency-bound at a third of the HBM rate on the 67 MB VAE tensors)
        const half_t* base = c < c1 ? x1 + (size_t)b * hw * c1 + c : x2 + (size_t)b * hw * c2 + (c - c1);
        const size_t rstride = c < c1 ? c1 : c2;
        int r = r0 + rsub;
        for (; r + (GN_UNR - 1) * rows_in_flight < r1; r += GN_UNR * rows_in_flight) {
            half8 v[GN_UNR];
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) v[u] = *(const half8*)(base + (size_t)(r + u * rows_in_flight) * rstride);
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = (float)v[u][k]; s[k] += f; q[k] = fmaf(f, f, q[k]); }
        }
        for (; r < r1; r += rows_in_flight) {
            const half8 v = *(const half8*)(base + (size_t)r * rstride);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; s[k] += f; q[k] = fmaf(f, f, q[k]); }
        }
        int g = c / cg;
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gk = (c + k) / cg;
            if (gk != g) { atomicAdd(&gsum[g], ss); atomicAdd(&gsq[g], qq); g = gk; ss = 0.f; qq = 0.f; }
            ss += s[k]; qq += q[k];
        }
        atomicAdd(&gsum[g], ss);
        atomicAdd(&gsq[g], qq);
    }
    __syncthreads();
    if (tid < 32) {   // per-block partial sums (no global atomics, nothing to zero): reduced by the apply kernel's prologue
        float* part = stats + ((size_t)b * gridDim.y + blockIdx.y) * 64;
        part[tid * 2] = gsum[tid];
        part[tid * 2 + 1] = gsq[tid];
    }
}

// sum of the statistics partials of batch element b -> st[64] (LDS); 256 threads.  Four independent loads in flight per thread:
// the partials are the head of the kernel's dependency chain (partials -> scale/shift -> first store), and these launches are short
// enough (5-10 us) that every serialized L2/HBM round trip shows.
__device__ __forceinline__ void gn_reduce_partials(const float* __restrict__ partials, int b, int n_chunks, float* st /*[64]*/,
                                                   float (*scratch)[64] /*[4][64]*/) {
    const int tid = threadIdx.x, v = tid & 63, q = tid >> 6;
    const float* src = partials + (size_t)b * n_chunks * 64 + v;
    // eight loads per trip, all requested before the first is consumed (index clamped, contribution masked): up to 96 folded
    // records per batch element are three round trips, not twenty-four
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int c = q; c < n_chunks; c += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)min(c + 4 * u, n_chunks - 1) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += c + 4 * u < n_chunks ? t[u] : 0.f;
    }
    scratch[q][v] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (tid < 64) st[tid] = (scratch[0][tid] + scratch[1][tid]) + (scratch[2][tid] + scratch[3][tid]);
    __syncthreads();
}

// Apply: same thread -> channel-slot ownership as the statistics kernel (grid (batch, chunks)): the per-channel scale and
// shift  y = x * a + b  (a = rstd * gamma, b = beta - mean * a) are formed ONCE per thread, the row loop is one 16-byte
// load, 8 FMAs (+ SiLU) and one 16-byte store — the first version recomputed group index, mean and rsqrt per element and
// was VALU-bound at a quarter of the HBM rate.
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ x1, int c1, const half_t* __restrict__ x2,
                                                       int c2, int hw, int rows_per_block, const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, float eps, int silu,
                                                       const float* __restrict__ partials, int n_chunks,
                                                       float* __restrict__ stats_out, half_t* __restrict__ y) {
    __shared__ float st[64];
    __shared__ float scratch[4][64];
    const int C = c1 + c2, slots = C / 8, cg = C / 32;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv_cnt = 1.f / ((float)hw * (float)cg);
    const int r0 = blockIdx.y * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const int rows_in_flight = slots >= 256 ? 1 : 256 / slots;
    const int rsub = slots >= 256 ? 0 : tid / slots;
    const int slot0 = slots >= 256 ? tid : tid % slots;
    // The first trip's rows and the affine parameters do not depend on the statistics: request them BEFORE the partials are
    // reduced (the compiler cannot move loads across the barrier inside gn_reduce_partials), so the launch pays one memory
    // round trip before its first store instead of three in a row — most of these launches are 5-10 us long.
    const bool act0 = slot0 < slots && rsub < rows_in_flight;
    const int cA = slot0 * 8;
    const half_t* baseA = cA < c1 ? x1 + (size_t)b * hw * c1 + cA : x2 + (size_t)b * hw * c2 + (cA - c1);
    const size_t rstrideA = cA < c1 ? c1 : c2;
    half8 pre[GN_UNR], gpre = {}, bpre = {};
    const int rA = r0 + rsub;
    const bool pre_ok = act0 && rA + (GN_UNR - 1) * rows_in_flight < r1;
    if (act0) { gpre = *(const half8*)(gamma + cA); bpre = *(const half8*)(beta + cA); }
    if (pre_ok) {
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) pre[u] = *(const half8*)(baseA + (size_t)(rA + u * rows_in_flight) * rstrideA);
    }
    gn_reduce_partials(partials, b, n_chunks, st, scratch);
    if (blockIdx.y == 0 && tid < 64) stats_out[b * 64 + tid] = st[tid];   // kept for the backward pass
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // second pass only when more than 256 slots (C = 2560)
        const int slot = slot0 + j * 256;
        if (slot >= slots || rsub >= rows_in_flight || (j == 1 && slots <= 256)) continue;
        const int c = slot * 8;
        half8 gv = gpre, bv = bpre;
        if (j == 1) { gv = *(const half8*)(gamma + c); bv = *(const half8*)(beta + c); }
        float sa[8], sb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (c + k) / cg;
            const float mean = st[g * 2] * inv_cnt;
            const float var = fmaxf(st[g * 2 + 1] * inv_cnt - mean * mean, 0.f);
            sa[k] = rsqrtf(var + eps) * (float)gv[k];
            sb[k] = (float)bv[k] - mean * sa[k];
        }
        const half_t* base = c < c1 ? x1 + (size_t)b * hw * c1 + c : x2 + (size_t)b * hw * c2 + (c - c1);
        const size_t rstride = c < c1 ? c1 : c2;
        half_t* ybase = y + (size_t)b * hw * C + c;
        auto emit = [&](const half8& v, int r) __attribute__((always_inline)) {
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float f = fmaf((float)v[k], sa[k], sb[k]);
                if (silu) f = silu_f(f);
                o[k] = (half_t)f;
            }
            *(half8*)(ybase + (size_t)r * C) = o;
        };
        int r = r0 + rsub;
        if (j == 0 && pre_ok) {
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) emit(pre[u], r + u * rows_in_flight);
            r += GN_UNR * rows_in_flight;
        }
        for (; r + (GN_UNR - 1) * rows_in_flight < r1; r += GN_UNR * rows_in_flight) {
            half8 v[GN_UNR];
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) v[u] = *(const half8*)(base + (size_t)(r + u * rows_in_flight) * rstride);
#pragma unroll
            for (int u = 0; u < GN_UNR; ++u) emit(v[u], r + u * rows_in_flight);
        }
        for (; r < r1; r += rows_in_flight) emit(*(const half8*)(base + (size_t)r * rstride), r);
    }
}

// ---- GroupNorm(+SiLU) backward w.r.t. the input (frozen gamma/beta: VAE encoder input gradient) -------------
// y = silu?(z), z = xh * gamma + beta, xh = (x - mean) * rstd.  With g = dy * silu'(z) * gamma:
//   dx = rstd * (g - mean_grp(g) - xh * mean_grp(g * xh))
// pass 1 accumulates sum(g) and sum(g*xh) per (batch, group) exactly like the forward statistics kernel,
// pass 2 applies the formula.  x is re-read instead of saving xh/z (bandwidth is cheaper than HBM capacity
// is scarce here? no: 288 GB — but x must be kept for the convolution-free recompute anyway).
__device__ __forceinline__ float silu_grad(float z) {
    const float sg = 1.f / (1.f + __expf(-z));
    return sg * (1.f + z * (1.f - sg));
}

// NT threads per block: 256, or 512 for the large tensors — the grid is capped at 256 blocks (every apply block sums all their
// partials), and with four waves per CU the silu' arithmetic of a wave was not hidden behind anybody's loads: 2.6 TB/s on the VAE's
// 512^2 x 128 tensors.  Eight waves per CU: 51 -> 34.5 us there, 28 -> 20 us at 256^2 x 256 (sixteen: 36.5 / 23 us, 128 registers, spills)
template <int NT>
__global__ __launch_bounds__(NT) void gn_bwd_stats_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy, int C,
                                                           int hw, int rows_per_block, const half_t* __restrict__ gamma,
                                                           const half_t* __restrict__ beta, float eps, int silu,
                                                           const float* __restrict__ fstats, float* __restrict__ bstats) {
    __shared__ float gsum[32], gsq[32];
    const int slots = C / 8, cg = C / 32;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 32) { gsum[tid] = 0.f; gsq[tid] = 0.f; }
    __syncthreads();
    const float inv_cnt = 1.f / ((float)hw * (float)cg);
    const int r0 = blockIdx.y * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const int rows_in_flight = slots >= NT ? 1 : NT / slots;
    const int rsub = slots >= NT ? 0 : tid / slots;
    const int slot0 = slots >= NT ? tid : tid % slots;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = slot0 + j * NT;
        if (slot >= slots || rsub >= rows_in_flight || (j == 1 && slots <= NT)) continue;
        const int c = slot * 8;
        float mean[8], rstd[8], gm[8], bt[8], s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (c + k) / cg;
            mean[k] = fstats[(b * 32 + g) * 2] * inv_cnt;
            rstd[k] = rsqrtf(fmaxf(fstats[(b * 32 + g) * 2 + 1] * inv_cnt - mean[k] * mean[k], 0.f) + eps);
            gm[k] = (float)gamma[c + k]; bt[k] = (float)beta[c + k];
            s[k] = 0.f; q[k] = 0.f;
        }
        int r = r0 + rsub;
        for (; r + rows_in_flight < r1; r += 2 * rows_in_flight) {   // two rows (4 loads) in flight per thread
            const size_t off0 = ((size_t)b * hw + r) * C + c, off1 = off0 + (size_t)rows_in_flight * C;
            const half8 xv0 = *(const half8*)(x + off0), dv0 = *(const half8*)(dy + off0);
            const half8 xv1 = *(const half8*)(x + off1), dv1 = *(const half8*)(dy + off1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh0 = ((float)xv0[k] - mean[k]) * rstd[k], xh1 = ((float)xv1[k] - mean[k]) * rstd[k];
                float g0 = (float)dv0[k] * gm[k], g1 = (float)dv1[k] * gm[k];
                if (silu) { g0 *= silu_grad(fmaf(xh0, gm[k], bt[k])); g1 *= silu_grad(fmaf(xh1, gm[k], bt[k])); }
                s[k] += g0 + g1;
                q[k] = fmaf(g0, xh0, fmaf(g1, xh1, q[k]));
            }
        }
        for (; r < r1; r += rows_in_flight) {
            const size_t off = ((size_t)b * hw + r) * C + c;
            const half8 xv = *(const half8*)(x + off), dv = *(const half8*)(dy + off);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = ((float)xv[k] - mean[k]) * rstd[k];
                float g = (float)dv[k] * gm[k];
                if (silu) g *= silu_grad(fmaf(xh, gm[k], bt[k]));
                s[k] += g;
                q[k] = fmaf(g, xh, q[k]);
            }
        }
        int g = c / cg;
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gk = (c + k) / cg;
            if (gk != g) { atomicAdd(&gsum[g], ss); atomicAdd(&gsq[g], qq); g = gk; ss = 0.f; qq = 0.f; }
            ss += s[k]; qq += q[k];
        }
        atomicAdd(&gsum[g], ss);
        atomicAdd(&gsq[g], qq);
    }
    __syncthreads();
    if (tid < 32) {
        float* part = bstats + ((size_t)b * gridDim.y + blockIdx.y) * 64;
        part[tid * 2] = gsum[tid];
        part[tid * 2 + 1] = gsq[tid];
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy, int C,
                                                           int hw, int rows_per_block, const half_t* __restrict__ gamma,
                                                           const half_t* __restrict__ beta, float eps, int silu,
                                                           const float* __restrict__ fstats, const float* __restrict__ bpartials,
                                                           int n_chunks, const half_t* __restrict__ dx_add,
                                                           half_t* __restrict__ dx) {
    __shared__ float st[64];
    __shared__ float scratch[4][64];
    const int slots = C / 8, cg = C / 32;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv_cnt = 1.f / ((float)hw * (float)cg);
    const int r0 = blockIdx.y * rows_per_block, r1 = min(hw, r0 + rows_per_block);
    const int rows_in_flight = slots >= 256 ? 1 : 256 / slots;
    const int rsub = slots >= 256 ? 0 : tid / slots;
    const int slot0 = slots >= 256 ? tid : tid % slots;
    // first two rows (x, dy, dx_add) and the affine parameters are requested before the partials are reduced (see gn_apply_kernel)
    const bool act0 = slot0 < slots && rsub < rows_in_flight;
    const int cA = slot0 * 8, rA = r0 + rsub;
    const bool pre_ok = act0 && rA + rows_in_flight < r1;
    half8 gpre = {}, bpre = {}, px[2], pd[2], pa[2] = {};
    if (act0) { gpre = *(const half8*)(gamma + cA); bpre = *(const half8*)(beta + cA); }
    if (pre_ok) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t off = ((size_t)b * hw + rA + u * rows_in_flight) * C + cA;
            px[u] = *(const half8*)(x + off); pd[u] = *(const half8*)(dy + off);
            if (dx_add) pa[u] = *(const half8*)(dx_add + off);
        }
    }
    gn_reduce_partials(bpartials, b, n_chunks, st, scratch);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = slot0 + j * 256;
        if (slot >= slots || rsub >= rows_in_flight || (j == 1 && slots <= 256)) continue;
        const int c = slot * 8;
        half8 gv = gpre, bv = bpre;
        if (j == 1) { gv = *(const half8*)(gamma + c); bv = *(const half8*)(beta + c); }
        float mean[8], rstd[8], gm[8], bt[8], m1[8], m2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (c + k) / cg;
            mean[k] = fstats[(b * 32 + g) * 2] * inv_cnt;
            rstd[k] = rsqrtf(fmaxf(fstats[(b * 32 + g) * 2 + 1] * inv_cnt - mean[k] * mean[k], 0.f) + eps);
            gm[k] = (float)gv[k]; bt[k] = (float)bv[k];
            m1[k] = st[g * 2] * inv_cnt; m2[k] = st[g * 2 + 1] * inv_cnt;
        }
        auto emit = [&](const half8& xv, const half8& dv, const half8& av, int r) __attribute__((always_inline)) {
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = ((float)xv[k] - mean[k]) * rstd[k];
                float gg = (float)dv[k] * gm[k];
                if (silu) gg *= silu_grad(fmaf(xh, gm[k], bt[k]));
                o[k] = (half_t)(rstd[k] * (gg - m1[k] - xh * m2[k]));
            }
            if (dx_add) {   // the gradient of the block's other consumer (ResnetBlock shortcut) is accumulated here
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (half_t)((float)o[k] + (float)av[k]);
            }
            *(half8*)(dx + ((size_t)b * hw + r) * C + c) = o;
        };
        int r = r0 + rsub;
        if (j == 0 && pre_ok) {
            emit(px[0], pd[0], pa[0], r);
            emit(px[1], pd[1], pa[1], r + rows_in_flight);
            r += 2 * rows_in_flight;
        }
        for (; r + rows_in_flight < r1; r += 2 * rows_in_flight) {   // two rows (4-6 loads) in flight per thread
            const size_t off0 = ((size_t)b * hw + r) * C + c, off1 = off0 + (size_t)rows_in_flight * C;
            const half8 xv0 = *(const half8*)(x + off0), dv0 = *(const half8*)(dy + off0);
            const half8 xv1 = *(const half8*)(x + off1), dv1 = *(const half8*)(dy + off1);
            half8 av0 = {}, av1 = {};
            if (dx_add) { av0 = *(const half8*)(dx_add + off0); av1 = *(const half8*)(dx_add + off1); }
            emit(xv0, dv0, av0, r);
            emit(xv1, dv1, av1, r + rows_in_flight);
        }
        for (; r < r1; r += rows_in_flight) {
            const size_t off = ((size_t)b * hw + r) * C + c;
            half8 av = {};
            if (dx_add) av = *(const half8*)(dx_add + off);
            emit(*(const half8*)(x + off), *(const half8*)(dy + off), av, r);
        }
    }
}

// fp16 2-D transpose (rows x cols -> cols x rows), 64x64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_f16_kernel(const half_t* __restrict__ x, int rows, int cols, int ldx,
                                                            half_t* __restrict__ y, int ldy) {
    __shared__ half_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int q = threadIdx.x; q < 64 * 64; q += 256) {
        const int r = q / 64, c = q % 64;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? x[(size_t)(r0 + r) * ldx + c0 + c] : (half_t)0.f;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 64 * 64; q += 256) {
        const int c = q / 64, r = q % 64;
        if (c0 + c < cols && r0 + r < rows) y[(size_t)(c0 + c) * ldy + r0 + r] = tile[r][c];
    }
}

// ---- LayerNorm: a wave owns a row at a time (grid-stride over rows), NV = ceil(C / 512) 16-byte loads per lane
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, int rows, int C,
                                                        const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                        float eps, half_t* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    half8 gm[NV], bt[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 8 + j * 512;
        if (c < C) { gm[j] = *(const half8*)(gamma + c); bt[j] = *(const half8*)(beta + c); }
    }
    const float inv_c = 1.f / (float)C;
    for (int row = wave; row < rows; row += n_waves) {
        const half_t* xr = x + (size_t)row * C;
        half8 v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                v[j] = *(const half8*)(xr + c);
#pragma unroll
                for (int k = 0; k < 8; ++k) s += (float)v[j][k];
            }
        }
        const float mean = asd_wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (lane * 8 + j * 512 < C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = (float)v[j][k] - mean; q = fmaf(d, d, q); }
            }
        const float rstd = rsqrtf(asd_wave_sum(q) * inv_c + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = lane * 8 + j * 512;
            if (c < C) {
                half8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (half_t)(((float)v[j][k] - mean) * rstd * (float)gm[j][k] + (float)bt[j][k]);
                *(half8*)(y + (size_t)row * C + c) = o;
            }
        }
    }
}


// ---- row softmax (single-head attention of the VAE mid block, executed as GEMM -> softmax -> GEMM) -----------
// One 256-thread block per row, the row lives in registers (NV half8 vectors per thread), fp32 statistics.
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    const float a = red[0], b = red[1], c = red[2], d = red[3];
    return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_kernel(const half_t* __restrict__ x, int ldx, int cols, float scale,
                                                      half_t* __restrict__ y, int ldy) {
    __shared__ float red[4];
    const half_t* xr = x + (size_t)blockIdx.x * ldx;
    half8 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x * 8 + j * 2048;
        if (c < cols) {
            v[j] = *(const half8*)(xr + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) m = fmaxf(m, (float)v[j][k]);
        }
    }
    m = block_reduce(m, red, true) * scale;
    float e[NV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if (threadIdx.x * 8 + j * 2048 < cols) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { e[j][k] = __expf(fmaf((float)v[j][k], scale, -m)); s += e[j][k]; }
        }
    const float inv = 1.f / block_reduce(s, red, false);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x * 8 + j * 2048;
        if (c < cols) {
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (half_t)(e[j][k] * inv);
            *(half8*)(y + (size_t)blockIdx.x * ldy + c) = o;
        }
    }
}

// dS = scale * P o (dP - rowsum(dP o P))
template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const half_t* __restrict__ p, const half_t* __restrict__ dp, int ld,
                                                          int cols, float scale, half_t* __restrict__ ds) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * ld;
    half8 pv[NV], dv[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x * 8 + j * 2048;
        if (c < cols) {
            pv[j] = *(const half8*)(p + base + c);
            dv[j] = *(const half8*)(dp + base + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) s = fmaf((float)pv[j][k], (float)dv[j][k], s);
        }
    }
    s = block_reduce(s, red, false);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = threadIdx.x * 8 + j * 2048;
        if (c < cols) {
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (half_t)(scale * (float)pv[j][k] * ((float)dv[j][k] - s));
            *(half8*)(ds + base + c) = o;
        }
    }
}

__global__ __launch_bounds__(256) void geglu_kernel(const half_t* __restrict__ h, int rows, int C, half_t* __restrict__ y) {
    const int slots = C / 8;
    const size_t total = (size_t)rows * slots;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / slots;
        const int c = (int)(i - row * slots) * 8;
        const half8 a = *(const half8*)(h + row * 2 * C + c), g = *(const half8*)(h + row * 2 * C + C + c);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)((float)a[k] * gelu_f((float)g[k]));
        *(half8*)(y + row * C + c) = o;
    }
}

__global__ __launch_bounds__(256) void silu_kernel(const half_t* __restrict__ x, size_t n8, half_t* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const half8 v = *(const half8*)(x + i * 8);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)silu_f((float)v[k]);
        *(half8*)(y + i * 8) = o;
    }
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, half_t* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    const float freq = expf(-logf(10000.f) * (float)k / (float)half);
    const float a = t[r] * freq;
    y[(size_t)r * dim + k] = (half_t)cosf(a);
    y[(size_t)r * dim + half + k] = (half_t)sinf(a);
}

__global__ __launch_bounds__(256) void concat_kernel(const half_t* __restrict__ x1, int c1, const half_t* __restrict__ x2, int c2,
                                                     size_t rows, half_t* __restrict__ y) {
    const int C = c1 + c2, slots = C / 8;
    const size_t total = rows * slots;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / slots;
        const int c = (int)(i - row * slots) * 8;
        *(half8*)(y + row * C + c) = c < c1 ? *(const half8*)(x1 + row * c1 + c) : *(const half8*)(x2 + row * c2 + (c - c1));
    }
}

// y[r, :c] = (half) x[r, :c], y[r, c:c_pad] = 0: fp32 gradient of the 8 moment channels -> the K granularity of the dgrad GEMM
__global__ __launch_bounds__(256) void pad_cast_kernel(const float* __restrict__ x, size_t rows, int c, half_t* __restrict__ y, int c_pad) {
    const size_t total = rows * (size_t)c_pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / c_pad;
        const int j = (int)(i - r * c_pad);
        y[i] = j < c ? (half_t)x[r * c + j] : (half_t)0.f;
    }
}

// records[b][n][64] -> out[b][gridDim.y][64]: the per-tile statistics records of a producer's epilogue, summed in gridDim.y slices per
// batch element (the apply kernel's prologue adds the slices)
// 1024 threads = 16 record lanes x 64 values; 8 independent loads in flight per thread: with 4 lanes and one load per trip the 8192
// records of a 64x64-tiled 512x512 layer were 128 dependent L2/HBM round trips per block (125 us for 2 MB).
__global__ __launch_bounds__(1024) void gn_reduce_records_kernel(const float* __restrict__ records, int n, float* __restrict__ out) {
    __shared__ float scratch[16][64];
    const int b = blockIdx.x, S = gridDim.y, per = (n + S - 1) / S;
    const int c0 = blockIdx.y * per, c1 = min(n, c0 + per);
    const int v = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float* src = records + (size_t)b * n * 64 + v;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int c = c0 + q;
    for (; c + 7 * 16 < c1; c += 8 * 16) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(c + u * 16) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += t[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (c + u * 16 < c1) acc[u] += src[(size_t)(c + u * 16) * 64];
    scratch[q][v] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += scratch[u][v];
        out[((size_t)b * S + blockIdx.y) * 64 + v] = s;
    }
}

// apply blocks per batch element: one trip of the row loop (GN_UNR rows x the rows a block holds in flight) per block while that
// keeps the launch under GN_CAP_A blocks — the small low-resolution tensors (8x8 ... 32x32) are latency-, not bandwidth-bound, and a
// block that walks 16 rows of a 1280-channel tensor four at a time spends its life in four dependent memory round trips
static int gn_apply_chunks(int hw, int C, int batch) {
    const int slots = C / 8, rif = slots >= 256 ? 1 : 256 / slots;
    const int chunks = asd_div_up(hw, GN_UNR * rif), cap_a = asd_div_up(GN_CAP_A, batch);
    return chunks > cap_a ? cap_a : chunks;
}

// dst[i][:] = src[idx[i]][:] for rows of row16 16-byte words (batch-entry gather / broadcast of NHWC activations)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int* __restrict__ idx, size_t row16,
                                                          uint4* __restrict__ dst) {
    const size_t i = blockIdx.y;
    const uint4* s = src + (size_t)idx[i] * row16;
    uint4* d = dst + i * row16;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < row16; q += (size_t)gridDim.x * 256) d[q] = s[q];
}

extern "C" {

int asd_gather_rows_f16(const void* src, const int32_t* idx_dev, int32_t n_out, int64_t row_halfs, void* dst, void* stream) {
    ASD_CHECK_ARG(src && idx_dev && dst && n_out > 0 && row_halfs > 0 && row_halfs % 8 == 0, "bad argument");
    const size_t row16 = (size_t)row_halfs / 8;
    int bx = (int)((row16 + 1023) / 1024);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, n_out), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (const int*)idx_dev, row16,
                       (uint4*)dst);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_groupnorm_f16(const void* x1, int32_t c1, const void* x2, int32_t c2, int32_t batch, int32_t hw, const void* gamma,
                      const void* beta, float eps, int32_t silu, void* y, float* stats, void* stream) {
    ASD_CHECK_ARG(x1 && gamma && beta && y && stats && batch > 0 && hw > 0, "null argument");
    const int C = c1 + (x2 ? c2 : 0);
    if (!x2) c2 = 0;
    ASD_CHECK_ARG(C % 32 == 0 && C % 8 == 0 && c1 % 8 == 0, "channels must be a multiple of 32");
    ASD_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "gamma / beta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* partials = stats + 64 * batch;   // [batch, chunks_s, 64] per-block partial sums (plain stores: nothing to zero)
    // statistics: >= 16 rows per block and at most ~512 blocks (each ends with 64 atomics on one hot set of addresses);
    // apply: no atomics, so up to ~2048 blocks
    int chunks = asd_div_up(hw, 16);
    // every apply block sums its batch element's statistics partials in its prologue (chunks_s x 256 B from L2): keep
    // (apply blocks) x (statistics chunks) small — with 515 x 1280 blocks the prologue read 2.6x the tensor itself
    const int cap_s = asd_div_up(256, batch);
    const int chunks_s = chunks > cap_s ? cap_s : chunks, chunks_a = gn_apply_chunks(hw, C, batch);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(batch, chunks_s), dim3(256), 0, s, (const half_t*)x1, c1, (const half_t*)x2, c2, hw,
                       asd_div_up(hw, chunks_s), partials);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(batch, chunks_a), dim3(256), 0, s, (const half_t*)x1, c1, (const half_t*)x2, c2, hw,
                       asd_div_up(hw, chunks_a), (const half_t*)gamma, (const half_t*)beta, eps, silu, partials, chunks_s, stats,
                       (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_groupnorm_apply_f16(const void* x, int32_t c, int32_t batch, int32_t hw, const void* gamma, const void* beta, float eps,
                            int32_t silu, const float* partials, int32_t records, void* y, float* stats /* ASD_GN_STATS_FLOATS(batch) */, void* stream) {
    ASD_CHECK_ARG(x && gamma && beta && y && stats && partials && batch > 0 && hw > 0 && records > 0, "null argument");
    ASD_CHECK_ARG(c % 32 == 0 && c % 8 == 0, "channels must be a multiple of 32");
    ASD_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "gamma / beta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int chunks_a = gn_apply_chunks(hw, c, batch);
    // every apply block sums the records of its batch element in its prologue: with many tiles, sum them once first
    const float* part = partials;
    int n = records;
    if (records > GN_FOLD_RECORDS) {                  // slices of <= 32 records, at most 16 of them left for the apply prologue
        int slices = asd_div_up(records, 32);
        if (slices > 16) slices = 16;
        float* tmp = stats + 64 * batch;              // the per-block partials area of ASD_GN_STATS_FLOATS(batch)
        hipLaunchKernelGGL(gn_reduce_records_kernel, dim3(batch, slices), dim3(1024), 0, s, partials, records, tmp);
        part = tmp;
        n = slices;
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3(batch, chunks_a), dim3(256), 0, s, (const half_t*)x, c, (const half_t*)nullptr, 0, hw,
                       asd_div_up(hw, chunks_a), (const half_t*)gamma, (const half_t*)beta, eps, silu, part, n, stats, (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_groupnorm_bwd_f16(const void* x, const void* dy, int32_t c, int32_t batch, int32_t hw, const void* gamma,
                          const void* beta, float eps, int32_t silu, const float* fwd_stats, const void* dx_add, void* dx,
                          float* bwd_stats, void* stream) {
    ASD_CHECK_ARG(x && dy && gamma && beta && fwd_stats && dx && bwd_stats && batch > 0 && hw > 0, "null argument");
    ASD_CHECK_ARG(c % 32 == 0, "channels must be a multiple of 32");
    ASD_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "gamma / beta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int chunks = asd_div_up(hw, 16);
    // every apply block sums its batch element's statistics partials in its prologue (chunks_s x 256 B from L2): keep
    // (apply blocks) x (statistics chunks) small — with 515 x 1280 blocks the prologue read 2.6x the tensor itself
    const int cap_s = asd_div_up(256, batch), cap_a = asd_div_up(GN_CAP_A, batch);
    const int chunks_s = chunks > cap_s ? cap_s : chunks, chunks_a = chunks > cap_a ? cap_a : chunks;
    if ((size_t)asd_div_up(hw, chunks_s) * c >= 32768 && c % 8 == 0 && 512 % (c / 8) == 0)      // >= 64 KB of each tensor per block
        hipLaunchKernelGGL(gn_bwd_stats_kernel<512>, dim3(batch, chunks_s), dim3(512), 0, s, (const half_t*)x, (const half_t*)dy, c, hw,
                           asd_div_up(hw, chunks_s), (const half_t*)gamma, (const half_t*)beta, eps, silu, fwd_stats, bwd_stats);
    else
        hipLaunchKernelGGL(gn_bwd_stats_kernel<256>, dim3(batch, chunks_s), dim3(256), 0, s, (const half_t*)x, (const half_t*)dy, c, hw,
                           asd_div_up(hw, chunks_s), (const half_t*)gamma, (const half_t*)beta, eps, silu, fwd_stats, bwd_stats);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(batch, chunks_a), dim3(256), 0, s, (const half_t*)x, (const half_t*)dy, c, hw,
                       asd_div_up(hw, chunks_a), (const half_t*)gamma, (const half_t*)beta, eps, silu, fwd_stats, bwd_stats, chunks_s,
                       (const half_t*)dx_add, (half_t*)dx);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_groupnorm_bwd_apply_f16(const void* x, const void* dy, int32_t c, int32_t batch, int32_t hw, const void* gamma, const void* beta,
                                float eps, int32_t silu, const float* fwd_stats, const float* partials, int32_t records,
                                const void* dx_add, void* dx, float* scratch, void* stream) {
    ASD_CHECK_ARG(x && dy && gamma && beta && fwd_stats && partials && dx && scratch && batch > 0 && hw > 0 && records > 0, "null argument");
    ASD_CHECK_ARG(c % 32 == 0, "channels must be a multiple of 32");
    ASD_CHECK_ARG(((((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0, "gamma / beta must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = asd_div_up(hw, 16);
    const int cap_a = asd_div_up(GN_CAP_A, batch), chunks_a = chunks > cap_a ? cap_a : chunks;
    const float* part = partials;
    int n = records;
    if (records > GN_FOLD_RECORDS) {
        int slices = asd_div_up(records, 32);
        if (slices > 16) slices = 16;
        hipLaunchKernelGGL(gn_reduce_records_kernel, dim3(batch, slices), dim3(1024), 0, s, partials, records, scratch);
        part = scratch;
        n = slices;
    }
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(batch, chunks_a), dim3(256), 0, s, (const half_t*)x, (const half_t*)dy, c, hw,
                       asd_div_up(hw, chunks_a), (const half_t*)gamma, (const half_t*)beta, eps, silu, fwd_stats, part, n,
                       (const half_t*)dx_add, (half_t*)dx);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_transpose_f16(const void* x, int32_t rows, int32_t cols, int32_t ldx, void* y, int32_t ldy, void* stream) {
    ASD_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= rows, "bad argument");
    hipLaunchKernelGGL(transpose_f16_kernel, dim3(asd_div_up(cols, 64), asd_div_up(rows, 64)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, rows, cols, ldx, (half_t*)y, ldy);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_layernorm_f16(const void* x, int32_t rows, int32_t c, const void* gamma, const void* beta, float eps, void* y,
                      void* stream) {
    ASD_CHECK_ARG(x && gamma && beta && y && rows > 0, "null argument");
    ASD_CHECK_ARG(c % 8 == 0 && c <= 2048, "channels must be a multiple of 8 and <= 2048");
    int blocks = asd_div_up(rows, 4);
    if (blocks > 2048) blocks = 2048;
    const dim3 g(blocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(NV) hipLaunchKernelGGL((layernorm_kernel<NV>), g, blk, 0, s, (const half_t*)x, rows, c, (const half_t*)gamma, \
                                         (const half_t*)beta, eps, (half_t*)y)
    if (c <= 512) LN_LAUNCH(1);
    else if (c <= 1024) LN_LAUNCH(2);
    else if (c <= 1536) LN_LAUNCH(3);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_softmax_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, float scale, void* y, int32_t ldy, void* stream) {
    ASD_CHECK_ARG(x && y && rows > 0 && cols > 0, "null argument");
    ASD_CHECK_ARG(cols % 8 == 0 && cols <= 8192 && ldx % 8 == 0 && ldy % 8 == 0, "cols must be a multiple of 8 and <= 8192");
    const dim3 g(rows), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define SM_LAUNCH(NV) hipLaunchKernelGGL((softmax_kernel<NV>), g, blk, 0, s, (const half_t*)x, ldx, cols, scale, (half_t*)y, ldy)
    if (cols <= 2048) SM_LAUNCH(1);
    else if (cols <= 4096) SM_LAUNCH(2);
    else SM_LAUNCH(4);
#undef SM_LAUNCH
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_softmax_bwd_f16(const void* p, const void* dp, int32_t ld, int32_t rows, int32_t cols, float scale, void* ds,
                        void* stream) {
    ASD_CHECK_ARG(p && dp && ds && rows > 0 && cols > 0, "null argument");
    ASD_CHECK_ARG(cols % 8 == 0 && cols <= 8192 && ld % 8 == 0, "cols must be a multiple of 8 and <= 8192");
    const dim3 g(rows), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define SMB_LAUNCH(NV) hipLaunchKernelGGL((softmax_bwd_kernel<NV>), g, blk, 0, s, (const half_t*)p, (const half_t*)dp, ld, cols, scale, (half_t*)ds)
    if (cols <= 2048) SMB_LAUNCH(1);
    else if (cols <= 4096) SMB_LAUNCH(2);
    else SMB_LAUNCH(4);
#undef SMB_LAUNCH
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_geglu_f16(const void* h, int32_t rows, int32_t c, void* y, void* stream) {
    ASD_CHECK_ARG(h && y && rows > 0 && c % 8 == 0, "bad argument");
    hipLaunchKernelGGL(geglu_kernel, dim3(asd_grid_for((size_t)rows * c / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)h, rows, c, (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_silu_f16(const void* x, int64_t n, void* y, void* stream) {
    ASD_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "bad argument");
    hipLaunchKernelGGL(silu_kernel, dim3(asd_grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (size_t)(n / 8), (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_timestep_embedding_f16(const float* t, int32_t n, int32_t dim, void* y, void* stream) {
    ASD_CHECK_ARG(t && y && n > 0 && dim % 2 == 0, "bad argument");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(asd_div_up(n * dim / 2, 256)), dim3(256), 0, (hipStream_t)stream, t, n,
                       dim, (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_concat_f16(const void* x1, int32_t c1, const void* x2, int32_t c2, int64_t rows, void* y, void* stream) {
    ASD_CHECK_ARG(x1 && x2 && y && rows > 0 && c1 % 8 == 0 && c2 % 8 == 0, "bad argument");
    hipLaunchKernelGGL(concat_kernel, dim3(asd_grid_for((size_t)rows * (c1 + c2) / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x1, c1, (const half_t*)x2, c2, (size_t)rows, (half_t*)y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_pad_cast_f16(const float* x, int32_t rows, int32_t c, void* y, int32_t c_pad, void* stream) {
    ASD_CHECK_ARG(x && y && rows > 0 && c > 0 && c_pad >= c, "bad argument");
    const size_t total = (size_t)rows * c_pad;
    hipLaunchKernelGGL(pad_cast_kernel, dim3(asd_grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (size_t)rows, c, (half_t*)y, c_pad);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
