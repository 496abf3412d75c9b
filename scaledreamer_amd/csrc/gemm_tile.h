// gemm_tile.h — device helpers shared by the MFMA GEMM / convolution kernels of gemm.hip and gemm_pp.hip: fragment types,
// XCD-aware block order, GroupNorm statistics in the epilogue, the common tile epilogue.
#pragma once
#include "asd_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// exact-form GELU x * Phi(x) (torch F.gelu default; attention.py:49-56) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 —
// three orders below the fp16 rounding of the result): ~15 VALU instead of libm erff's ~60 on every element of a GEGLU epilogue
// (26 M elements in the 64^2-level feed-forward: a third of that launch)
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float tail = poly * t * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);    // 1 - erf(z)
    const float phi = x >= 0.f ? 1.f - 0.5f * tail : 0.5f * tail;                                // Phi(x)
    return x * phi;
}

// issue one 8-row x 128-B slab: lane -> (row = lane>>3, physical chunk = lane&7); LDS destination is linear
__device__ __forceinline__ void load_slab(const char* src_row_chunk, char* lds_slab_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src_row_chunk, (LDS_AS void*)lds_slab_base, 16, 0, 0);
}

// XCD-aware block -> work-item mapping.  Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each XCD
// has its own 4 MB L2, and everything an L2 misses comes over the fabric at ~7.3 TB/s for the whole chip — while tile loads that
// hit the L2 run at > 24 TB/s (tools/load_probe.hip: 5 co-XCD blocks streaming the same window 24.6 TB/s, 5 blocks with
// consecutive ids 7.3 TB/s).  So the blocks that share operand tiles must sit on the SAME XCD at the same time: XCD x takes the
// contiguous range [x * per, (x + 1) * per) of the work-item order, and that order walks compact (group_m x group_n)
// super-tiles, so the ~32-64 blocks an XCD runs concurrently read few distinct A and W tiles.
// Items are (split-K slice, tile) with the slice outermost.  Returns false for the padding blocks of the last XCD.
__device__ __forceinline__ bool asd_xcd_item(int bid, int items, int& item) {
    const int per = (items + 7) >> 3;
    item = (bid & 7) * per + (bid >> 3);
    return (bid >> 3) < per && item < items;
}
__device__ __forceinline__ void asd_grouped_tile(int t, int tiles_m, int tiles_n, int gm, int gn, int& tm, int& tn) {
    const int band = t / (gm * tiles_n), r = t - band * gm * tiles_n;
    const int gsz = min(gm, tiles_m - band * gm);            // rows of this band (the last band may be short)
    const int st = r / (gsz * gn), rr = r - st * gsz * gn;   // super-tile along N, index inside it
    tm = band * gm + rr % gsz;
    tn = st * gn + rr / gsz;
}

// One pipeline stage holds a BM x 64 A tile and a BN x 64 W tile with 128-byte LDS rows: a wave-level
// global_load_lds instruction covers 8 rows x one full 128-B cache line.  The 16-B chunk index is XOR-swizzled with
// (row & 7) on the source address and on the ds_read_b128 side (conflict-free).  Two stages (double buffer): the loads
// of tile k+1 are issued right after the barrier that publishes tile k.
//
// Measured on MI355X (main-loop ablation builds; numbers in DESIGN.md section 4): removing the MFMAs or the ds_reads from this loop does not change its time,
// removing the global->LDS tile loads makes it 1.5-2.6x faster, and every shape lands at ~8 TB/s of aggregate L2->LDS
// traffic.  The kernel is bound by bytes loaded per flop = (1/BM + 1/BN) / 128 B, so the tile is chosen as large as the
// problem allows (WM x WN waves, each owning a (BM/WM) x (BN/WN) register tile), up to 256 x 320.
// ---- GroupNorm statistics in the producer's epilogue -----------------------------------------------------------------------
// When p.gn_partials is set, the block that stores an output tile also reduces sum / sum of squares of the (fp16-rounded) values it
// stores per GroupNorm group (32 groups of p.gn_cg consecutive channels) and writes ONE 64-float record {sum_g, sumsq_g} at
// index  tile_m * tiles_n + tile_n  (plain stores: nothing to zero, no global atomics).  The consumer's GroupNorm then skips its
// statistics pass over the tensor (asd_groupnorm_apply_f16 sums the records).  A tile never straddles two batch elements: the
// host only enables this when the rows of a batch element are a multiple of the tile's rows (asd_gemm_gn_records).
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row; valid in lane 15 of the row
    int x;
#define ASD_ROW_ADD(CTRL) x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true); v += __int_as_float(x)
    ASD_ROW_ADD(0x111); ASD_ROW_ADD(0x112); ASD_ROW_ADD(0x114); ASD_ROW_ADD(0x118);
#undef ASD_ROW_ADD
    return v;
}
__device__ __forceinline__ void gn_tile_begin(float* lds64) {
    __syncthreads();                        // every wave is done with the main loop's LDS
    if (threadIdx.x < 64) lds64[threadIdx.x] = 0.f;
    __syncthreads();
}
// adds the statistics terms of the 4 stored values o (row m, channels n..n+3) to the lane's column sums: forward {v, v^2}, or — when
// p.gn_bwd_x is set — the GroupNorm input-gradient reductions {g, g * xhat} with g = dy * silu'(z) * gamma (o is dy)
__device__ __forceinline__ float silu_grad_f(float z) {
    const float sg = 1.f / (1.f + __expf(-z));
    return sg * (1.f + z * (1.f - sg));
}
struct GnCol {                 // per-column constants of the backward form, formed once per column fragment (the tile lies in ONE batch element)
    float mean[4], rstd[4], gm[4], bt[4];
};
__device__ __forceinline__ void gn_col_load(const asd_gemm_args& p, int m_any, int n, GnCol& c) {
    if (!p.gn_bwd_x || n >= p.N) return;
    const int b = m_any / p.gn_rows;
    const float inv_cnt = 1.f / ((float)p.gn_rows * (float)p.gn_cg);
    const half4 gm = *(const half4*)((const half_t*)p.gn_bwd_gamma + n), bt = *(const half4*)((const half_t*)p.gn_bwd_beta + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int g = (n + r) / p.gn_cg;
        c.mean[r] = p.gn_bwd_fstats[(b * 32 + g) * 2] * inv_cnt;
        c.rstd[r] = rsqrtf(fmaxf(p.gn_bwd_fstats[(b * 32 + g) * 2 + 1] * inv_cnt - c.mean[r] * c.mean[r], 0.f) + p.gn_eps);
        c.gm[r] = (float)gm[r]; c.bt[r] = (float)bt[r];
    }
}
__device__ __forceinline__ void gn_tile_accum(const asd_gemm_args& p, const GnCol& c, const floatx4& o, int m, int n, floatx4& cs, floatx4& cq) {
    if (!p.gn_bwd_x) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[r] += o[r]; cq[r] = fmaf(o[r], o[r], cq[r]); }
        return;
    }
    const half4 xv = *(const half4*)((const half_t*)p.gn_bwd_x + (size_t)m * p.N + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float xh = ((float)xv[r] - c.mean[r]) * c.rstd[r];
        float gg = o[r] * c.gm[r];
        if (p.gn_silu) gg *= silu_grad_f(fmaf(xh, c.gm[r], c.bt[r]));
        cs[r] += gg;
        cq[r] = fmaf(gg, xh, cq[r]);
    }
}
// s, q: this lane's column sums for channels n..n+3 (over its rows); lanes of one 16-lane row hold the same channels
__device__ __forceinline__ void gn_tile_flush(float* lds64, floatx4 s, floatx4 q, int n, int N, int cg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]); }
    if ((threadIdx.x & 15) == 15) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= N) continue;
            const int g = (n + r) / cg;
            atomicAdd(&lds64[2 * g], s[r]);
            atomicAdd(&lds64[2 * g + 1], q[r]);
        }
    }
}
__device__ __forceinline__ void gn_tile_end(const asd_gemm_args& p, const float* lds64, int record) {
    __syncthreads();
    if (threadIdx.x < 64) p.gn_partials[(size_t)record * 64 + threadIdx.x] = lds64[threadIdx.x];
}

// LayerNorm folded into its consumer (asd_gemm_args.ln_mode): v <- rstd * (v - mean * s) + c on CNT consecutive columns n.. of row m.
// mode 1: (mean, rstd) belong to the row (arguments), s / c to the columns; mode 2: s / c belong to the row, the statistics to the columns
// (read from ln_stats[N][2]).
template <int CNT>
__device__ __forceinline__ void ln_fold(const asd_gemm_args& p, float* v, int m, int n, float mean, float rstd) {
    if (p.ln_mode == 1) {
#pragma unroll
        for (int q = 0; q < CNT / 4; ++q) {
            const floatx4 s = *(const floatx4*)(p.ln_sc + n + 4 * q), c = *(const floatx4*)(p.ln_sc + p.N + n + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = fmaf(rstd, v[4 * q + r] - mean * s[r], c[r]);
        }
    } else {
        const float s = p.ln_sc[m], c = p.ln_sc[p.M + m];
#pragma unroll
        for (int q = 0; q < CNT / 2; ++q) {
            const floatx4 st = *(const floatx4*)(p.ln_stats + 2 * (n + 2 * q));      // {mean, rstd} of columns n + 2q, n + 2q + 1
            v[2 * q] = fmaf(st[1], v[2 * q] - st[0] * s, c);
            v[2 * q + 1] = fmaf(st[3], v[2 * q + 1] - st[2] * s, c);
        }
    }
}

// bias + row_bias + SiLU + residual + store of 4 consecutive output channels of row m (shared by all GEMM / conv kernels)
__device__ __forceinline__ floatx4 gemm_store4(const asd_gemm_args& p, floatx4 v, int m, int n) {
    if (p.bias) {
        const half4 b = *(const half4*)((const half_t*)p.bias + n);
        v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
    }
    if (p.row_bias) {
        const half4 b = *(const half4*)((const half_t*)p.row_bias + (size_t)(m / p.rows_per_group) * p.ld_row_bias + n);
        v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
    }
    if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.f + __expf(-v[r]));
    }
    if (p.residual) {
        const half4 b = *(const half4*)((const half_t*)p.residual + (size_t)m * p.ldr + n);
        v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
    }
    if (p.out_f32) {
        *(floatx4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
        return v;
    }
    half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *(half4*)((half_t*)p.C + (size_t)m * p.ldc + n) = o;
    return floatx4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};       // what a later pass over the tensor would read
}

// ---- wide-row epilogue -------------------------------------------------------------------------------------------------
// The 16x16x32 MFMA leaves a lane with 4 consecutive output channels of one row (pixel): with the W tile in natural row order a wave
// store covers 16 rows x 32 B.  When asd_gemm_args.wide_rows is set the W tile is brought into LDS in a permuted ROW order instead —
// inside every aligned group of 32 channels, LDS row jj*16 + g*4 + r holds channel g*8 + jj*4 + r — so fragments 2s and 2s+1 of a
// lane are 8 CONSECUTIVE channels: one 16-byte store (and one 16-byte bias / residual load) per row, 64 B runs per row and
// instruction, half the epilogue's memory instructions.  Only the global source row of each LDS row changes; fragment reads, the
// swizzle and the MFMA order are untouched, so results are bit-identical.
__device__ __forceinline__ int wide_slab_rows(int ws) { return (ws >> 2) * 32 + (ws & 1) * 16 + ((ws >> 1) & 1) * 4; }   // first channel of 8-row slab ws
__device__ __forceinline__ int wide_lane_row(int lrow) { return (lrow >> 2) * 8 + (lrow & 3); }                        // + this for row lrow of the slab

__device__ __forceinline__ void gemm_store8(const asd_gemm_args& p, floatx4& lo, floatx4& hi, int m, int n) {
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (p.bias) {
        const half8 b = *(const half8*)((const half_t*)p.bias + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)b[k];
    }
    if (p.row_bias) {
        const half8 b = *(const half8*)((const half_t*)p.row_bias + (size_t)(m / p.rows_per_group) * p.ld_row_bias + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)b[k];
    }
    if (p.act == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] / (1.f + __expf(-v[k]));
    }
    if (p.residual) {
        const half8 b = *(const half8*)((const half_t*)p.residual + (size_t)m * p.ldr + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)b[k];
    }
    if (p.out_f32) {
        float* dst = (float*)p.C + (size_t)m * p.ldc + n;
        *(floatx4*)dst = floatx4{v[0], v[1], v[2], v[3]};
        *(floatx4*)(dst + 4) = floatx4{v[4], v[5], v[6], v[7]};
        lo = floatx4{v[0], v[1], v[2], v[3]}; hi = floatx4{v[4], v[5], v[6], v[7]};
        return;
    }
    const half8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
    *(half8*)((half_t*)p.C + (size_t)m * p.ldc + n) = o;
    lo = floatx4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]}; hi = floatx4{(float)o[4], (float)o[5], (float)o[6], (float)o[7]};
}

// Epilogue of a wave's TM x TN fragment tile (all GEMM / conv kernels): split-K partial slabs, or bias / row_bias / SiLU / residual /
// store, plus the GroupNorm reductions when asked.  nb = first channel of the wave, row(i) = output row of fragment row i (< 0: none),
// row0 = any row of the tile (the batch element of the GroupNorm constants).
// ln_mean / ln_rstd: statistics of the lane's row in fragment row i (ln_mode 1; null otherwise)
template <int TM, int TN, typename RowFn>
__device__ __forceinline__ void tile_epilogue(const asd_gemm_args& p, floatx4 (&acc)[TM][TN], int nb, int kz, int row0, RowFn row, bool gn, float* gn_lds,
                                              const float* ln_mean = nullptr, const float* ln_rstd = nullptr) {
    const int g = (threadIdx.x & 63) >> 4;
    if constexpr (TN % 2 == 0) {
        if (p.wide_rows) {
#pragma unroll
            for (int s2 = 0; s2 < TN / 2; ++s2) {
                const int n = nb + s2 * 32 + g * 8;
                floatx4 cs0 = {0.f, 0.f, 0.f, 0.f}, cq0 = cs0, cs1 = cs0, cq1 = cs0;     // forward statistics only (the backward form keeps
                                                                                         // 16 constants per 4 channels live: narrow path)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = row(i);
                    if (m < 0 || n >= p.N) continue;
                    if (p.split_k > 1) {
                        float* dst = p.workspace + ((size_t)kz * p.M + m) * p.N + n;
                        *(floatx4*)dst = acc[i][2 * s2];
                        *(floatx4*)(dst + 4) = acc[i][2 * s2 + 1];
                    } else {
                        floatx4 lo = acc[i][2 * s2], hi = acc[i][2 * s2 + 1];
                        if (p.ln_mode) {
                            float v8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            ln_fold<8>(p, v8, m, n, ln_mean ? ln_mean[i] : 0.f, ln_rstd ? ln_rstd[i] : 0.f);
                            lo = floatx4{v8[0], v8[1], v8[2], v8[3]}; hi = floatx4{v8[4], v8[5], v8[6], v8[7]};
                        }
                        gemm_store8(p, lo, hi, m, n);
                        if (gn) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                cs0[r] += lo[r]; cq0[r] = fmaf(lo[r], lo[r], cq0[r]);
                                cs1[r] += hi[r]; cq1[r] = fmaf(hi[r], hi[r], cq1[r]);
                            }
                        }
                    }
                }
                if (gn) { gn_tile_flush(gn_lds, cs0, cq0, n, p.N, p.gn_cg); gn_tile_flush(gn_lds, cs1, cq1, n + 4, p.N, p.gn_cg); }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = nb + j * 16 + g * 4;
        floatx4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
        GnCol gc;
        if (gn) gn_col_load(p, row0, n, gc);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = row(i);
            if (m < 0 || n >= p.N) continue;
            if (p.split_k > 1) *(floatx4*)(p.workspace + ((size_t)kz * p.M + m) * p.N + n) = acc[i][j];
            else {
                floatx4 a4 = acc[i][j];
                if (p.ln_mode) {
                    float v4[4] = {a4[0], a4[1], a4[2], a4[3]};
                    ln_fold<4>(p, v4, m, n, ln_mean ? ln_mean[i] : 0.f, ln_rstd ? ln_rstd[i] : 0.f);
                    a4 = floatx4{v4[0], v4[1], v4[2], v4[3]};
                }
                const floatx4 o = gemm_store4(p, a4, m, n);
                if (gn) gn_tile_accum(p, gc, o, m, n, cs, cq);
            }
        }
        if (gn) gn_tile_flush(gn_lds, cs, cq, n, p.N, p.gn_cg);
    }
}

// wait until at most min(ahead, MAX) tiles of PER loads each are still in flight (s_waitcnt takes an immediate)
template <int PER, int MAX>
__device__ __forceinline__ void ring_wait(int ahead) {
    if constexpr (MAX <= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (ahead >= MAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAX * PER) : "memory");
        else ring_wait<PER, MAX - 1>(ahead);
    }
}
