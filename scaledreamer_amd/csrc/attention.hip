// attention.hip — flash-style multi-head attention forward for the SD-2.1 UNet (head_dim 64, fp16 in/out,
// fp32 softmax and accumulation) on gfx950.  Replaces CrossAttention.forward's einsum/softmax/einsum
// (extern/mvdream/ldm/modules/attention.py:168-194; diffusers' attention processors on the SD path):
// self-attention over L = 4096/1024/256/64 tokens and cross-attention over 77 text tokens.
//
// Everything is computed "transposed" so the probabilities never leave registers:
//   S^T = K Q^T   (MFMA A = K fragment from LDS, B = Q fragment held in VGPRs)
//         -> lane (l&15) owns ONE query column, its 4 accumulator rows are 4 consecutive keys
//   O^T = V^T P^T (MFMA A = V^T fragment from LDS, B = P^T = exp(S^T - m) converted to fp16 in place)
//         -> the same lane owns the same query, 4 consecutive head-dim rows -> 8-byte stores
// The k-index of the second MFMA is a permutation of the key order (two 4-key groups of two 16-key
// sub-tiles); V^T fragments are read with the same permutation, so no LDS round trip / transpose of P.
// V arrives already transposed ([channel][token]) from its projection GEMM (operands swapped).
// K and V^T tiles (64 keys) stream through LDS with global_load_lds, double buffered; 16-B chunks are
// XOR-swizzled (K: chunk ^ (row&7) for ds_read_b128, V^T: chunk ^ ((row>>1)&7) for ds_read_b64).
#include "asd_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

typedef float float2_ __attribute__((ext_vector_type(2)));

// max of three (one VALU instruction; exact).  Written as asm because fmaxf() makes the compiler canonicalise every operand that
// comes out of an MFMA with an extra v_max x, x: 52 instructions for the 16-score maximum instead of 8.
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fma1(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Reductions over the four 16-lane rows of the wave (the lanes l, l^16, l^32, l^48 hold the same query column): gfx950's row swaps
// (v_permlane16_swap / v_permlane32_swap: VALU, no LDS round trip) instead of two ds_bpermute + s_waitcnt lgkmcnt(0) each.
__device__ __forceinline__ float rows_max(float v) {
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max3f(__uint_as_float(t[0]), __uint_as_float(t[1]), -3.0e38f);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max3f(__uint_as_float(t[0]), __uint_as_float(t[1]), -3.0e38f);
}
__device__ __forceinline__ float rows_sum(float v) {
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

#define HD 64      // head dim
#ifndef ATTN_DEFER_LOG2
#define ATTN_DEFER_LOG2 8.0f   // 0: rescale whenever a maximum moves (the first form)
#endif
#define KT 64      // keys per tile
#ifndef ATTN_SCALAR_FMA
#define ATTN_SCALAR_FMA 1
#endif
#ifndef ATTN_SUM_MFMA
#define ATTN_SUM_MFMA 1   // 0: row sums on the vector pipe (the first form; A/B partner)
#endif
// queries per wave = 16 * QS, per block = 64 * QS (4 waves).  QS = 2: 128 registers, four waves per SIMD.  QS = 4 (self-attention
// over >= 2048 keys): every K / V^T fragment read from LDS and every tile brought from L2 serves twice the MFMAs, half the barriers
// per flop; 218 registers, two waves per SIMD: 210 -> 196 us at L = 4096 (same-box A/B), nothing at L = 1024, slower below.
// Measured and NOT kept on the wide form: the loop body as one basic block (unconditional accumulator rescale) with
// sched_group_barrier pipelines asking for the second query group's MFMAs under the first group's softmax — the scheduler keeps
// the MFMAs clustered and the kernel is 2-5 % slower (200-206 us).

struct AttnArgs {
    const half_t* q; int ldq;
    const half_t* k; int ldk;
    const half_t* vT; int ldv;
    half_t* o; int ldo;
    int batch, heads, lq, lk, lk_stride;
    float scale;
    const char* zero;
};

// launch bounds: with plain __launch_bounds__(256) hipcc parked the MFMA results in AGPRs and moved 192 registers per key tile
// between the two files (v_accvgpr_read / _write around the softmax: 330 VALU instructions per tile instead of ~140); naming
// 4 waves per SIMD keeps everything in 128 VGPRs.  L = 4096: 265 -> 199 us.
template <int QS>
__global__ __launch_bounds__(256, QS == 2 ? 4 : 2) void attention_fwd_kernel(const AttnArgs p) {
    constexpr int QW = 16 * QS, QB = 4 * QW;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile 8 KB | V^T tile 8 KB]
    constexpr int TILE_BYTES = KT * HD * 2;                         // 8192
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order (block i runs on XCD i % 8, one L2 per XCD): XCD x takes a contiguous range of (batch, head, q-block)
    // items with the q-block innermost, so the q-blocks of one head stream that head's K / V^T through ONE L2 instead of eight
    const int qblocks = (p.lq + QB - 1) / QB, items = qblocks * p.heads * p.batch;
    const int per = (items + 7) >> 3, item = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || item >= items) return;
    const int b = item / (qblocks * p.heads), h = (item / qblocks) % p.heads;
    const int q0 = (item % qblocks) * QB + wave * QW;
    const int lq16 = lane & 15, lg = lane >> 4;

    // ---- Q fragments (B operand): lane holds Q[q = q0 + 16*qs + (l&15)][d = 32*ks + (l>>4)*8 .. +8] --------
    half8 qf[QS][2];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        int qi = q0 + qs * 16 + lq16;
        if (qi >= p.lq) qi = p.lq - 1;
        const half_t* src = p.q + ((size_t)b * p.lq + qi) * p.ldq + h * HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qs][ks] = *(const half8*)(src + ks * 32 + lg * 8);
    }

    // ---- tile loader: per wave 2 K slabs + 2 V^T slabs of 8 rows x 128 B --------------------------------
    const int srow = lane >> 3, pch = lane & 7;
    const size_t kbase = (size_t)b * p.lk_stride;
    auto issue = [&](int t, int buf) {
        char* Ks = smem + buf * 2 * TILE_BYTES;
        char* Vs = Ks + TILE_BYTES;
        const int key0 = t * KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + srow;  // key row of the K tile / d row of the V^T tile
            {   // K: logical chunk = physical ^ (row & 7)
                const int c = pch ^ (row & 7);
                const int key = key0 + row;
                const char* src = key < p.lk_stride ? (const char*)(p.k + (kbase + key) * p.ldk + h * HD) + c * 16 : p.zero;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(Ks + (wave * 2 + i) * 8 * 128), 16, 0, 0);
            }
            {   // V^T: logical chunk = physical ^ ((row >> 1) & 7); chunk c covers keys key0 + 8c .. +8
                const int c = pch ^ ((row >> 1) & 7);
                const char* src = key0 + c * 8 < p.lk_stride
                                      ? (const char*)(p.vT + (size_t)(h * HD + row) * p.ldv + kbase + key0) + c * 16 : p.zero;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)(Vs + (wave * 2 + i) * 8 * 128), 16, 0, 0);
            }
        }
    };

    floatx4 oacc[QS][4];  // [qs][dt]: O[q = 16qs + (l&15)][d = 16dt + (l>>4)*4 + r]
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[qs][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) m_run[qs] = -1e30f;
#if ATTN_SUM_MFMA
    // Row sums on the matrix pipe: one more V^T fragment whose row 0 is all ones makes sum_k P[q][k] row 0 of a fifth output tile — the loop issues
    // ~112 vector slots per 16 scores (64 of them the exponentials) against 64 MFMAs per tile, and the matrix pipe is a third busy: 13 slots per query
    // group and tile (pairwise adds, the cross-lane sum, the running-sum update) become two MFMAs.  The sum is then the sum of the fp16-rounded
    // probabilities that multiply V (the normaliser of exactly what was accumulated).  Lane (q, lg = 0) holds it in register 0.
    // Same-box: attention per step 1.35 -> 1.24 ms (tools/attn_bench.py), the step -0.085 ms in three pairs (gpurun_out/attn1/ab.txt).
    // Measured on top of it and NOT kept: scale and shift folded into the S MFMA (Q pre-multiplied by scale * log2 e, accumulator started at
    // -m_run: L = 4096 179 -> 164 us, but a rounding of Q per element that costs 2e-2 absolute on the large-score test — the probabilities of
    // scores ~60 move by 1 %); the rescale factor behind a wave-uniform branch (nothing); k-step 0's V^T fragments requested in front of the
    // softmax (187.5 -> 185.6 us).
    floatx4 lacc[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) lacc[qs] = floatx4{0.f, 0.f, 0.f, 0.f};
    const half_t one_or_zero = lq16 == 0 ? (half_t)1.f : (half_t)0.f;
    const half8 ones_row0 = {one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero, one_or_zero};
#else
    float l_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) l_run[qs] = 0.f;
#endif

    const int n_tiles = (p.lk + KT - 1) / KT;
    const float sl2 = p.scale * 1.44269504088896340736f;   // softmax in the log2 domain
    issue(0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) issue(t + 1, buf ^ 1);
        const char* Ks = smem + buf * 2 * TILE_BYTES;
        const char* Vs = Ks + TILE_BYTES;

        // ---- S^T = K Q^T --------------------------------------------------------------------------------
        floatx4 s[4][QS];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int row = kt * 16 + lq16;
            half8 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = *(const half8*)(Ks + row * 128 + (((ks * 4 + lg) ^ (row & 7)) * 16));
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) {
                floatx4 a = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[qs][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[qs][1], a, 0, 0, 0);
                s[kt][qs] = a;
            }
        }
        // ---- online softmax per query column -------------------------------------------------------------------
        // VALU-lean form (the loop is VALU-bound: 32 scores per lane and tile): scores are kept in the log2 domain
        // (one fma folds scale * log2(e) and the running max, v_exp_f32 is 2^x natively), the key-bound mask runs only on
        // the last tile, and the accumulator rescale is skipped while no
        // lane's running max moved.
        const int key_base = t * KT + lg * 4;
        half8 pb[2][QS];  // probabilities, fp16: [j][qs] = B operand of PV k-step j: keys of sub-tiles kt = 2j (elements 0-3) and 2j + 1 (4-7)
        if (t == n_tiles - 1 && (p.lk & (KT - 1)) != 0) {
            asm volatile("" ::: "memory");   // keep this a (wave-uniform) branch: if-converted it costs 32 selects on every tile
#pragma unroll
            for (int qs = 0; qs < QS; ++qs)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key_base + kt * 16 + r >= p.lk) s[kt][qs][r] = -1e30f;
        }
#pragma unroll
        for (int qs = 0; qs < QS; ++qs) {
            // 16 scores per lane -> 8 x v_max3_f32 (depth 3); fmaxf() costs a canonicalising v_max per MFMA-produced operand on top
            const float a0 = max3f(s[0][qs][0], s[0][qs][1], s[0][qs][2]), a1 = max3f(s[0][qs][3], s[1][qs][0], s[1][qs][1]);
            const float a2 = max3f(s[1][qs][2], s[1][qs][3], s[2][qs][0]), a3 = max3f(s[2][qs][1], s[2][qs][2], s[2][qs][3]);
            const float a4 = max3f(s[3][qs][0], s[3][qs][1], s[3][qs][2]);
            float mx = max3f(max3f(a0, a1, a2), max3f(a3, a4, s[3][qs][3]), -3.0e38f);
            mx = rows_max(mx);                                       // over the 4 lane groups that share this query column
            // Deferred running maximum (cdna_hip_programming.md T13): while no query's tile maximum exceeds its running maximum by more than
            // DEFER (log2 domain: P <= 2^DEFER = 256, exact in fp16's range and precision), the old maximum stays and nothing is rescaled —
            // on random data the accumulator rescale then runs on the first tiles only.  All of this tile's P and its row sum use the
            // maximum decided HERE, before any of them is formed, and the previous tile's P V has completed (loop order).
            const float cand = mx * sl2;
            const bool grow = __builtin_amdgcn_ballot_w64(cand > m_run[qs] + ATTN_DEFER_LOG2) != 0;     // wave-uniform
            const float m_new = grow ? max3f(m_run[qs], cand, -3.0e38f) : m_run[qs];
            const float alpha = grow ? __builtin_amdgcn_exp2f(m_run[qs] - m_new) : 1.f;
            const float2_ sl2v = {sl2, sl2}, nm = {-m_new, -m_new};
#if !ATTN_SUM_MFMA
            float2_ sum2 = {0.f, 0.f};
#endif
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                // wide form: single v_fma_f32 (asm: -O3 would pack neighbours into v_pk_fma_f32 again, which costs more than two plain FMAs beside
                // MFMAs: MI355X_MICROARCH.md, per-instruction cycle constants — L = 4096: 179 -> 171 us); narrow form (four waves per SIMD at 128
                // registers): v_pk_fma_f32, two exponents per instruction (with single FMAs L = 1024 goes 31 -> 38 us)
                float2_ x01, x23;
                if constexpr (QS == 4 && ATTN_SCALAR_FMA) {
                    x01 = float2_{fma1(s[kt][qs][0], sl2, -m_new), fma1(s[kt][qs][1], sl2, -m_new)};
                    x23 = float2_{fma1(s[kt][qs][2], sl2, -m_new), fma1(s[kt][qs][3], sl2, -m_new)};
                } else {
                    x01 = __builtin_elementwise_fma(__builtin_shufflevector(s[kt][qs], s[kt][qs], 0, 1), sl2v, nm);
                    x23 = __builtin_elementwise_fma(__builtin_shufflevector(s[kt][qs], s[kt][qs], 2, 3), sl2v, nm);
                }
                const float2_ e01 = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1])};
                const float2_ e23 = {__builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
#if !ATTN_SUM_MFMA
                sum2 += e01 + e23;
#endif
                half8& dst = pb[kt >> 1][qs];                              // v_cvt_pk_f16_f32 (RNE) x2, written in place
                dst[(kt & 1) * 4 + 0] = (half_t)e01[0]; dst[(kt & 1) * 4 + 1] = (half_t)e01[1];
                dst[(kt & 1) * 4 + 2] = (half_t)e23[0]; dst[(kt & 1) * 4 + 3] = (half_t)e23[1];
            }
#if !ATTN_SUM_MFMA
            const float sum = rows_sum(sum2[0] + sum2[1]);
            l_run[qs] = l_run[qs] * alpha + sum;
#endif
            if (grow) {   // wave-uniform: some query's maximum moved by more than the deferral
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    oacc[qs][dt][0] *= alpha; oacc[qs][dt][1] *= alpha; oacc[qs][dt][2] *= alpha; oacc[qs][dt][3] *= alpha;
                }
#if ATTN_SUM_MFMA
                lacc[qs][0] *= alpha;
#endif
            }
            m_run[qs] = m_new;
        }
        // ---- O^T += V^T P^T : k-step j covers keys 32j..32j+31 in the permuted order --------------------------
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int row = dt * 16 + lq16;
                const int sw = (row >> 1) & 7;
                // keys 32j + 4*lg .. +4 live in chunk (4j + lg/2), byte (lg&1)*8; keys +16 in chunk (4j + 2 + lg/2)
                const int c_lo = (4 * j + (lg >> 1)) ^ sw, c_hi = (4 * j + 2 + (lg >> 1)) ^ sw;
                // volatile: two ds_read_b64 (2 LDS cycles each).  Left to itself the compiler pairs the reads of DIFFERENT fragments
                // into ds_read2st64_b64 (8 cycles per pair) and then needs 6 v_mov per pair to sort the halves out.
                const half4 v_lo = *(const volatile LDS_AS half4*)(Vs + row * 128 + c_lo * 16 + (lg & 1) * 8);
                const half4 v_hi = *(const volatile LDS_AS half4*)(Vs + row * 128 + c_hi * 16 + (lg & 1) * 8);
                const half8 vf = half8{v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
#pragma unroll
                for (int qs = 0; qs < QS; ++qs) oacc[qs][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pb[j][qs], oacc[qs][dt], 0, 0, 0);
            }
#if ATTN_SUM_MFMA
#pragma unroll
            for (int qs = 0; qs < QS; ++qs) lacc[qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones_row0, pb[j][qs], lacc[qs], 0, 0, 0);
#endif
        }
        __syncthreads();  // next tile landed (vmcnt(0)) and everyone is done with this buffer
    }

    // ---- normalise and store: lane owns query (l&15), 4 consecutive d -----------------------------------------
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qi = q0 + qs * 16 + lq16;
#if ATTN_SUM_MFMA
        const float inv = 1.f / __shfl(lacc[qs][0], lq16, 64);          // lane (q, lg = 0) = lane q holds the query's sum (all lanes take part)
#else
        const float inv = 1.f / l_run[qs];
#endif
        if (qi >= p.lq) continue;
        half_t* dst = p.o + ((size_t)b * p.lq + qi) * p.ldo + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const floatx4 v = oacc[qs][dt];
            const half4 o = {(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv)};
            *(half4*)(dst + dt * 16 + lg * 4) = o;
        }
    }
}

extern "C" {

int asd_attention_f16(const void* q, int32_t ldq, const void* k, int32_t ldk, const void* vT, int32_t ldv, void* o,
                      int32_t ldo, int32_t batch, int32_t heads, int32_t lq, int32_t lk, int32_t lk_stride, float scale,
                      const void* zero_page, void* stream) {
    ASD_CHECK_ARG(q && k && vT && o && zero_page, "null argument");
    ASD_CHECK_ARG(batch > 0 && heads > 0 && lq > 0 && lk > 0 && lk_stride >= lk, "bad sizes");
    ASD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && lk_stride % 8 == 0,
                  "leading dimensions / key stride must keep 16-byte alignment");
    AttnArgs a{(const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)vT, ldv, (half_t*)o, ldo,
               batch, heads, lq, lk, lk_stride, scale, (const char*)zero_page};
    static const int qs_env = getenv("ASD_ATTN_QS") ? atoi(getenv("ASD_ATTN_QS")) : 0;     // A/B switch (tools): force 2 or 4
    const bool wide = qs_env ? qs_env == 4 : (lk >= 2048 && lq >= 2048);
    if (wide) {
        const dim3 grid(8 * asd_div_up(asd_div_up(lq, 256) * heads * batch, 8));
        hipLaunchKernelGGL(attention_fwd_kernel<4>, grid, dim3(256), 4 * KT * HD * 2, (hipStream_t)stream, a);
    } else {
        const dim3 grid(8 * asd_div_up(asd_div_up(lq, 128) * heads * batch, 8));
        hipLaunchKernelGGL(attention_fwd_kernel<2>, grid, dim3(256), 4 * KT * HD * 2, (hipStream_t)stream, a);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
