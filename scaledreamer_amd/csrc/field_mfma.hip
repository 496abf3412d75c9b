// field_mfma.hip — the two 32 -> 64 -> {1, 3} heads of the implicit-volume field, backward, on the matrix pipe (round 6).
//
// What it replaces: the MLP half of field_bwd_sample_kernel (csrc/field.hip) — per sample 8192 scalar-operand FMAs to re-evaluate the hidden
// layers from the saved encoding, form the hidden-layer gradients and back-propagate them to the encoding (tcnn's fully-fused MLP backward
// behind threestudio/models/networks.py:214-251 VanillaMLP; call site geometry/implicit_volume.py:109-207), which ran at ~18 % of the vector rate
// and was the larger half of the field-gradient span.  Here a wave owns 64 samples and every product is a
// v_mfma_f32_32x32x16_f16 in split-fp16 arithmetic (x * s = hi + lo with s a power of two, product = hi hi + hi lo + lo hi, fp32 accumulate: 22
// significant bits per operand — the scheme of csrc/conv3d.hip / trifield_mfma.hip):
//
//   Z^T  = W1 E^T        [64 h x 64 samples]   samples are the MFMA's N columns, so a lane (sample, half) keeps its own sample's hidden units:
//                                               the per-sample power-of-two scale of E factors out of the contraction, and only sign(Z^T) is needed
//   dA^T = [Z^T > 0] g    g = draw[s] w2d[h]  (density head)  |  sum_o df[s][o] w2f[o][h]  (feature head)
//   dE^T = W1^T dA^T     [32 k x 64 samples]   dA^T in the accumulator layout IS the B operand of this product in a permuted k (= h) order,
//                                               in which the W1^T fragments are gathered once per wave: nothing goes through LDS
//   Z    = E W1^T         [64 samples x 64 h]   the same fragments with the operands swapped: hidden units are columns (lanes), samples rows
//                                               (registers), so the second-layer weight gradients sum_s g_o[s] relu(a[s][h]) are in-lane sums
//
// A v_permlane32_swap per register pair turns the accumulator layout (lane = sample mod 32, half = which 16 of the 32 / 64 rows) into whole rows
// per lane: lane L leaves with the 32 encoding gradients and the 128 hidden-layer gradients of sample L as 16-byte stores of contiguous
// rows.  Outputs: DA[row][128] (the first-layer weight gradient's operand: field_wgrad_kernel), dE[row][32] (scattered into the hash table by
// field_bwd_sample_kernel<..., PRE = true>), dW2 (block sums + one atomic per weight and block).
// ReLU masks: the sign of the 22-bit pre-activation, except within its error bound of zero, where the unit is re-evaluated with the forward
// kernel's own fp32 FMA chain — the masks are the forward pass's masks (a sign taken from the split product alone flipped ~1 unit in 2e6 and
// left 186 of 12.6 M table-gradient entries outside 1e-4 of the oracle's: tests/test_gpu_renderer_kernels.py::test_field_backward).
#include <math.h>

#include "asd_common.h"

typedef _Float16 fm_h;
typedef fm_h fm_h8 __attribute__((ext_vector_type(8)));
typedef float fm_f16x __attribute__((ext_vector_type(16)));
typedef float fm_f4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float fm_pow2_scale(float amax) {       // 2^(13 - floor(log2 amax)): amax * scale in [2^13, 2^14); 1 for amax == 0
    if (!(amax > 1e-30f)) return 1.f;              // (zero rows; magnitudes below fp16's reach after any scale contribute nothing)
    const int e = (int)((__float_as_uint(amax) >> 23) & 255u) - 127;      // floor(log2 amax) for normal numbers
    return __uint_as_float((unsigned)(127 + 13 - e) << 23);
}
__device__ __forceinline__ void fm_split8(const float (&x)[8], float s, fm_h8& hi, fm_h8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = x[j] * s;
        const fm_h h = (fm_h)v;
        hi[j] = h;
        lo[j] = (fm_h)(v - (float)h);
    }
}
__device__ __forceinline__ fm_f16x fm_mma3(const fm_h8 ah, const fm_h8 al, const fm_h8 bh, const fm_h8 bl, fm_f16x acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    return acc;
}
// x: lanes 32-63 <-> y: lanes 0-31 (v_permlane32_swap)
__device__ __forceinline__ void fm_swap(float& x, float& y) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(t[0]);
    y = __uint_as_float(t[1]);
}

constexpr int FM_H = 64, FM_K = 32, FM_C = 3;

struct FmArgs {
    asd_field_cfg c;
    const float* w1[2];          // [64][32] density, feature
    const float* w2d;            // [64]
    const float* w2f;            // [3][64]
    const float* enc;            // [n][32]
    const float* sigma;          // [n]
    const float* d_sigma;        // [n] or NULL
    const float* d_features;     // [n][3] or NULL
    const int* n_dev;
    int n;
    float* da_out;               // [n][128]
    float* denc_out;             // [n][32]
    float* dw2d;                 // [64]   +=
    float* dw2f;                 // [3][64] +=
    float* dw1_slabs;            // WG: [gridDim][2][64][32]
};

template <bool WG>
__global__ __launch_bounds__(256, WG ? 1 : 2) void field_bwd_mlp_mfma_kernel(const FmArgs a) {
    __shared__ float w1s[2][FM_H][FM_K];          // 16 KB: fp32 weights (the transposed fragments are gathered from here)
    __shared__ float w2s[4][FM_H];                // w2d, w2f[0..2]
    __shared__ float gs[4][4][64];                // per wave: g_o[s] * inv scale of E (o = 0: draw, 1..3: df) for the Z pass
    __shared__ float red[4][4][FM_H];             // per wave: second-layer weight-gradient sums
    __shared__ fm_h8 frag[16][2][64];             // weight fragments (hi, lo), lane-linear
    __shared__ float gp[WG ? 4 : 1][4][64];       // WG: per wave, the plain g_o[s]
    __shared__ unsigned mb[WG ? 4 : 1][64][2];    // WG: per wave, every sample's 64-bit ReLU mask of the current head
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int nn = a.n_dev ? min(*a.n_dev, a.n) : a.n;
    if ((int)blockIdx.x * 256 >= nn) {             // capacity-sized launch: nothing lives in this block (its first tiles would be empty)
        if constexpr (WG) {                        // ... but the slab reduction sums every wave's slab
            float* slab = a.dw1_slabs + (size_t)blockIdx.x * (2 * FM_H * FM_K);
            for (int q = tid; q < 2 * FM_H * FM_K; q += 256) slab[q] = 0.f;
        }
        return;
    }
    for (int q = tid; q < 2 * FM_H * FM_K; q += 256) (&w1s[0][0][0])[q] = a.w1[q / (FM_H * FM_K)][q % (FM_H * FM_K)];
    if (tid < FM_H) w2s[0][tid] = a.w2d[tid];
    else if (tid < 4 * FM_H) w2s[tid / FM_H][tid % FM_H] = a.w2f[tid - FM_H];
    __syncthreads();

    // ---- weight fragments, once per wave ---------------------------------------------------------------------------------------------------
    // wf[hd][mt][ks]: rows h = 32 mt + l31, k = 16 ks + 8 half + j   (A operand of Z^T, B operand of Z)
    // wt[hd][mt][tp]: rows k = l31 (encoding index), slot j <-> h = 32 mt + 8 (2 tp + j / 4) + 4 half + j % 4   (A operand of dE^T: the order in
    //                 which a lane's accumulator registers of Z^T hold their hidden units)
    float wmax[2] = {0.f, 0.f};
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
        for (int q = lane; q < FM_H * FM_K; q += 64) wmax[hd] = fmaxf(wmax[hd], fabsf((&w1s[hd][0][0])[q]));
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax[hd] = fmaxf(wmax[hd], __shfl_xor(wmax[hd], off, 64));
    const float sw[2] = {fm_pow2_scale(wmax[0]), fm_pow2_scale(wmax[1])};
    // the 16 (hi, lo) fragment pairs live in LDS (32 KB, lane-linear: conflict-free 16-byte reads), not in 128 registers per lane: the kernel then fits
    // two waves per SIMD.  Wave w builds pairs 4 w .. 4 w + 3; index f = ((kind * 2 + hd) * 2 + mt) * 2 + t, kind 0 = wf, 1 = wt.
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
        const int f = wave * 4 + ff, kind = f >> 3, hd = (f >> 2) & 1, mt = (f >> 1) & 1, t = f & 1;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            x[j] = kind == 0 ? w1s[hd][32 * mt + l31][16 * t + 8 * half + j] : w1s[hd][32 * mt + 8 * (2 * t + j / 4) + 4 * half + (j & 3)][l31];
        fm_h8 hi, lo;
        fm_split8(x, sw[hd], hi, lo);
        frag[f][0][lane] = hi;
        frag[f][1][lane] = lo;
    }
    __syncthreads();
#define WF_H(hd, mt, t) frag[((0 * 2 + (hd)) * 2 + (mt)) * 2 + (t)][0][lane]
#define WF_L(hd, mt, t) frag[((0 * 2 + (hd)) * 2 + (mt)) * 2 + (t)][1][lane]
#define WT_H(hd, mt, t) frag[((1 * 2 + (hd)) * 2 + (mt)) * 2 + (t)][0][lane]
#define WT_L(hd, mt, t) frag[((1 * 2 + (hd)) * 2 + (mt)) * 2 + (t)][1][lane]

    float dw1a[2][16], dw1b[2][16];      // WG: first-layer weight gradients of the wave's tiles, [ht][4 q + r] = dW1[32 ht + l31][8 q + 4 half + r] (density / feature)
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw1a[ht][r] = dw1b[ht][r] = 0.f;
    float dw2[4][2];           // [o][ht]: lanes < 32 after the half sum: hidden unit 32 ht + l31
#pragma unroll
    for (int o = 0; o < 4; ++o) dw2[o][0] = dw2[o][1] = 0.f;

    // a wave walks tiles of 64 samples: the staging of the weights and the fragment images (~25 KB of LDS traffic and a block barrier) is paid once
    // per block, not once per 256 samples
    const int n_tiles = (nn + 63) >> 6;
#pragma unroll 1
    for (int tile = (int)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int)gridDim.x * 4) {
        const int s0 = tile * 64;                 // first sample of the wave's tile
        // ---- the tile's encodings as fragments: ef[nt][ks]: sample 32 nt + l31, k = 16 ks + 8 half + j ------------------------------------
        fm_h8 efh[2][2], efl[2][2];
        float se[2];                               // power-of-two scale of the lane's sample in tile nt (both halves agree)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int i = s0 + 32 * nt + l31;
            float x[2][8];
            float amax = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (i < nn) {
                    const fm_f4* src = reinterpret_cast<const fm_f4*>(a.enc + (size_t)i * FM_K + 16 * ks + 8 * half);
                    const fm_f4 v0 = src[0], v1 = src[1];
                    x[ks][0] = v0[0]; x[ks][1] = v0[1]; x[ks][2] = v0[2]; x[ks][3] = v0[3];
                    x[ks][4] = v1[0]; x[ks][5] = v1[1]; x[ks][6] = v1[2]; x[ks][7] = v1[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[ks][j] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(x[ks][j]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            se[nt] = fm_pow2_scale(amax);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fm_split8(x[ks], se[nt], efh[nt][ks], efl[nt][ks]);
        }
        // ---- per-sample output gradients: g[0] = d sigma * act'(raw) (from sigma itself), g[1..3] = d features -------------------------------
        float g[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int i = s0 + 32 * nt + l31;
            const bool live = i < nn;
            const float s = live ? a.sigma[i] : 0.f;
            const float ds = (live && a.d_sigma) ? a.d_sigma[i] : 0.f;
            float ag;
            if (a.c.activation == ASD_ACT_SOFTPLUS) ag = 1.f - expf(-s);
            else if (a.c.activation == ASD_ACT_EXP) ag = s;
            else if (a.c.activation == ASD_ACT_TRUNC_EXP) ag = fminf(s, 3269017.37f /* e^15 */);
            else ag = 1.f;
            g[nt][0] = ds * ag;
#pragma unroll
            for (int o = 0; o < FM_C; ++o) g[nt][1 + o] = (live && a.d_features) ? a.d_features[(size_t)i * FM_C + o] : 0.f;
            if (half == 0) {                       // for the Z pass: rows are samples, so every lane needs every sample's g (already divided by the scales)
#pragma unroll
                for (int o = 0; o < 4; ++o) gs[wave][o][32 * nt + l31] = g[nt][o] / (se[nt] * sw[o == 0 ? 0 : 1]);
                if constexpr (WG) {
#pragma unroll
                    for (int o = 0; o < 4; ++o) gp[wave][o][32 * nt + l31] = g[nt][o];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();           // gs is private to the wave: its LDS operations complete in order, the compiler must not reorder them

        float dencT[2][16];                        // sum over both heads of dE^T, true scale: [nt][register]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dencT[nt][r] = 0.f;
        // WG: E^T fragments for the first-layer weight gradient dW1^T = E^T dA (contraction over the tile's samples: ONE scale for the tile).
        // et[st][tp]: row k = l31 (encoding index), slot j <-> sample 32 st + 8 (2 tp + j / 4) + 4 half + j % 4 — the order in which a lane's
        // accumulator registers of Z hold their samples; for a fixed j the 32 lanes of a half read one sample's 128-byte row.
        fm_h8 eth[WG ? 2 : 1][2], etl[WG ? 2 : 1][2];
        float sT = 1.f;
        if constexpr (WG) {
            float x[2][2][8];
            float amax = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = s0 + 32 * st + 8 * (2 * tp + j / 4) + 4 * half + (j & 3);
                        const float v = i < nn ? a.enc[(size_t)i * FM_K + l31] : 0.f;
                        x[st][tp][j] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
            sT = fm_pow2_scale(amax);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) fm_split8(x[st][tp], sT, eth[st][tp], etl[st][tp]);
        }

#pragma unroll 1                    // (one head at a time: both heads unrolled side by side need more than the 512 registers of a lone wave)
        for (int hd = 0; hd < 2; ++hd) {
            // ---- Z^T = W1 E^T -> ReLU masks -> dA^T (true scale) -------------------------------------------------------------------------------------
            float da[2][2][16];                    // [mt][nt][4 q + r]: hidden unit 32 mt + 8 q + 4 half + r of sample 32 nt + l31
            float damax[2] = {0.f, 0.f};
            unsigned mword[2][2] = {{0u, 0u}, {0u, 0u}};      // [nt][mt]: this lane's nibbles of its sample's 64-bit ReLU mask (bit = hidden unit)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    fm_f16x z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) z = fm_mma3(WF_H(hd, mt, ks), WF_L(hd, mt, ks), efh[nt][ks], efl[nt][ks], z);
                    // ReLU masks.  sign(z) decides, except where |z| is within the error bound of the 22-bit products (2^12 in scaled units: 32 terms
                    // of at most 2^28, operands good to 2^-22): those few units (~1 in 10^4) re-evaluate the pre-activation with the forward
                    // kernel's own fp32 FMA chain (field.hip: mlp1 / mlpC, k ascending from zero), so the masks ARE the forward pass's masks.
                    unsigned pos = 0u;
                    float zmin = 3.0e38f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { pos |= (z[r] > 0.f ? 1u : 0u) << r; zmin = fminf(zmin, fabsf(z[r])); }
                    const int si = s0 + 32 * nt + l31;
#ifdef FM_ABL_NOSLOW
                    const bool unsure = false;
#else
                    const bool unsure = zmin < 4096.f && si < nn;
#endif
                    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {
                        if (unsure) {
                            const float* erow = a.enc + (size_t)si * FM_K;
#pragma unroll 1
                            for (int r = 0; r < 16; ++r) {
                                float zr = 0.f;
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr) zr = rr == r ? z[rr] : zr;
                                if (fabsf(zr) < 4096.f) {
                                    const float* wrow = &w1s[hd][32 * mt + 8 * (r >> 2) + 4 * half + (r & 3)][0];
                                    float acc1 = 0.f;
#pragma unroll 4
                                    for (int k = 0; k < FM_K; ++k) acc1 = fmaf(wrow[k], erow[k], acc1);
                                    pos = (pos & ~(1u << r)) | ((acc1 > 0.f ? 1u : 0u) << r);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int hb = 32 * mt + 8 * q + 4 * half;
                        if constexpr (WG) mword[nt][mt] |= ((pos >> (4 * q)) & 15u) << (8 * q + 4 * half);
                        float gh[4];
                        if (hd == 0) {
                            const fm_f4 w = *reinterpret_cast<const fm_f4*>(&w2s[0][hb]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) gh[r] = g[nt][0] * w[r];
                        } else {
                            const fm_f4 w0 = *reinterpret_cast<const fm_f4*>(&w2s[1][hb]), w1 = *reinterpret_cast<const fm_f4*>(&w2s[2][hb]),
                                        w2 = *reinterpret_cast<const fm_f4*>(&w2s[3][hb]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) gh[r] = fmaf(g[nt][3], w2[r], fmaf(g[nt][2], w1[r], g[nt][1] * w0[r]));
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = ((pos >> (4 * q + r)) & 1u) ? gh[r] : 0.f;
                            da[mt][nt][4 * q + r] = v;
                            damax[nt] = fmaxf(damax[nt], fabsf(v));
                        }
                    }
                }
            if constexpr (WG) {                    // every sample's whole mask, for the Z pass (lanes are hidden units there)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const unsigned w0 = mword[nt][0] | __shfl_xor(mword[nt][0], 32, 64), w1 = mword[nt][1] | __shfl_xor(mword[nt][1], 32, 64);
                    if (half == 0) { mb[wave][32 * nt + l31][0] = w0; mb[wave][32 * nt + l31][1] = w1; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            float sd[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                damax[nt] = fmaxf(damax[nt], __shfl_xor(damax[nt], 32, 64));
                sd[nt] = fm_pow2_scale(damax[nt]);
            }
            // ---- dE^T += W1^T dA^T: k-step (mt, tp) takes registers 8 tp .. 8 tp + 7 of dA^T[mt][nt] as its B fragment ----------------------------
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                fm_f16x d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int tp = 0; tp < 2; ++tp) {
                        float x[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[j] = da[mt][nt][8 * tp + j];
                        fm_h8 bh, bl;
                        fm_split8(x, sd[nt], bh, bl);
                        d = fm_mma3(WT_H(hd, mt, tp), WT_L(hd, mt, tp), bh, bl, d);
                    }
                const float inv = 1.f / (sw[hd] * sd[nt]);
#pragma unroll
                for (int r = 0; r < 16; ++r) dencT[nt][r] = fmaf(d[r], inv, dencT[nt][r]);
            }
            if constexpr (!WG) {
                // ---- DA rows: lane L leaves with hidden units 32 mt .. + 31 of sample L (swap: registers of tile 0 / tile 1 -> low / high four) --------
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) fm_swap(da[mt][0][r], da[mt][1][r]);
                    const int i = s0 + lane;
#ifdef FM_ABL_NODA
                    if (i < nn && da[mt][0][0] == 12345.f) {
#else
                    if (i < nn) {
#endif
                        float* dst = a.da_out + (size_t)i * (2 * FM_H) + hd * FM_H + 32 * mt;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            *reinterpret_cast<fm_f4*>(dst + 8 * q) = fm_f4{da[mt][0][4 * q], da[mt][0][4 * q + 1], da[mt][0][4 * q + 2], da[mt][0][4 * q + 3]};
                            *reinterpret_cast<fm_f4*>(dst + 8 * q + 4) = fm_f4{da[mt][1][4 * q], da[mt][1][4 * q + 1], da[mt][1][4 * q + 2], da[mt][1][4 * q + 3]};
                        }
                    }
                }
            }
            // ---- Z = E W1^T (hidden units are lanes, samples registers): second-layer weight gradients as in-lane sums; WG: dA in this layout is
            //      the B operand of dW1^T = E^T dA ------------------------------------------------------------------------------------------------
#ifndef FM_ABL_NOZ
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                float daz[WG ? 2 : 1][16];
                float dzmax = 0.f;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    fm_f16x z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) z = fm_mma3(efh[st][ks], efl[st][ks], WF_H(hd, ht, ks), WF_L(hd, ht, ks), z);
                    // z[4 q + r] = scaled pre-activation of (sample 32 st + 8 q + 4 half + r, hidden unit 32 ht + l31)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sb = 32 * st + 8 * q + 4 * half;
                        unsigned mk = 0u;
                        if constexpr (WG) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) mk |= ((mb[wave][sb + r][ht] >> l31) & 1u) << r;
                        }
                        if (hd == 0) {
                            const fm_f4 gv = *reinterpret_cast<const fm_f4*>(&gs[wave][0][sb]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) dw2[0][ht] = fmaf(fmaxf(z[4 * q + r], 0.f), gv[r], dw2[0][ht]);
                            if constexpr (WG) {
                                const fm_f4 gq = *reinterpret_cast<const fm_f4*>(&gp[wave][0][sb]);
                                const float w = w2s[0][32 * ht + l31];
#pragma unroll
                                for (int r = 0; r < 4; ++r) daz[st][4 * q + r] = ((mk >> r) & 1u) ? gq[r] * w : 0.f;
                            }
                        } else {
#pragma unroll
                            for (int o = 1; o < 4; ++o) {
                                const fm_f4 gv = *reinterpret_cast<const fm_f4*>(&gs[wave][o][sb]);
#pragma unroll
                                for (int r = 0; r < 4; ++r) dw2[o][ht] = fmaf(fmaxf(z[4 * q + r], 0.f), gv[r], dw2[o][ht]);
                            }
                            if constexpr (WG) {
                                const fm_f4 g1 = *reinterpret_cast<const fm_f4*>(&gp[wave][1][sb]), g2 = *reinterpret_cast<const fm_f4*>(&gp[wave][2][sb]),
                                            g3 = *reinterpret_cast<const fm_f4*>(&gp[wave][3][sb]);
                                const float w0 = w2s[1][32 * ht + l31], w1 = w2s[2][32 * ht + l31], w2 = w2s[3][32 * ht + l31];
#pragma unroll
                                for (int r = 0; r < 4; ++r) daz[st][4 * q + r] = ((mk >> r) & 1u) ? fmaf(g3[r], w2, fmaf(g2[r], w1, g1[r] * w0)) : 0.f;
                            }
                        }
                        if constexpr (WG) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) dzmax = fmaxf(dzmax, fabsf(daz[st][4 * q + r]));
                        }
                    }
                }
                if constexpr (WG) {
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) dzmax = fmaxf(dzmax, __shfl_xor(dzmax, off, 64));
                    const float sD = fm_pow2_scale(dzmax);
                    fm_f16x d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int tp = 0; tp < 2; ++tp) {
                            float x[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) x[j] = daz[st][8 * tp + j];
                            fm_h8 bh, bl;
                            fm_split8(x, sD, bh, bl);
                            d = fm_mma3(eth[st][tp], etl[st][tp], bh, bl, d);
                        }
                    // d[4 q + r] = scaled dW1[hidden unit 32 ht + l31][k = 8 q + 4 half + r]
                    const float inv = 1.f / (sT * sD);
                    if (hd == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dw1a[ht][r] = fmaf(d[r], inv, dw1a[ht][r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dw1b[ht][r] = fmaf(d[r], inv, dw1b[ht][r]);
                    }
                }
            }
#endif
        }
        // ---- dE rows --------------------------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) fm_swap(dencT[0][r], dencT[1][r]);
        const int i = s0 + lane;
        if (i < nn) {
            float* dst = a.denc_out + (size_t)i * FM_K;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<fm_f4*>(dst + 8 * q) = fm_f4{dencT[0][4 * q], dencT[0][4 * q + 1], dencT[0][4 * q + 2], dencT[0][4 * q + 3]};
                *reinterpret_cast<fm_f4*>(dst + 8 * q + 4) = fm_f4{dencT[1][4 * q], dencT[1][4 * q + 1], dencT[1][4 * q + 2], dencT[1][4 * q + 3]};
            }
        }
    }
    if constexpr (WG) {
        // the block's slab [head][hidden unit][k] (the layout of field_wgrad_kernel's slabs: slab_reduce_kernel sums the blocks' slabs in a fixed order):
        // the four waves' sums meet in the 32 KB of the fragment images, which nobody reads any more
        __syncthreads();
        float* buf = reinterpret_cast<float*>(&frag[0][0][0]) + (wave & 1) * (2 * FM_H * FM_K);
        auto put = [&](bool add) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = (32 * ht + l31) * FM_K + 8 * q + 4 * half;
                    fm_f4 va = fm_f4{dw1a[ht][4 * q], dw1a[ht][4 * q + 1], dw1a[ht][4 * q + 2], dw1a[ht][4 * q + 3]};
                    fm_f4 vb = fm_f4{dw1b[ht][4 * q], dw1b[ht][4 * q + 1], dw1b[ht][4 * q + 2], dw1b[ht][4 * q + 3]};
                    if (add) {
                        const fm_f4 pa = *reinterpret_cast<fm_f4*>(buf + o), pb = *reinterpret_cast<fm_f4*>(buf + FM_H * FM_K + o);
                        va[0] += pa[0]; va[1] += pa[1]; va[2] += pa[2]; va[3] += pa[3];
                        vb[0] += pb[0]; vb[1] += pb[1]; vb[2] += pb[2]; vb[3] += pb[3];
                    }
                    *reinterpret_cast<fm_f4*>(buf + o) = va;
                    *reinterpret_cast<fm_f4*>(buf + FM_H * FM_K + o) = vb;
                }
        };
        if (wave < 2) put(false);
        __syncthreads();
        if (wave >= 2) put(true);
        __syncthreads();
        const float* b0 = reinterpret_cast<const float*>(&frag[0][0][0]);
        float* slab = a.dw1_slabs + (size_t)blockIdx.x * (2 * FM_H * FM_K);
        for (int q = tid; q < 2 * FM_H * FM_K / 4; q += 256) {
            const fm_f4 x = reinterpret_cast<const fm_f4*>(b0)[q], y = reinterpret_cast<const fm_f4*>(b0 + 2 * FM_H * FM_K)[q];
            reinterpret_cast<fm_f4*>(slab)[q] = fm_f4{x[0] + y[0], x[1] + y[1], x[2] + y[2], x[3] + y[3]};
        }
    }
    // ---- second-layer weight gradients: halves, waves, one atomic per weight and block -----------------------------------------------------------------
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const float v = dw2[o][ht] + __shfl_xor(dw2[o][ht], 32, 64);
            if (half == 0) red[wave][o][32 * ht + l31] = v;
        }
    __syncthreads();
    {
        const int o = tid >> 6, h = tid & 63;
        const float v = (red[0][o][h] + red[1][o][h]) + (red[2][o][h] + red[3][o][h]);
        if (o == 0) atomicAdd(&a.dw2d[h], v);
        else if (a.dw2f) atomicAdd(&a.dw2f[(o - 1) * FM_H + h], v);
    }
}

}  // namespace

// MLP half of asd_field_bwd on the matrix pipe (no finite-difference normal, 16 levels x 2 features, 64 hidden units, 3 feature outputs).
// da_out [n, 128], denc_out [n, 32]; dw2d / dw2f are accumulated into.
int asd_field_bwd_mlp_mfma_blocks(int32_t n) {
    // one per CU: the all-in-one form holds one wave per SIMD (442 registers), so 512 blocks were two rounds of 256 with twice the weight staging
    // and twice the slabs to reduce (same box, span of 140 k samples: 0.338 -> 0.326 ms; 1024 blocks 0.35)
    static const int max_blocks = getenv("ASD_FIELD_MFMA_BLOCKS") ? atoi(getenv("ASD_FIELD_MFMA_BLOCKS")) : 256;
    return asd_div_up(n, 256) < max_blocks ? asd_div_up(n, 256) : max_blocks;
}

// dw1_slabs != NULL: the first-layer weight gradients leave as asd_field_bwd_mlp_mfma_blocks(n) slabs of [2][64][32] floats (one per block) and
// da_out is not written; NULL: da_out [n, 128] is, for field_wgrad_kernel
int asd_field_bwd_mlp_mfma(const asd_field_cfg* cfg, const float* w1d, const float* w2d, const float* w1f, const float* w2f, const float* enc, const float* sigma,
                           int32_t n, const int32_t* n_dev, const float* d_sigma, const float* d_features, float* da_out, float* denc_out, float* dw2d, float* dw2f,
                           float* dw1_slabs, hipStream_t s) {
    FmArgs a;
    a.c = *cfg;
    a.w1[0] = w1d; a.w1[1] = w1f; a.w2d = w2d; a.w2f = w2f; a.enc = enc; a.sigma = sigma; a.d_sigma = d_sigma; a.d_features = d_features;
    a.n_dev = n_dev; a.n = n; a.da_out = da_out; a.denc_out = denc_out; a.dw2d = dw2d; a.dw2f = dw2f; a.dw1_slabs = dw1_slabs;
    const int blocks = asd_field_bwd_mlp_mfma_blocks(n);
    if (dw1_slabs) hipLaunchKernelGGL(field_bwd_mlp_mfma_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(field_bwd_mlp_mfma_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
    return ASD_OK;
}
