// gemm.hip — fp16 MFMA GEMM / implicit-GEMM convolution for the frozen SD-2.1 UNet (gfx950).
//
//   C[M,N] = act( A[M,K] . W[N,K]^T + bias[N] + row_bias[m / rows_per_group][N] ) + residual[M,N]
//
// A is either a row-major activation matrix (Linear / 1x1 conv on NHWC tensors) or an implicit im2col view
// of an NHWC tensor for 3x3 convolutions (stride 1/2, optional fused nearest-2x upsample, zero padding):
// k = (ky, kx, cin), weights pre-packed [Cout][ky][kx][Cin].  Replaces the cuDNN/cuBLAS calls behind
// diffusers' UNet2DConditionModel in the reference (stable_diffusion_asd_guidance.py:319-331; layer
// inventory SURVEY.md Appendix A.1).
//
// CDNA4 mapping: 256 threads = 4 waves (2x2), block tile 128 x BN (BN = 128 | 64), BK = 32 = one
// v_mfma_f32_16x16x32_f16 step, fp32 accumulation.  Both operands stream HBM/L2 -> LDS with
// global_load_lds (16 B per lane, no VGPR round trip), double buffered, one barrier per k-step.  LDS rows are
// 64 B; the 16-B chunk index is XOR-swizzled with bit 3 of the row (st_16x32) on the SOURCE address and on
// the ds_read_b128 side, which makes the fragment reads bank-conflict free.  The MFMA is issued with the
// weight fragment as the A operand, so each lane ends up with 4 consecutive output channels of one row
// -> 8-byte stores.  Out-of-range rows / taps read a zero page instead of branching.
#include <stdlib.h>

#include "asd_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

#define BK 32

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// issue one 8-row x 128-B slab: lane -> (row = lane>>3, physical chunk = lane&7); LDS destination is linear
__device__ __forceinline__ void load_slab(const char* src_row_chunk, char* lds_slab_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src_row_chunk, (LDS_AS void*)lds_slab_base, 16, 0, 0);
}

// One pipeline stage holds a 128 x 64 A tile and a BN x 64 W tile with 128-byte LDS rows: a wave-level
// global_load_lds instruction then covers 8 rows x one full 128-B cache line (64-B rows touch twice as many lines
// per byte, and the texture-address path, not the MFMA pipe, limits this kernel).  The 16-B chunk index is
// XOR-swizzled with (row & 7) on the source address and on the ds_read_b128 side (conflict-free).  A k-step
// issues 2 x TM x TN MFMAs per wave.  NST stages form a ring: with NST = 3 the loads of tile k+2 are issued while
// tile k is consumed and only tile k+1 is waited for (counted s_waitcnt vmcnt + raw s_barrier; a
// __syncthreads() would drain the LDS-DMA queue).  K % 64 != 0 (K = 32-multiples) uses the KHALF variant that
// leaves the second half of the last row chunk to the zero page.
template <int BM, int BN, bool CONV, int NST>
__global__ __launch_bounds__(BM * 2) void gemm_f16_kernel(const asd_gemm_args p) {
    constexpr int NW = BM / 32;               // waves: (BM/64) x 2
    constexpr int TM = 4;                     // 64 rows per wave in m
    constexpr int TN = BN / 32;               // BN/2 columns per wave in n
    constexpr int RB = 128;                   // LDS row bytes (64 halfs)
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ASLABS = BM / 8, WSLABS = BN / 8;       // 8-row slabs (1 KiB per wave instruction)
    constexpr int AS = ASLABS / NW;                        // A slabs per wave (4)
    constexpr int WS = (WSLABS + NW - 1) / NW;             // W slabs per wave (upper bound)
    constexpr int LOADS_PER_TILE = AS + WS;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NST][A | W]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // tile order: consecutive blocks walk down M inside one N panel, so a weight panel stays hot in L2
    const int tiles_m = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;
    const int kz = blockIdx.z;  // split-K slice
    const int k_steps_total = (p.K + 63) / 64;
    const int k_per = (k_steps_total + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = kz * k_per, ks1 = min(k_steps_total, ks0 + k_per);

    // ---- per-lane source descriptors: lane -> (row = lane>>3 of an 8-row slab, physical chunk = lane&7) -------
    const int lrow = lane >> 3, pchunk = lane & 7;
    const char* zero = (const char*)p.zero_page;
    const char* a_base[AS];
    int a_y[AS], a_x[AS], a_valid[AS], a_lch[AS];
    const char* a_img[AS];
#pragma unroll
    for (int i = 0; i < AS; ++i) {
        const int row = (wave * AS + i) * 8 + lrow;       // row inside the block tile
        const int m = m0 + row;
        a_lch[i] = pchunk ^ (row & 7);                     // logical 16-B chunk (8 halfs) that lands in this slot
        a_valid[i] = m < p.M;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, r = m - b * hw;
            a_y[i] = r / p.Wout;
            a_x[i] = r - a_y[i] * p.Wout;
            a_img[i] = (const char*)p.A + (size_t)b * p.Hin * p.Win * p.Cin * 2;
            a_base[i] = nullptr;
        } else {
            a_base[i] = (const char*)p.A + (size_t)m * p.lda * 2 + a_lch[i] * 16;
        }
    }
    const char* w_base[WS];
    int w_valid[WS], w_lch[WS];
#pragma unroll
    for (int i = 0; i < WS; ++i) {
        const int slab = wave * WS + i;
        const int row = slab * 8 + lrow;
        const int n = n0 + row;
        w_lch[i] = pchunk ^ (row & 7);
        w_valid[i] = n < p.N && slab < WSLABS;
        w_base[i] = (const char*)p.W + (size_t)n * p.ldw * 2 + w_lch[i] * 16;
    }

    auto issue = [&](int ks, int stage) {
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + A_BYTES;
        const int k0 = ks * 64;
#pragma unroll
        for (int i = 0; i < AS; ++i) {
            const char* src;
            const int kc = k0 + a_lch[i] * 8;               // first k of this lane's chunk
            if (CONV) {
                const int tap = kc / p.Cin, c0 = kc - tap * p.Cin;   // chunks never straddle a tap (Cin % 8 == 0)
                const int ky = tap / 3, kx = tap - ky * 3;
                int yi = a_y[i] * p.stride + ky - p.pad, xi = a_x[i] * p.stride + kx - p.pad;
                bool ok = a_valid[i] && kc < p.K;
                if (p.upsample == 1) {  // conv over the nearest-2x upsampled image: bounds in the upsampled frame
                    ok = ok && yi >= 0 && xi >= 0 && yi < 2 * p.Hin && xi < 2 * p.Win;
                    yi >>= 1; xi >>= 1;
                } else if (p.upsample == 2) {
                    // input gradient of a stride-2 convolution: dX[y,x] += dY[(y+pad-ky)/2, (x+pad-kx)/2] W[ky,kx]
                    // for the taps where both numerators are even (the other taps read the zero page)
                    yi = a_y[i] + p.pad - ky; xi = a_x[i] + p.pad - kx;
                    ok = ok && yi >= 0 && xi >= 0 && !(yi & 1) && !(xi & 1) && (yi >> 1) < p.Hin && (xi >> 1) < p.Win;
                    yi >>= 1; xi >>= 1;
                } else {
                    ok = ok && yi >= 0 && xi >= 0 && yi < p.Hin && xi < p.Win;
                }
                src = ok ? a_img[i] + ((size_t)(yi * p.Win + xi) * p.Cin + c0) * 2 : zero;
            } else {
                src = (a_valid[i] && kc < p.K) ? a_base[i] + (size_t)k0 * 2 : zero;
            }
            load_slab(src, As + (wave * AS + i) * 8 * RB);
        }
#pragma unroll
        for (int i = 0; i < WS; ++i) {
            if (wave * WS + i >= WSLABS) continue;  // wave-uniform: fewer W slabs than waves
            const char* src = (w_valid[i] && k0 + w_lch[i] * 8 < p.K) ? w_base[i] + (size_t)k0 * 2 : zero;
            load_slab(src, Bs + (wave * WS + i) * 8 * RB);
        }
    };

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment reads: lane -> row (lane&15) of a 16-row sub-tile, k-quarter (lane>>4); chunk = kh*4 + quarter
    const int frow = lane & 15, fq = lane >> 4;

    const int nk = ks1 - ks0;
    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) issue(ks0 + s, s);
        for (int k = 0; k < nk; ++k) {
            // tile k must have landed; with a 3-stage ring the newest prefetch (tile k+1) may stay in flight
            if (NST >= 3 && k + 1 < nk && WSLABS % NW == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // tile k visible block-wide; everyone is done reading tile k-1
            if (k + NST - 1 < nk) issue(ks0 + k + NST - 1, (k + NST - 1) % NST);
            const char* As = smem + (k % NST) * STAGE_BYTES;
            const char* Bs = As + A_BYTES;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                half8 xa[TM], wb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * 64 + i * 16 + frow;
                    xa[i] = *(const half8*)(As + row * RB + (((kh * 4 + fq) ^ (row & 7)) * 16));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * (BN / 2) + j * 16 + frow;
                    wb[j] = *(const half8*)(Bs + row * RB + (((kh * 4 + fq) ^ (row & 7)) * 16));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // acc[i][j][r] = C[m = m0 + wm*64 + i*16 + (lane&15)][n = n0 + wn*BN/2 + j*16 + (lane>>4)*4 + r]
    const int em = lane & 15, en = (lane >> 4) * 4;
    if (gridDim.z > 1) {  // split-K: fp32 partial slabs, finished by splitk_epilogue_kernel
        float* ws = p.workspace + (size_t)kz * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * 64 + i * 16 + em;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + en;
                if (n < p.N) *(floatx4*)(ws + (size_t)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 64 + i * 16 + em;
        if (m >= p.M) continue;
        const half_t* rb = p.row_bias ? (const half_t*)p.row_bias + (size_t)(m / p.rows_per_group) * p.N : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + en;
            if (n >= p.N) continue;
            floatx4 v = acc[i][j];
            if (p.bias) {
                const half4 b = *(const half4*)((const half_t*)p.bias + n);
                v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
            }
            if (rb) {
                const half4 b = *(const half4*)(rb + n);
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
            } else {
                half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *(half4*)((half_t*)p.C + (size_t)m * p.ldc + n) = o;
            }
        }
    }
}

// sums the split-K slabs and applies the same epilogue (4 outputs per thread)
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const asd_gemm_args p, int splits) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total4 = (size_t)p.M * p.N / 4;
    if (q >= total4) return;
    const int m = (int)((q * 4) / p.N), n = (int)((q * 4) % p.N);
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const floatx4 t = *(const floatx4*)(p.workspace + ((size_t)s * p.M + m) * p.N + n);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    if (p.bias) {
        const half4 b = *(const half4*)((const half_t*)p.bias + n);
        v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
    }
    if (p.row_bias) {
        const half4 b = *(const half4*)((const half_t*)p.row_bias + (size_t)(m / p.rows_per_group) * p.N + n);
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
    } else {
        half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *(half4*)((half_t*)p.C + (size_t)m * p.ldc + n) = o;
    }
}

extern "C" {

int asd_gemm_f16(const asd_gemm_args* a, void* stream) {
    ASD_CHECK_ARG(a && a->A && a->W && a->C && a->zero_page, "null argument");
    ASD_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "empty problem");
    ASD_CHECK_ARG(a->K % 8 == 0, "K must be a multiple of 8");
    ASD_CHECK_ARG(a->N % 4 == 0 && a->ldc % 4 == 0, "N and ldc must be multiples of 4");
    ASD_CHECK_ARG(a->ldw % 8 == 0 && (a->conv || a->lda % 8 == 0), "leading dimensions must be multiples of 8 halfs (16 B)");
    if (a->conv) {
        ASD_CHECK_ARG(a->Cin % 8 == 0 && a->K == 9 * a->Cin, "conv: Cin must be a multiple of 8 and K = 9*Cin");
        ASD_CHECK_ARG(a->Hout > 0 && a->Wout > 0 && a->M % (a->Hout * a->Wout) == 0, "conv: M must be B*Hout*Wout");
    }
    ASD_CHECK_ARG(a->split_k >= 1 && (a->split_k == 1 || a->workspace), "split-K needs a workspace");
    const int bn = a->N % 128 == 0 ? 128 : 64;
    static int nst_env = -1;  // ASD_GEMM_NST: ring depth (tuning knob), default 2
    if (nst_env < 0) { const char* e = getenv("ASD_GEMM_NST"); nst_env = e ? atoi(e) : 2; }
    const int nst = nst_env == 3 ? 3 : 2;
    static int bm_env = -1;  // ASD_GEMM_BM: 128 | 256 (tuning knob)
    if (bm_env < 0) { const char* e = getenv("ASD_GEMM_BM"); bm_env = e ? atoi(e) : 0; }
    // 256-row tiles (8 waves: 5.3 MFMAs per LDS-DMA instruction instead of 4 / 2.7) when they still fill the chip
    int bm = (asd_div_up(a->M, 256) * asd_div_up(a->N, bn) >= 200) ? 256 : 128;
    if (bm_env == 128 || bm_env == 256) bm = bm_env;
    const int tiles = asd_div_up(a->M, bm) * asd_div_up(a->N, bn);
    const dim3 grid(tiles, 1, a->split_k), block(bm * 2);
    const size_t lds = (size_t)nst * (bm + bn) * 128;
    hipStream_t s = (hipStream_t)stream;
#define GEMM_LAUNCH(BM_, BN_, CONV_, NST_)                                                                               \
    do {                                                                                                                 \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)gemm_f16_kernel<BM_, BN_, CONV_, NST_>,                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, NST_ * (BM_ + BN_) * 128);             \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((gemm_f16_kernel<BM_, BN_, CONV_, NST_>), grid, block, lds, s, *a);                           \
    } while (0)
#define GEMM_DISPATCH_NST(BN_, CONV_)                                                            \
    do {                                                                                          \
        if (bm == 256) GEMM_LAUNCH(256, BN_, CONV_, 2);                                           \
        else if (nst == 2) GEMM_LAUNCH(128, BN_, CONV_, 2);                                       \
        else GEMM_LAUNCH(128, BN_, CONV_, 3);                                                     \
    } while (0)
    if (bn == 128) {
        if (a->conv) GEMM_DISPATCH_NST(128, true); else GEMM_DISPATCH_NST(128, false);
    } else {
        if (a->conv) GEMM_DISPATCH_NST(64, true); else GEMM_DISPATCH_NST(64, false);
    }
#undef GEMM_DISPATCH_NST
#undef GEMM_LAUNCH
    if (a->split_k > 1) {
        const size_t total4 = (size_t)a->M * a->N / 4;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(asd_div_up(total4, 256)), dim3(256), 0, s, *a, a->split_k);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
