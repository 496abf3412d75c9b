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
#include "asd_common.h"

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

#define BM 128
#define BK 32
#define ROW_BYTES (BK * 2)  // 64

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// issue one 16-row x 64-B slab: lane -> (row = lane>>2, physical chunk = lane&3); LDS destination is linear
__device__ __forceinline__ void load_slab(const char* src_row_chunk, char* lds_slab_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src_row_chunk, (LDS_AS void*)lds_slab_base, 16, 0, 0);
}

template <int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const asd_gemm_args p) {
    constexpr int TM = 4;            // 64 rows per wave in m
    constexpr int TN = BN / 32;      // BN/2 columns per wave in n
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A_BYTES + B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive blocks of one XCD (b % 8) walk down M for a fixed N panel, so the
    // weight panel stays in that XCD's L2
    const int tiles_m = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int m0 = (bid % tiles_m) * BM, n0 = (bid / tiles_m) * BN;
    const int kz = blockIdx.z;  // split-K slice
    const int k_steps_total = p.K / BK;
    const int k_per = (k_steps_total + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ks0 = kz * k_per, ks1 = min(k_steps_total, ks0 + k_per);

    // ---- per-lane source descriptors -----------------------------------------------------------
    const int lrow = lane >> 2;                              // row inside a 16-row slab
    const int pchunk = lane & 3;                             // physical 16-B chunk
    const int lchunk = pchunk ^ (((lrow >> 3) & 1) << 1);    // logical chunk loaded into that slot (swizzle)
    const char* zero = (const char*)p.zero_page;
    // A: 8 slabs of 16 rows; wave w loads slabs 2w, 2w+1
    const char* a_base[2];
    int a_y[2], a_x[2], a_valid[2];
    const char* a_img[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + (wave * 2 + i) * 16 + lrow;
        a_valid[i] = m < p.M;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, r = m - b * hw;
            a_y[i] = r / p.Wout;
            a_x[i] = r - a_y[i] * p.Wout;
            a_img[i] = (const char*)p.A + (size_t)b * p.Hin * p.Win * p.Cin * 2;
            a_base[i] = nullptr;
        } else {
            a_base[i] = (const char*)p.A + (size_t)m * p.lda * 2 + lchunk * 16;
        }
    }
    // W: BN/16 slabs; wave w loads slabs w*(BN/64) .. (BN=128: 2 slabs, BN=64: 1 slab)
    constexpr int WS = BN / 64;
    const char* w_base[WS];
    int w_valid[WS];
#pragma unroll
    for (int i = 0; i < WS; ++i) {
        const int n = n0 + (wave * WS + i) * 16 + lrow;
        w_valid[i] = n < p.N;
        w_base[i] = (const char*)p.W + (size_t)n * p.ldw * 2 + lchunk * 16;
    }

    auto issue = [&](int ks, int buf) {
        char* As = smem + buf * (A_BYTES + B_BYTES);
        char* Bs = As + A_BYTES;
        const int k0 = ks * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* src;
            if (CONV) {
                const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
                const int ky = tap / 3, kx = tap - ky * 3;
                int yi = a_y[i] * p.stride + ky - p.pad, xi = a_x[i] * p.stride + kx - p.pad;
                bool ok = a_valid[i];
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
                src = ok ? a_img[i] + ((size_t)(yi * p.Win + xi) * p.Cin + c0) * 2 + lchunk * 16 : zero;
            } else {
                src = a_valid[i] ? a_base[i] + (size_t)k0 * 2 : zero;
            }
            load_slab(src, As + (wave * 2 + i) * 16 * ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < WS; ++i) {
            const char* src = w_valid[i] ? w_base[i] + (size_t)k0 * 2 : zero;
            load_slab(src, Bs + (wave * WS + i) * 16 * ROW_BYTES);
        }
    };

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane -> row (lane&15), k-quarter (lane>>4), swizzled chunk
    const int frow = lane & 15, fq = lane >> 4;
    const int fchunk = fq ^ (((frow >> 3) & 1) << 1);
    const int frag_off = frow * ROW_BYTES + fchunk * 16;

    if (ks0 < ks1) {
        issue(ks0, 0);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the first tile has landed
        __syncthreads();
        for (int ks = ks0; ks < ks1; ++ks) {
            const int buf = (ks - ks0) & 1;
            if (ks + 1 < ks1) issue(ks + 1, buf ^ 1);
            const char* As = smem + buf * (A_BYTES + B_BYTES);
            const char* Bs = As + A_BYTES;
            half8 xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) xa[i] = *(const half8*)(As + (wm * 64 + i * 16) * ROW_BYTES + frag_off);
#pragma unroll
            for (int j = 0; j < TN; ++j) wb[j] = *(const half8*)(Bs + (wn * (BN / 2) + j * 16) * ROW_BYTES + frag_off);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            __syncthreads();  // (drains the in-flight LDS-DMA of the next tile: vmcnt(0) + barrier)
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
    ASD_CHECK_ARG(a->K % BK == 0, "K must be a multiple of 32");
    ASD_CHECK_ARG(a->N % 4 == 0 && a->ldc % 4 == 0, "N and ldc must be multiples of 4");
    ASD_CHECK_ARG(a->ldw % 8 == 0 && (a->conv || a->lda % 8 == 0), "leading dimensions must be multiples of 8 halfs (16 B)");
    if (a->conv) {
        ASD_CHECK_ARG(a->Cin % BK == 0 && a->K == 9 * a->Cin, "conv: Cin must be a multiple of 32 and K = 9*Cin");
        ASD_CHECK_ARG(a->Hout > 0 && a->Wout > 0 && a->M % (a->Hout * a->Wout) == 0, "conv: M must be B*Hout*Wout");
    }
    ASD_CHECK_ARG(a->split_k >= 1 && (a->split_k == 1 || a->workspace), "split-K needs a workspace");
    const int bn = a->N % 128 == 0 ? 128 : 64;
    const int tiles = asd_div_up(a->M, BM) * asd_div_up(a->N, bn);
    const dim3 grid(tiles, 1, a->split_k), block(256);
    const size_t lds = 2 * (BM + bn) * ROW_BYTES;
    hipStream_t s = (hipStream_t)stream;
    if (bn == 128) {
        if (a->conv) hipLaunchKernelGGL((gemm_f16_kernel<128, true>), grid, block, lds, s, *a);
        else hipLaunchKernelGGL((gemm_f16_kernel<128, false>), grid, block, lds, s, *a);
    } else {
        if (a->conv) hipLaunchKernelGGL((gemm_f16_kernel<64, true>), grid, block, lds, s, *a);
        else hipLaunchKernelGGL((gemm_f16_kernel<64, false>), grid, block, lds, s, *a);
    }
    if (a->split_k > 1) {
        const size_t total4 = (size_t)a->M * a->N / 4;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(asd_div_up(total4, 256)), block, 0, s, *a, a->split_k);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
