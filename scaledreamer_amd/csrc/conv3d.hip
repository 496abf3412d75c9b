// conv3d.hip — the StyleGAN-3D generator's 3x3x3 convolutions (custom/amortized/extern/stylegan_3dconv_modules.py:64-82 modulated_conv3d,
// :117-171 SynthesisLayer / SynthesisBlock) on the fp16 matrix pipe at fp32-class accuracy.
//
// The reference trains this generator in fp32 (F.conv3d, groups = batch).  gfx950 has no reduced-precision fp32 matrix path (no xf32;
// v_mfma_f32_*_f32 runs at the vector rate, 1/16 of fp16), so every fp32 operand is SPLIT into two fp16 planes
//     x * s = hi + lo,   hi = fp16(x * s),  lo = fp16(x * s - hi)        (s: a power of two that puts max|x| into [2^14, 2^15))
// and a product of two operands is formed from three fp16 MFMA products with fp32 accumulation,
//     a . b  =  a_hi b_hi + a_hi b_lo + a_lo b_hi      (+ a_lo b_lo ~ 2^-22 |a||b|, dropped),
// i.e. 22 significant bits per operand (fp32 has 24) at 1/3 of the fp16 MFMA rate instead of 1/16.  Products of fp16 values are exact in
// fp32, so the only roundings are the two splits and the fp32 accumulation the library path has as well.
//
//   forward / input gradient   conv3d_pp_kernel<TN>: the ping-pong LDS-window schedule of gemm_pp.hip (eight waves, two per SIMD, staggered by one
//                              barrier) carried to three dimensions: a block owns a 16 x 16 voxel patch of one depth slice x BN = 32 TN output
//                              channels; per (kd, 32-channel chunk) the 18 x 18 window of depth slice d + kd - 1 — BOTH planes — is brought into
//                              LDS once and the nine in-plane taps run against it; weight tiles (both planes) stream through a three-slot ring.
//                              Fragment reads per MFMA: 2 (TM + TN) per 3 TM TN (the K-concatenated form [hi | hi | lo] x [hi ; lo ; hi] needs 3 (TM + TN)).
//                              The input gradient is the same convolution with mirrored taps and transposed weights (asd_conv3d_pack_w).
//   weight gradient            dW[co][kd,ky,kx][ci] = sum_v dY[v][co] X[v + tap][ci]: a GEMM contracting over VOXELS.  Both tensors are transposed
//                              once into channel-major zero-padded planes [C][(D + 2)(H + 1)(W + 8)]; tap (kd, ky) of X and tap kx of dY are then
//                              the SAME rows read at an offset along K, so gemm_f16_kernel runs M = 9 Cin x N = 3 Cout x K = voxels with segmented rows
//                              (asd_gemm_args.a_seg_*; LDS-DMA loads take 2-byte aligned sources at full rate, tools/lds_dma_align_probe.hip), three
//                              launches (hi.hi, hi.lo, lo.hi) into the split-K slabs and one reduction that also restores the weight layout.
// Activations and gradients cross this file as fp32 channel-last volumes [N][D][H][W][C] (what the voxel sampler of amortized.hip reads).
#include <cstdlib>
#include <cstring>

#include "gemm_tile.h"

// ---- scales ---------------------------------------------------------------------------------------------------------------------
// amax_bits: bit pattern of max|x| (non-negative floats order like their bit patterns).  scale = 2^(14 - floor(log2 amax)).
__device__ __forceinline__ float split_scale(unsigned amax_bits) {
    if (amax_bits == 0u || amax_bits >= 0x7f800000u) return 1.f;       // all zero (or not finite: nothing to save)
    int se = 14 - ((int)((amax_bits >> 23) & 0xffu) - 127);
    se = se > 100 ? 100 : (se < -100 ? -100 : se);
    return __int_as_float((unsigned)(se + 127) << 23);
}
__device__ __forceinline__ void split2(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}

__device__ __forceinline__ float abs4max(const floatx4 v, float m) {
    return fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);
}
// the wave's max|.| into *out (bit pattern; callers zero *out first).  Tens of thousands of waves committing to ONE address serialise at its L2
// channel (~5 ns each: 81 us for any tensor past the grid cap, whatever its size) — and almost none of them raises the maximum: a wave reads
// the word first (an L2 load: monotone, so a stale value only costs a redundant atomic) and issues the atomic only if it would change it
__device__ __forceinline__ void wave_amax_commit(float m, unsigned* out) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        const unsigned b = __float_as_uint(m);
        if (b > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, b);
    }
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, unsigned* __restrict__ out) {
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight
        const floatx4 a = *(const floatx4*)(x + 4 * i), b = *(const floatx4*)(x + 4 * (i + stride));
        const floatx4 c = *(const floatx4*)(x + 4 * (i + 2 * stride)), d = *(const floatx4*)(x + 4 * (i + 3 * stride));
        m = abs4max(a, abs4max(b, abs4max(c, abs4max(d, m))));
    }
    for (; i < n4; i += stride) m = abs4max(*(const floatx4*)(x + 4 * i), m);
    // one atomic per BLOCK (and at most 1024 blocks): 7 000 waves finishing together all see the old word and queue their atomics on one L2
    // channel — 100 us for a 28 MB tensor (tools/absmax_time.py)
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b = __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
        if (b > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, b);
    }
}
static inline int c3_amax_grid(size_t n4) { size_t g = (n4 + 1023) / 1024; return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g)); }

// x fp32 [rows][C] -> hi / lo fp16 [rows][C] (channel-last volume: rows = voxels), 8 channels per thread
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, size_t n8, const unsigned* __restrict__ amax,
                                                         half_t* __restrict__ hi, half_t* __restrict__ lo) {
    const float s = split_scale(*amax);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const floatx4 a = *(const floatx4*)(x + 8 * i), b = *(const floatx4*)(x + 8 * i + 4);
        half8 h, l;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            half_t u, v;
            split2(a[r] * s, u, v); h[r] = u; l[r] = v;
            split2(b[r] * s, u, v); h[4 + r] = u; l[4 + r] = v;
        }
        *(half8*)(hi + 8 * i) = h;
        *(half8*)(lo + 8 * i) = l;
    }
}

// ---- channel-major padded planes for the weight gradient ---------------------------------------------------------------------------
// x fp32 [D][H][W][C] (one sample) -> hi / lo fp16 [C][ld] with voxel (d, h, w) at GUARD + ((d + 1) Hp + h) Wp + w, Hp = H + 1, Wp = W + 8:
// one zero row behind every H rows, eight zero columns behind every row, one zero plane in front of and behind the volume — each gap
// serves the -1 side of what follows it and the +1 side of what precedes it.  Block = one padded row (dp, hp) x 32 channels; the pad
// rows / planes are written as zeros here, so the buffer needs no memset (guards and the K tail are cleared by the caller).
__global__ __launch_bounds__(256) void split_cm_kernel(const float* __restrict__ x, int D, int H, int W, int C, const unsigned* __restrict__ amax,
                                                       half_t* __restrict__ hi, half_t* __restrict__ lo, size_t ld, int guard) {
    __shared__ half_t th[32][72], tl[32][72];      // [channel][w], 64-wide w tiles (+8: the store phase reads rows of 8 halfs, 16-B aligned)
    const int Hp = H + 1, Wp = W + 8;
    const int row = blockIdx.x, c0 = blockIdx.y * 32;
    const int dp = row / Hp, hp = row - dp * Hp;
    const bool interior = dp >= 1 && dp <= D && hp < H;
    const float s = split_scale(*amax);
    const size_t base = (size_t)guard + (size_t)row * Wp;
    const int tid = threadIdx.x;
    for (int w0 = 0; w0 < Wp; w0 += 64) {
        if (interior) {
            // load: thread -> (w = tid >> 2, 8 channels at (tid & 3) * 8)
            const int w = w0 + (tid >> 2), cc = (tid & 3) * 8;
            floatx4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
            if (w < W) {
                const float* src = x + (((size_t)(dp - 1) * H + hp) * W + w) * C + c0 + cc;
                a = *(const floatx4*)src; b = *(const floatx4*)(src + 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                half_t h, l;
                split2(a[r] * s, h, l); th[cc + r][tid >> 2] = h; tl[cc + r][tid >> 2] = l;
                split2(b[r] * s, h, l); th[cc + 4 + r][tid >> 2] = h; tl[cc + 4 + r][tid >> 2] = l;
            }
        }
        __syncthreads();
        // store: thread -> (channel = tid >> 3, 8 consecutive w at (tid & 7) * 8); Wp % 8 == 0 and guard % 8 == 0: 16-byte aligned
        const int c = tid >> 3, wq = (tid & 7) * 8;
        if (w0 + wq < Wp) {
            half8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = h;
            if (interior) { h = *(const half8*)&th[c][wq]; l = *(const half8*)&tl[c][wq]; }
            *(half8*)(hi + (size_t)(c0 + c) * ld + base + w0 + wq) = h;
            *(half8*)(lo + (size_t)(c0 + c) * ld + base + w0 + wq) = l;
        }
        __syncthreads();
    }
}

// ---- weights ------------------------------------------------------------------------------------------------------------------
// w fp32 [O][I][27] (torch [O, I, kd, kh, kw]) -> hi / lo fp16
//   transpose == 0 (forward):         [O][tap][I]
//   transpose == 1 (input gradient):  [I][26 - tap][O]   (mirrored taps, channels swapped)
__global__ __launch_bounds__(256) void pack_w_kernel(const float* __restrict__ w, int O, int I, int transpose, const unsigned* __restrict__ amax,
                                                     half_t* __restrict__ hi, half_t* __restrict__ lo) {
    const float s = split_scale(*amax);
    const size_t n = (size_t)O * I * 27;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // destination-ordered: i = (r * 27 + t) * Cn + c
        const int Cn = transpose ? O : I;
        const int c = (int)(i % Cn);
        const size_t q = i / Cn;
        const int t = (int)(q % 27), r = (int)(q / 27);
        const size_t src = transpose ? ((size_t)c * I + r) * 27 + (26 - t) : ((size_t)r * I + c) * 27 + t;
        half_t h, l;
        split2(w[src] * s, h, l);
        hi[i] = h; lo[i] = l;
    }
}

// ---- the convolution ----------------------------------------------------------------------------------------------------------
struct conv3d_kargs {
    const char* x_hi; const char* x_lo;     // fp16 [N * D][H][W][Cin]
    const char* w_hi; const char* w_lo;     // fp16 [Cout][27][Cin]
    float* y;                               // fp32 [N * D * H * W][ldc]
    int N, D, H, W, Cin, Cout, ldc;
    const unsigned* amax_x; const unsigned* amax_w;
    const float* bias;                      // [Cout] or null
    const float* noise;                     // [N * D * H * W] or null, times *noise_strength
    const float* noise_strength;
    int act;                                // 0: none; 1: clamp(lrelu(v, 0.2) * gain, +-clamp)
    float gain, clamp;
    unsigned* amax_out;                     // optional: max|y| (bit pattern, atomicMax) for the consumer's split
    int group_m, group_n;
};

template <int N>
__device__ __forceinline__ void c3_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void c3_vm_wait_n(int n) {      // wave-uniform n <= 7
    switch (n) {
        case 0: c3_vm_wait<0>(); break;
        case 1: c3_vm_wait<1>(); break;
        case 2: c3_vm_wait<2>(); break;
        case 3: c3_vm_wait<3>(); break;
        case 4: c3_vm_wait<4>(); break;
        case 5: c3_vm_wait<5>(); break;
        case 6: c3_vm_wait<6>(); break;
        default: c3_vm_wait<7>(); break;
    }
}
__device__ __forceinline__ void c3_mfma(floatx4& c, const half8& w, const half8& x) {       // in place (gemm_pp.hip: pp_mfma)
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
#define C3_PIN() __builtin_amdgcn_sched_barrier(0)
#define C3_BARRIER()                    \
    do {                                \
        C3_PIN();                       \
        __builtin_amdgcn_s_barrier();   \
        C3_PIN();                       \
    } while (0)

template <int TN>
__global__ __launch_bounds__(512) void conv3d_pp_kernel(const conv3d_kargs p) {
    constexpr int TM = 4, WM = 4, WN = 2, PH = WM * TM, BN = WN * TN * 16, PITCH = 24;
    constexpr int WLINES = PH + 2, WIN_BYTES = WLINES * PITCH * 64, W_BYTES = BN * 64;      // ONE plane of a window / of a weight tile
    constexpr int WSLABS = BN / 16, NWL = 2 * WSLABS / 8;            // weight slabs (16 rows x 64 B) per plane; loads per wave and K-step (both planes)
    constexpr int NPP = 3 * WLINES / 2, NPIECE = 2 * NPP;            // window pieces (1 KiB = 16 rows) per plane / per chunk
    constexpr int WROUNDS = (NPIECE + 7) / 8, PPT = (WROUNDS + 3) / 4;   // pieces per wave in each of taps 1..4
    static_assert(TN % 2 == 0 && NWL >= 1 && NWL + PPT <= 7, "the weight slabs of both planes are dealt to the eight waves; c3_vm_wait_n covers 0..7");
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [window 0: hi | lo][window 1: hi | lo][weights 0: hi | lo][1][2]
    char* const win = smem;
    char* const wring = smem + 4 * WIN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;          // waves w and w + 4 share a SIMD: one of each group
    const int grp = wn;
    const int H = p.H, Wd = p.W;
    const int tiles_x = Wd / 16, tiles_y = H / PH;
    const int tiles_m = p.N * p.D * tiles_y * tiles_x;
    const int tiles_n = p.Cout / BN;
    int item, tm, tn_;
    if (!asd_xcd_item(blockIdx.x, tiles_m * tiles_n, item)) return;
    asd_grouped_tile(item, tiles_m, tiles_n, p.group_m, p.group_n, tm, tn_);
    const int n0 = tn_ * BN;
    const int b = tm / (tiles_y * tiles_x), tr = tm - b * tiles_y * tiles_x;       // b = n * D + d: depth slice of the output patch
    const int y0 = (tr / tiles_x) * PH, x0 = (tr - (tr / tiles_x) * tiles_x) * 16;
    const int d = b % p.D;
    const int cps = p.Cin / 32;                                   // 32-channel chunks per depth tap
    const int kd_lo = d == 0 ? 1 : 0, kd_hi = d == p.D - 1 ? 1 : 2;   // depth taps inside the volume (the others multiply zeros: skipped)
    const int nch = (kd_hi - kd_lo + 1) * cps;                    // chunks of this block, (kd, cc) with cc fastest
    const size_t slice_bytes = (size_t)H * Wd * p.Cin * 2;
    const int ldw2 = 27 * p.Cin * 2;                              // bytes per weight row

    // ---- loaders: lane -> (row rho = lane >> 2 of a 16-row slab, physical chunk lane & 3) -------------------------------------------
    const int rho = lane >> 2, pch = lane & 3;
    unsigned woff[NWL];                                // byte offset of this lane's 16 B inside a weight plane for tap 0 / chunk 0
    int wdst[NWL];                                     // LDS byte offset of the slab inside a ring slot
    const char* wsrc[NWL];                             // plane the slab comes from
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int sidx = wave + 8 * j;                 // 0 .. 2 WSLABS - 1: hi slabs, then lo slabs
        const int plane = sidx / WSLABS, sl = sidx - plane * WSLABS;
        const int R = sl * 16 + rho;                   // row of the tile = output channel n0 + R
        const int q = pch ^ (((rho >> 2) & 1) << 1);
        woff[j] = (unsigned)(n0 + R) * (unsigned)ldw2 + q * 16;
        wdst[j] = plane * W_BYTES + sl * 1024;
        wsrc[j] = plane ? p.w_lo : p.w_hi;
    }
    // chunk c of this block -> (kd = kd_lo + c / cps, cc = c % cps)
    auto load_w = [&](int c, int t, int slot) __attribute__((always_inline)) {
        const int kd = kd_lo + c / cps, cc = c - (c / cps) * cps;
        const size_t koff = ((size_t)(kd * 9 + t) * p.Cin + (size_t)cc * 32) * 2;
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const char* base = wsrc[j] + koff;
            asm volatile("" : "+s"(base));
            unsigned o = woff[j];
            asm volatile("" : "+v"(o));
            load_slab(base + o, wring + slot * (2 * W_BYTES) + wdst[j]);
        }
    };
    // window piece pc (0 .. NPIECE - 1: plane pc / NPP) of chunk c: rows [16 k, 16 k + 16) of the 48-row pair of window lines.  Rows outside
    // the image (and the six padding rows of a line) are not loaded: their lanes are masked off and the rows were zeroed once in the prologue
    auto piece = [&](int pc, int c, char* wbuf, bool zero_pass) __attribute__((always_inline)) {
        const int plane = pc >= NPP ? 1 : 0, pq = pc - plane * NPP;
        const int pair = pq / 3, k = pq - pair * 3;
        int rho_o = rho;
        asm volatile("" : "+v"(rho_o));
        const int o = k * 16 + rho_o;
        const int second = o >= PITCH ? 1 : 0;
        const int wy = 2 * pair + second, col = o - PITCH * second;
        const int yi = y0 - 1 + wy, xi = x0 - 1 + col;
        const bool ok = col < 18 && (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)Wd;
        char* const dst = wbuf + plane * WIN_BYTES + (pair * 48 + k * 16) * 64;
        if (zero_pass) {
            if (!ok) {
                *(uint4*)(dst + lane * 16) = uint4{0u, 0u, 0u, 0u};
                *(uint4*)(dst + 2 * WIN_BYTES + lane * 16) = uint4{0u, 0u, 0u, 0u};
            }
            return;
        }
        const int kd = kd_lo + c / cps, cc = c - (c / cps) * cps;
        const int q = pch ^ (((col >> 2) & 1) << 1);
        const unsigned off = (unsigned)(yi * Wd + xi) * (unsigned)(p.Cin * 2) + q * 16;
        const char* base = (plane ? p.x_lo : p.x_hi) + (size_t)(b + kd - 1) * slice_bytes + (size_t)cc * 64;
        asm volatile("" : "+s"(base));
        if (ok) load_slab(base + off, dst);
    };

    // ---- fragment addressing: lane -> (row i = lane & 15 of a 16-row fragment, logical chunk fq = lane >> 4) --------------------------
    const int fi = lane & 15, fq = lane >> 4;
    int la[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = fi + kx;
        la[kx] = (wm * TM * PITCH + col) * 64 + ((fq ^ (((col >> 2) & 1) << 1)) << 4);
    }
    const int lb = (wn * TN * 16 + fi) * 64 + ((fq ^ (((fi >> 2) & 1) << 1)) << 4);

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    if (nch > 0) {
        // ---- prologue: the first window, weight tiles of K-steps 0 and 1 ------------------------------------------------------------
        for (int pc = wave; pc < NPIECE; pc += 8) {
            piece(pc, 0, win, true);
            piece(pc, 0, win, false);
        }
        load_w(0, 0, 0);
        load_w(0, 1, 1);
        c3_vm_wait<0>();
        C3_BARRIER();
        if (grp == 1) C3_BARRIER();      // the second group runs one barrier behind

#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const bool last = c + 1 >= nch;
            const int par = c & 1;
            const char* const wcur = win + par * (2 * WIN_BYTES);
            char* const wnext = win + (par ^ 1) * (2 * WIN_BYTES);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const char* const Ab = wcur + la[kx] + ky * PITCH * 64;
                const char* const Ws = wring + (t % 3) * (2 * W_BYTES) + lb;
                const bool iss = !(last && t >= 7);    // K-step s + 2 exists
                half8 Bh[TN], Bl[TN], Ah[TM], Al[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j) { Bh[j] = *(const half8*)(Ws + j * 1024); Bl[j] = *(const half8*)(Ws + W_BYTES + j * 1024); }
#pragma unroll
                for (int a = 0; a < TM; ++a) { Ah[a] = *(const half8*)(Ab + a * PITCH * 64); Al[a] = *(const half8*)(Ab + WIN_BYTES + a * PITCH * 64); }
                C3_PIN();
                int nwin = 0;
                if (!last && t >= 1 && t <= 4) {
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int pc = ((t - 1) * PPT + k) * 8 + wave;
                        if (pc < NPIECE) { piece(pc, c + 1, wnext, false); ++nwin; }
                    }
                }
                if (iss) {
                    if (t + 2 < 9) load_w(c, t + 2, (t + 2) % 3);
                    else load_w(c + 1, t + 2 - 9, (t + 2) % 3);
                    c3_vm_wait_n(NWL + nwin);          // weights of K-step s + 1 (and every window piece issued before this tap) have landed
                } else if (t == 7) {
                    c3_vm_wait<0>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the partner may refill the slot
                C3_BARRIER();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int j = 0; j < TN; ++j) c3_mfma(acc[a][j], Bl[j], Ah[a]);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int j = 0; j < TN; ++j) c3_mfma(acc[a][j], Bh[j], Al[a]);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int j = 0; j < TN; ++j) c3_mfma(acc[a][j], Bh[j], Ah[a]);
                __builtin_amdgcn_s_setprio(0);
                C3_BARRIER();
            }
        }
        if (grp == 0) C3_BARRIER();      // the first group waits for the second one's last segment
    }

    // ---- epilogue: acc[i][j][r] = Y[voxel (b, y0 + wm*TM + i, x0 + (lane&15))][n0 + wn*TN*16 + j*16 + (lane>>4)*4 + r] ------------------
    const float inv = 1.f / (split_scale(*p.amax_x) * split_scale(*p.amax_w));
    const float ns = p.noise ? *p.noise_strength : 0.f;
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const size_t m = ((size_t)b * H + y0 + wm * TM + i) * Wd + x0 + fi;
        const float nz = p.noise ? p.noise[m] * ns : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 16 + j * 16 + fq * 4;
            floatx4 v = acc[i][j];
            floatx4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *(const floatx4*)(p.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = v[r] * inv + nz + bb[r];
                if (p.act == 1) {
                    u = (u >= 0.f ? u : 0.2f * u) * p.gain;
                    u = fminf(fmaxf(u, -p.clamp), p.clamp);
                }
                v[r] = u;
            }
            am = abs4max(v, am);
            *(floatx4*)(p.y + m * p.ldc + n) = v;
        }
    }
    if (p.amax_out) wave_amax_commit(am, p.amax_out);
}

// ---- weight gradient: reduction of the split-K slabs of the two launches -------------------------------------------------------------
// s1 fp32 [n1][M = 9 Cin][6 Cout]: X_hi against [dY_hi | dY_lo] (column (plane, kx, co)); s2 fp32 [n2][M][3 Cout]: X_lo against dY_hi.
// row (t9 = kd * 3 + ky, ci)  ->  dw fp32 [Cout][Cin][27], times 1 / (s_x s_dy)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ s1, int n1, const float* __restrict__ s2, int n2, int Cin, int Cout,
                                                           const unsigned* __restrict__ amax_x, const unsigned* __restrict__ amax_dy, float* __restrict__ dw) {
    const int N = 3 * Cout;
    const size_t MN = (size_t)9 * Cin * N;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    const int n = (int)(i % N), m = (int)(i / N);
    float v = 0.f;
    for (int s = 0; s < n1; ++s) {
        const float* row = s1 + ((size_t)s * 9 * Cin + m) * (2 * N);
        v += row[n] + row[N + n];
    }
    for (int s = 0; s < n2; ++s) v += s2[(size_t)s * MN + i];
    const int kx = n / Cout, co = n - kx * Cout, t9 = m / Cin, ci = m - t9 * Cin;
    v *= 1.f / (split_scale(*amax_x) * split_scale(*amax_dy));
    dw[((size_t)co * Cin + ci) * 27 + t9 * 3 + kx] = v;
}

// ---- layer glue on channel-last fp32 volumes -----------------------------------------------------------------------------------------
// gradient through the layer's activation: y = clamp(lrelu(z) * gain, +-clamp), z = conv + noise * ns + bias.  dz = dy * act'(y), read off
// the OUTPUT (sign(y) = sign(z); |y| == clamp where the clamp cut).  Also leaves d_bias[c] = sum_rows dz (atomics of block partials) and
// d_noise[row] = sum_c dz (the gradient of noise * ns w.r.t. the per-voxel term; the caller contracts it with the noise it drew).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ sub,
                                                      const unsigned char* __restrict__ act_mask, size_t rows, int C,
                                                      float gain, float clamp, float* __restrict__ dz, float* __restrict__ d_bias, float* __restrict__ d_rowsum,
                                                      unsigned* __restrict__ amax_out) {
    extern __shared__ float cb[];                     // [C] block partial of d_bias
    for (int c = threadIdx.x; c < C; c += 256) cb[c] = 0.f;
    __syncthreads();
    const int tpr = C / 4;                            // threads per row; 256 % tpr == 0 (host check): a thread keeps its 4 channels over all its rows
    const int rpb = 256 / tpr;                        // rows per block pass
    const int c = (threadIdx.x % tpr) * 4, rr = threadIdx.x / tpr;
    floatx4 bs = {0.f, 0.f, 0.f, 0.f};
    float am = 0.f;
    for (size_t r0 = (size_t)blockIdx.x * rpb; r0 < rows; r0 += (size_t)gridDim.x * rpb) {
        const size_t r = r0 + rr;
        float rs = 0.f;
        if (r < rows) {
            const floatx4 g = *(const floatx4*)(dy + r * C + c);
            floatx4 o = {0.f, 0.f, 0.f, 0.f};
            unsigned mk = 0;
            if (act_mask) {   // the forward pass left the branch of every element (y = act(.) + sub: (y - sub) is not exact in fp32)
                mk = act_mask[(r * C + c) >> 2];
            } else {
                o = *(const floatx4*)(y + r * C + c);
                if (sub) {    // y = act(.) + sub: the activation's own output, up to the rounding of the sum
                    const floatx4 sb = *(const floatx4*)(sub + r * C + c);
                    o[0] -= sb[0]; o[1] -= sb[1]; o[2] -= sb[2]; o[3] -= sb[3];
                }
            }
            floatx4 z;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float sl;
                if (act_mask) sl = ((mk >> (2 * k + 1)) & 1u) ? 0.f : (((mk >> (2 * k)) & 1u) ? gain : 0.2f * gain);
                else sl = fabsf(o[k]) >= clamp ? 0.f : (o[k] >= 0.f ? gain : 0.2f * gain);
                z[k] = g[k] * sl;
                rs += z[k];
                bs[k] += z[k];
            }
            am = abs4max(z, am);
            *(floatx4*)(dz + r * C + c) = z;
        }
        if (d_rowsum) {
            if (tpr <= 64) {          // the tpr lanes of a row sit in one wave (tpr is a power of two)
                for (int o2 = tpr >> 1; o2 >= 1; o2 >>= 1) rs += __shfl_xor(rs, o2, 64);
                if ((threadIdx.x % tpr) == 0 && r < rows) d_rowsum[r] = rs;
            } else if (r < rows) {
                atomicAdd(&d_rowsum[r], rs);      // (zeroed by the caller)
            }
        }
    }
    if (amax_out) wave_amax_commit(am, amax_out);
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(&cb[c + k], bs[k]);
    __syncthreads();
    if (d_bias)
        for (int cc = threadIdx.x; cc < C; cc += 256) atomicAdd(&d_bias[cc], cb[cc]);
}

// y = act(x + noise * ns + bias) on a channel-last volume (the layers whose activation follows an upsampling: upsample_fwd_kernel fuses the same)
__device__ __forceinline__ float layer_act(float u, int act, float gain, float clamp) {
    if (act == 1) {
        u = (u >= 0.f ? u : 0.2f * u) * gain;
        u = fminf(fmaxf(u, -clamp), clamp);
    }
    return u;
}

// trilinear 2x upsampling with align_corners = True (SmoothUpsample, stylegan_3dconv_modules.py:56-62) of x [N][r][r][r][C] -> [N][2r][2r][2r][C],
// fused with the layer tail y = act(up + noise * ns + bias) and / or an accumulation into y (the skip volume: img = up(img) + torgb(x)).
// source coordinate of output index a: a * (r - 1) / (2 r - 1)
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, int N, int r, int C, const float* __restrict__ noise,
                                                           const float* __restrict__ noise_strength, const float* __restrict__ bias, int act, float gain,
                                                           float clamp, const float* __restrict__ add, float* __restrict__ y, unsigned* __restrict__ amax_out,
                                                           unsigned char* __restrict__ act_mask) {
    const int R = 2 * r, c4 = C / 4;
    const size_t total = (size_t)N * R * R * R * c4;
    const float sc = (float)(r - 1) / (float)(R - 1);
    const float ns = noise ? *noise_strength : 0.f;
    float am = 0.f;
    // R and C / 4 are powers of two on every level of the generator: index arithmetic in shifts (the general form is 64-bit div / mod
    // five times per output quad — more instructions than the eight gathers)
    const bool pow2 = (R & (R - 1)) == 0 && (c4 & (c4 - 1)) == 0;
    const int lr = 31 - __builtin_clz(R), lc = 31 - __builtin_clz(c4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int c, ox, oy, oz, n;
        if (pow2) {
            c = (int)(i & (size_t)(c4 - 1)) * 4;
            const size_t v = i >> lc;
            ox = (int)(v & (size_t)(R - 1)); oy = (int)((v >> lr) & (size_t)(R - 1)); oz = (int)((v >> (2 * lr)) & (size_t)(R - 1));
            n = (int)(v >> (3 * lr));
        } else {
            c = (int)(i % c4) * 4;
            size_t v = i / c4;
            ox = (int)(v % R); v /= R;
            oy = (int)(v % R); v /= R;
            oz = (int)(v % R);
            n = (int)(v / R);
        }
        const float fx = ox * sc, fy = oy * sc, fz = oz * sc;
        const int x0 = min((int)fx, r - 1), y0 = min((int)fy, r - 1), z0 = min((int)fz, r - 1);
        const int x1 = min(x0 + 1, r - 1), y1 = min(y0 + 1, r - 1), z1 = min(z0 + 1, r - 1);
        const float wx = fx - x0, wy = fy - y0, wz = fz - z0;
        const float* xb = x + (size_t)n * r * r * r * C + c;
        floatx4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zi = (k & 4) ? z1 : z0, yi = (k & 2) ? y1 : y0, xi = (k & 1) ? x1 : x0;
            const float wt = ((k & 4) ? wz : 1.f - wz) * ((k & 2) ? wy : 1.f - wy) * ((k & 1) ? wx : 1.f - wx);
            const floatx4 t = *(const floatx4*)(xb + (((size_t)zi * r + yi) * r + xi) * C);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = fmaf(wt, t[q], o[q]);
        }
        const size_t row = pow2 ? i >> lc : i / c4;
        const float nz = noise ? noise[row] * ns : 0.f;
        floatx4 bb = {0.f, 0.f, 0.f, 0.f};
        if (bias) bb = *(const floatx4*)(bias + c);
        floatx4 ad = {0.f, 0.f, 0.f, 0.f};
        if (add) ad = *(const floatx4*)(add + row * C + c);
        unsigned mk = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float u = o[q] + nz + bb[q];
            // two bits per element for the backward pass: slope 1 (u >= 0) | clamped (the gradient of torch.clamp is zero outside [-c, c])
            mk |= (u >= 0.f ? 1u : 0u) << (2 * q);
            mk |= (fabsf((u >= 0.f ? u : 0.2f * u) * gain) > clamp ? 1u : 0u) << (2 * q + 1);
            o[q] = layer_act(u, act, gain, clamp) + ad[q];
        }
        if (act_mask) act_mask[i] = (unsigned char)mk;
        am = abs4max(o, am);
        *(floatx4*)(y + row * C + c) = o;
    }
    if (amax_out) wave_amax_commit(am, amax_out);
}

// transpose of the 1-D interpolation along one axis: dx [outer][r][inner] from dy [outer][2 r][inner] (three passes undo the upsampling).
// output a reads inputs i0(a), i0(a) + 1 with weights 1 - f(a), f(a); input i therefore collects the outputs around 2 i (at most five).
__global__ __launch_bounds__(256) void upsample1d_bwd_kernel(const float* __restrict__ dy, size_t outer, int r, size_t inner4, float* __restrict__ dx) {
    const int R = 2 * r;
    const float sc = (float)(r - 1) / (float)(R - 1);
    const size_t total = outer * r * inner4;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t in = t % inner4;
        const size_t q = t / inner4;
        const int i = (int)(q % r);
        const size_t o = q / r;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
        const int a_lo = max(0, 2 * i - 3), a_hi = min(R - 1, 2 * i + 3);
        for (int a = a_lo; a <= a_hi; ++a) {
            const float f = a * sc;
            const int i0 = min((int)f, r - 1), i1 = min(i0 + 1, r - 1);
            const float w1 = f - i0;
            float wt = 0.f;
            if (i0 == i) wt += 1.f - w1;
            if (i1 == i) wt += w1;
            if (wt != 0.f) {
                const floatx4 g = *(const floatx4*)(dy + ((o * R + a) * inner4 + in) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(wt, g[k], acc[k]);
            }
        }
        *(floatx4*)(dx + ((o * r + i) * inner4 + in) * 4) = acc;
    }
}


// ---- toRGB: the 1x1x1 modulated convolution onto the 32-channel skip volume (ToRGBLayer, stylegan_3dconv_modules.py:283-296) ---------------
// y[v][o] = sum_i x[v][i] w[o][i] + b[o] on channel-last rows, exact fp32 (v_fmac chains; HBM-bound: K = Cin <= 512 against 4 (Cin + 32) bytes per
// voxel).  Block = 64 voxels x 32 outputs: the x tile goes through LDS (coalesced 16-byte loads, conflict-free 16-byte reads at a pitch of
// 132 floats), wave q owns outputs 8 q .. 8 q + 7 whose weights are wave-uniform (scalar loads), lane = voxel.
#define TORGB_O 32
__global__ __launch_bounds__(256) void torgb_fwd_kernel(const float* __restrict__ x, size_t rows, int Cin, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ add, float* __restrict__ y, unsigned* __restrict__ amax_out) {
    __shared__ __attribute__((aligned(16))) float xs[64][132];
    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t v0 = (size_t)blockIdx.x * 64;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = bias ? bias[8 * q + k] : 0.f;
    for (int k0 = 0; k0 < Cin; k0 += 128) {
        const int kw = min(128, Cin - k0);                   // Cin % 32 == 0
        __syncthreads();
        for (int t = threadIdx.x; t < 64 * (kw / 4); t += 256) {
            const int r = t / (kw / 4), c = (t - r * (kw / 4)) * 4;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (v0 + r < rows) v = *(const floatx4*)(x + (v0 + r) * Cin + k0 + c);
            *(floatx4*)&xs[r][c] = v;
        }
        __syncthreads();
        for (int c = 0; c < kw; c += 4) {
            const floatx4 xv = *(const floatx4*)&xs[lane][c];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* wr = w + (size_t)(8 * q + k) * Cin + k0 + c;        // wave-uniform
                acc[k] = fmaf(xv[0], wr[0], acc[k]); acc[k] = fmaf(xv[1], wr[1], acc[k]);
                acc[k] = fmaf(xv[2], wr[2], acc[k]); acc[k] = fmaf(xv[3], wr[3], acc[k]);
            }
        }
    }
    float am = 0.f;
    if (v0 + lane < rows) {
        float* dst = y + (v0 + lane) * TORGB_O + 8 * q;
        floatx4 a = {acc[0], acc[1], acc[2], acc[3]}, b = {acc[4], acc[5], acc[6], acc[7]};
        if (add) {
            const floatx4 p0 = *(const floatx4*)(add + (v0 + lane) * TORGB_O + 8 * q), p1 = *(const floatx4*)(add + (v0 + lane) * TORGB_O + 8 * q + 4);
            a[0] += p0[0]; a[1] += p0[1]; a[2] += p0[2]; a[3] += p0[3]; b[0] += p1[0]; b[1] += p1[1]; b[2] += p1[2]; b[3] += p1[3];
        }
        am = abs4max(a, abs4max(b, 0.f));
        *(floatx4*)dst = a; *(floatx4*)(dst + 4) = b;
    }
    if (amax_out) wave_amax_commit(am, amax_out);
}
// dx[v][i] (+)= sum_o dy[v][o] w[o][i]: lane = voxel, wave q owns 16 input channels per pass; dy row (32 floats) through LDS
__global__ __launch_bounds__(256) void torgb_dgrad_kernel(const float* __restrict__ dy, size_t rows, int Cin, const float* __restrict__ w, const float* __restrict__ dx_add,
                                                          float* __restrict__ dx) {
    __shared__ __attribute__((aligned(16))) float gs[64][36];
    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t v0 = (size_t)blockIdx.x * 64;
    for (int t = threadIdx.x; t < 64 * 8; t += 256) {
        const int r = t >> 3, c = (t & 7) * 4;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
        if (v0 + r < rows) v = *(const floatx4*)(dy + (v0 + r) * TORGB_O + c);
        *(floatx4*)&gs[r][c] = v;
    }
    __syncthreads();
    for (int i0 = 16 * q; i0 < Cin; i0 += 64) {
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int o = 0; o < TORGB_O; o += 4) {
            const floatx4 g = *(const floatx4*)&gs[lane][o];
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) {
                const float* wr = w + (size_t)(o + oo) * Cin + i0;                  // wave-uniform
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(g[oo], wr[k], acc[k]);
            }
        }
        if (v0 + lane < rows) {
            float* dst = dx + (v0 + lane) * Cin + i0;
            const float* ad = dx_add ? dx_add + (v0 + lane) * Cin + i0 : nullptr;
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                floatx4 o4 = {acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
                if (ad) { const floatx4 p = *(const floatx4*)(ad + k); o4[0] += p[0]; o4[1] += p[1]; o4[2] += p[2]; o4[3] += p[3]; }
                *(floatx4*)(dst + k) = o4;
            }
        }
    }
}
// dw[o][i] += sum_v dy[v][o] x[v][i] over this block's voxels (atomics of the block totals; dw zeroed by the caller); d_bias[o] += sum_v dy[v][o].
// lane = input channel i0 + lane (coalesced reads of x), every thread accumulates ALL 32 outputs (the dy row is wave-uniform: scalar loads,
// 32 FMAs per loaded x), the block's four waves take every fourth row and are summed through LDS
__global__ __launch_bounds__(256) void torgb_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, size_t rows, int Cin, size_t rows_per_block,
                                                          float* __restrict__ dw, float* __restrict__ d_bias) {
    __shared__ float red[4][TORGB_O][64];
    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float sb = 0.f;                                   // lanes 0..31: column sum of dy over this wave's rows (first channel pass only)
    for (int i0 = 0; i0 < Cin; i0 += 64) {
        const int i = i0 + lane;
        float acc[TORGB_O];
#pragma unroll
        for (int k = 0; k < TORGB_O; ++k) acc[k] = 0.f;
        if (i < Cin) {
#pragma unroll 2
            for (size_t r = r0 + q; r < r1; r += 4) {
                const float xv = x[r * Cin + i];
                const float* g = dy + r * TORGB_O;                                   // wave-uniform
                if (i0 == 0 && lane < TORGB_O) sb += g[lane];
#pragma unroll
                for (int k = 0; k < TORGB_O; ++k) acc[k] = fmaf(xv, g[k], acc[k]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TORGB_O; ++k) red[q][k][lane] = acc[k];
        __syncthreads();
        // 32 x 64 sums of four waves: thread t -> outputs t >> 6 + 4 j, channel t & 63
        for (int k = q; k < TORGB_O; k += 4) {
            const float t = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
            if (i < Cin) atomicAdd(dw + (size_t)k * Cin + i, t);
        }
    }
    if (d_bias && lane < TORGB_O) atomicAdd(d_bias + lane, sb);
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// ---- the per-sample weights of a modulated convolution (stylegan_3dconv_modules.py:64-82 modulated_conv3d) ------------------------------
// wm[n][co][ci][k] = W[co][ci][k] s[n][ci] g d[n][co],  d = rsqrt(sum_{ci,k} (W s g)^2 + 1e-8) with demodulation, 1 without (toRGB: g is its
// weight_gain).  As tensor ops that is six passes over N Cout Cin K floats forward and a dozen backward (28 MB each at 512 -> 512 channels);
// here one launch forward, two backward.  A block owns one output channel; a thread owns whole (ci) rows of K taps, so the style gradient of
// a row is a register sum.
__device__ __forceinline__ float c3_block_sum(float v, float* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void modw_fwd_kernel(const float* __restrict__ W, const float* __restrict__ s, int Cin, int K, float gain, int demod,
                                                       float* __restrict__ wm, float* __restrict__ dcoef) {
    __shared__ float red[4];
    const int co = blockIdx.x, n = blockIdx.y, Cout = gridDim.x;
    const float* w = W + (size_t)co * Cin * K;
    const float* sn = s + (size_t)n * Cin;
    float* o = wm + ((size_t)n * Cout + co) * Cin * K;
    const int E = Cin * K;
    float d = 1.f;
    if (demod) {
        float ss = 0.f;
        for (int e = threadIdx.x; e < E; e += 256) { const float u = w[e] * (sn[e / K] * gain); ss = fmaf(u, u, ss); }
        d = rsqrtf(c3_block_sum(ss, red) + 1e-8f);
    }
    for (int e = threadIdx.x; e < E; e += 256) o[e] = w[e] * (sn[e / K] * gain) * d;
    if (threadIdx.x == 0) dcoef[(size_t)n * Cout + co] = d;
}

// with u = W s g and wm = u d:  d_u = d (d_wm - wm sum(d_wm wm)) (demodulated) | d_wm (not);  dW = sum_n d_u s g;  ds[n][ci] = g sum_{co,k} d_u W.
// A block owns one output channel and walks its Cin K elements in chunks of 256 rows (ci) x K taps with coalesced accesses: thread t holds
// elements t + 256 j of the chunk (its dW sums over n stay in registers), the products d_u W go through LDS where thread r sums row r
// (stride K = 27 words: conflict-free) and adds it to ds[n][ci] with one atomic — Cout adds per address over the launch.
template <int K>
__global__ __launch_bounds__(256) void modw_bwd_kernel(const float* __restrict__ d_wm, const float* __restrict__ wm, const float* __restrict__ W,
                                                       const float* __restrict__ s, const float* __restrict__ dcoef, int N, int Cin, float gain,
                                                       int demod, float* __restrict__ dW, float* __restrict__ ds) {
    __shared__ float red[4];
    __shared__ float t_sh[8];
    __shared__ float prod[256 * K];
    const int co = blockIdx.x, Cout = gridDim.x;
    const int E = Cin * K;
    if (demod) {
        for (int n = 0; n < N; ++n) {
            const float* g = d_wm + ((size_t)n * Cout + co) * E;
            const float* m = wm + ((size_t)n * Cout + co) * E;
            float a = 0.f;
            for (int e = threadIdx.x; e < E; e += 256) a = fmaf(g[e], m[e], a);
            a = c3_block_sum(a, red);
            if (threadIdx.x == 0) t_sh[n] = a;
        }
        __syncthreads();
    }
    const float* w = W + (size_t)co * E;
    for (int r0 = 0; r0 < Cin; r0 += 256) {
        const int e0 = r0 * K, ne = min(256, Cin - r0) * K;
        float acc[K], wv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int el = threadIdx.x + 256 * j;
            acc[j] = 0.f;
            wv[j] = el < ne ? w[e0 + el] : 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const size_t base = ((size_t)n * Cout + co) * E + e0;
            const float dc = dcoef[(size_t)n * Cout + co], tn = demod ? t_sh[n] : 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int el = threadIdx.x + 256 * j;
                float du = 0.f;
                if (el < ne) {
                    du = d_wm[base + el];
                    if (demod) du = dc * (du - wm[base + el] * tn);
                    acc[j] = fmaf(du, s[(size_t)n * Cin + r0 + el / K] * gain, acc[j]);
                }
                prod[el] = du * wv[j];
            }
            __syncthreads();
            if (r0 + (int)threadIdx.x < Cin) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) a += prod[threadIdx.x * K + k];
                atomicAdd(&ds[(size_t)n * Cin + r0 + threadIdx.x], a * gain);
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int el = threadIdx.x + 256 * j;
            if (el < ne) dW[(size_t)co * E + e0 + el] = acc[j];
        }
    }
}

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int c3_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

struct c3_cm_layout { int Hp, Wp; size_t Vp, ld, guard; };
static c3_cm_layout c3_cm(int D, int H, int W) {
    c3_cm_layout l;
    l.Hp = H + 1; l.Wp = W + 8;
    l.Vp = (size_t)(D + 2) * l.Hp * l.Wp;
    l.ld = (l.Vp + 63) / 64 * 64;
    l.guard = ((size_t)l.Hp * l.Wp + l.Wp + 64 + 7) / 8 * 8;
    return l;
}

// split-K of the weight-gradient GEMM (M = 9 Cin, N = 3 Cout, 128-row tiles): about two blocks per CU
static int c3_wgrad_split(const asd_conv3d_desc* d, size_t ld, int planes) {
    const int M = 9 * d->Cin, N = 3 * d->Cout * planes;
    const int bn = N % 128 == 0 ? 128 : 64, tiles = asd_div_up(M, 128) * (N / bn);
    int split = 512 / tiles;
    split = split < 2 ? 2 : (split > ASD_CONV3D_WGRAD_MAX_SPLIT ? ASD_CONV3D_WGRAD_MAX_SPLIT : split);
    const int ksteps = (int)(ld / 64);
    if (split > ksteps) split = ksteps < 2 ? 2 : ksteps;
    return split;
}

extern "C" {

int64_t asd_conv3d_workspace_bytes(const asd_conv3d_desc* d, int32_t pass) {
    if (!d) return 0;
    const size_t vox = (size_t)d->D * d->H * d->W;
    if (pass == 2) {      // weight gradient: four channel-major planes + slabs + scalars
        const c3_cm_layout l = c3_cm(d->D, d->H, d->W);
        const size_t plane_x = ((size_t)d->Cin * l.ld + 2 * l.guard) * 2, plane_y = ((size_t)d->Cout * l.ld + 2 * l.guard) * 2;
        const size_t slabs = ((size_t)2 * c3_wgrad_split(d, l.ld, 2) + c3_wgrad_split(d, l.ld, 1)) * 9 * d->Cin * 3 * d->Cout * 4;
        return (int64_t)(256 + 2 * al256(plane_x) + 2 * al256(plane_y) + al256(slabs));
    }
    const int cin = pass == 1 ? d->Cout : d->Cin, cout = pass == 1 ? d->Cin : d->Cout;
    const size_t xs = vox * cin * 2, ws = (size_t)cout * 27 * cin * 2;     // one sample at a time
    return (int64_t)(256 + 2 * al256(xs) + 2 * al256(ws));
}

static int c3_check(const asd_conv3d_desc* d) {
    ASD_CHECK_ARG(d && d->N > 0 && d->D > 0 && d->H > 0 && d->W > 0, "empty volume");
    ASD_CHECK_ARG(d->H % 16 == 0 && d->W % 16 == 0, "H and W must be multiples of 16 (pad the volume)");
    ASD_CHECK_ARG(d->Cin % 64 == 0 && d->Cout % 64 == 0, "Cin and Cout must be multiples of 64");
    ASD_CHECK_ARG((size_t)d->H * d->W * (d->Cin > d->Cout ? d->Cin : d->Cout) * 2 < ((size_t)1 << 32), "a depth slice is addressed with 32-bit byte offsets");
    return ASD_OK;
}

// one sample of the forward form: y[D,H,W,cout] = conv(x[D,H,W,cin], w[cout][cin][27] (transpose: w is [cin][cout][27], mirrored))
static int c3_run(int D, int H, int W, int cin, int cout, const float* x, const unsigned* amax_x, const float* w, int transpose_w, float* y,
                  const asd_conv3d_epilogue* ep, size_t noise_off, char* ws, hipStream_t s) {
    const size_t vox = (size_t)D * H * W;
    unsigned* amax = (unsigned*)ws;                                   // [0] x, [1] w
    half_t* xh = (half_t*)(ws + 256);
    half_t* xl = (half_t*)((char*)xh + al256(vox * cin * 2));
    half_t* wh = (half_t*)((char*)xl + al256(vox * cin * 2));
    half_t* wl = (half_t*)((char*)wh + al256((size_t)cout * 27 * cin * 2));
    (void)hipMemsetAsync(amax, 0, 8, s);
    if (!amax_x) {
        hipLaunchKernelGGL(absmax_kernel, dim3(c3_amax_grid(vox * cin / 4)), dim3(256), 0, s, x, vox * cin / 4, amax);
        amax_x = amax;
    }
    hipLaunchKernelGGL(absmax_kernel, dim3(c3_amax_grid((size_t)cout * cin * 27 / 4)), dim3(256), 0, s, w, (size_t)cout * cin * 27 / 4, amax + 1);
    hipLaunchKernelGGL(split_rows_kernel, dim3(c3_grid(vox * cin / 8)), dim3(256), 0, s, x, vox * cin / 8, amax_x, xh, xl);
    hipLaunchKernelGGL(pack_w_kernel, dim3(c3_grid((size_t)cout * cin * 27)), dim3(256), 0, s, w, transpose_w ? cin : cout, transpose_w ? cout : cin,
                       transpose_w, amax + 1, wh, wl);
    conv3d_kargs k;
    k.x_hi = (const char*)xh; k.x_lo = (const char*)xl; k.w_hi = (const char*)wh; k.w_lo = (const char*)wl;
    k.y = y; k.N = 1; k.D = D; k.H = H; k.W = W; k.Cin = cin; k.Cout = cout; k.ldc = cout;
    k.amax_x = amax_x; k.amax_w = amax + 1;
    k.amax_out = ep ? ep->amax_out : nullptr;
    k.bias = ep ? ep->bias : nullptr;
    k.noise = ep && ep->noise ? ep->noise + noise_off : nullptr;
    k.noise_strength = ep ? ep->noise_strength : nullptr;
    k.act = ep ? ep->act : 0; k.gain = ep ? ep->gain : 1.f; k.clamp = ep ? ep->clamp : 0.f;
    const int tn = cout % 128 == 0 ? 4 : 2, bn = 32 * tn;
    const int tiles_m = D * (H / 16) * (W / 16), tiles_n = cout / bn;
    // super-tile of the block order: one XCD's 32 concurrent blocks walk patches of one slice for all channel tiles
    k.group_n = tiles_n > 4 ? 4 : tiles_n;
    k.group_m = 32 / k.group_n;
    const int blocks = 8 * asd_div_up(tiles_m * tiles_n, 8);
    const size_t lds = (size_t)4 * 18 * 24 * 64 + (size_t)6 * bn * 64;
    if (tn == 4) {
        static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
        if (!attr) { (void)hipFuncSetAttribute((const void*)conv3d_pp_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        ASD_PROBE_START(s);
        hipLaunchKernelGGL((conv3d_pp_kernel<4>), dim3(blocks), dim3(512), lds, s, k);
        ASD_PROBE_STOP(s);
    } else {
        static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
        if (!attr) { (void)hipFuncSetAttribute((const void*)conv3d_pp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        ASD_PROBE_START(s);
        hipLaunchKernelGGL((conv3d_pp_kernel<2>), dim3(blocks), dim3(512), lds, s, k);
        ASD_PROBE_STOP(s);
    }
    return ASD_OK;
}

int asd_conv3d_fwd(const asd_conv3d_desc* d, const float* x, const float* w, int64_t w_sample_stride, float* y, const asd_conv3d_epilogue* ep,
                   void* ws, int64_t ws_bytes, void* stream) {
    if (c3_check(d) != ASD_OK) return ASD_ERR_ARG;
    ASD_CHECK_ARG(x && w && y && ws && ws_bytes >= asd_conv3d_workspace_bytes(d, 0), "null argument / workspace too small");
    ASD_CHECK_ARG(!ep || (ep->act >= 0 && ep->act <= 1 && (!ep->noise || ep->noise_strength)), "bad epilogue");
    const size_t vox = (size_t)d->D * d->H * d->W;
    for (int n = 0; n < d->N; ++n)
        c3_run(d->D, d->H, d->W, d->Cin, d->Cout, x + n * vox * d->Cin, d->amax_x, w + (size_t)n * w_sample_stride, 0, y + n * vox * d->Cout, ep, n * vox,
               (char*)ws, (hipStream_t)stream);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_conv3d_dgrad(const asd_conv3d_desc* d, const float* dy, const float* w, int64_t w_sample_stride, float* dx, void* ws, int64_t ws_bytes,
                     void* stream) {
    if (c3_check(d) != ASD_OK) return ASD_ERR_ARG;
    ASD_CHECK_ARG(dy && w && dx && ws && ws_bytes >= asd_conv3d_workspace_bytes(d, 1), "null argument / workspace too small");
    const size_t vox = (size_t)d->D * d->H * d->W;
    for (int n = 0; n < d->N; ++n)
        c3_run(d->D, d->H, d->W, d->Cout, d->Cin, dy + n * vox * d->Cout, d->amax_dy, w + (size_t)n * w_sample_stride, 1, dx + n * vox * d->Cin, nullptr, 0,
               (char*)ws, (hipStream_t)stream);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_conv3d_wgrad(const asd_conv3d_desc* d, const float* x, const float* dy, float* dw, int64_t dw_sample_stride, void* ws, int64_t ws_bytes,
                     void* zero_page, void* stream) {
    if (c3_check(d) != ASD_OK) return ASD_ERR_ARG;
    ASD_CHECK_ARG(x && dy && dw && ws && zero_page && ws_bytes >= asd_conv3d_workspace_bytes(d, 2), "null argument / workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int D = d->D, H = d->H, W = d->W, Cin = d->Cin, Cout = d->Cout;
    const size_t vox = (size_t)D * H * W;
    const c3_cm_layout l = c3_cm(D, H, W);
    ASD_CHECK_ARG(((size_t)(Cin > Cout ? Cin : Cout) * l.ld + 2 * l.guard) * 2 < ((size_t)1 << 31), "a channel-major plane is addressed with 32-bit byte offsets");
    char* base = (char*)ws;
    unsigned* amax = (unsigned*)base;                                 // [0] x, [1] dy
    const size_t plane_x = ((size_t)Cin * l.ld + 2 * l.guard) * 2, plane_y = ((size_t)Cout * l.ld + 2 * l.guard) * 2;
    half_t* xh = (half_t*)(base + 256);
    half_t* xl = (half_t*)((char*)xh + al256(plane_x));
    half_t* yh = (half_t*)((char*)xl + al256(plane_x));
    half_t* yl = (half_t*)((char*)yh + al256(plane_y));
    float* slabs = (float*)((char*)yl + al256(plane_y));
    const int M = 9 * Cin, N = 3 * Cout;
    const int split1 = c3_wgrad_split(d, l.ld, 2), split2 = c3_wgrad_split(d, l.ld, 1);
    ASD_CHECK_ARG((char*)yl - (char*)yh < ((ptrdiff_t)1 << 30), "the two gradient planes must lie within 1 GiB of each other");
    for (int n = 0; n < d->N; ++n) {
        (void)hipMemsetAsync(amax, 0, 8, s);
        const unsigned* ax = d->amax_x ? d->amax_x : amax;
        const unsigned* ay = d->amax_dy ? d->amax_dy : amax + 1;
        if (!d->amax_x) hipLaunchKernelGGL(absmax_kernel, dim3(c3_amax_grid(vox * Cin / 4)), dim3(256), 0, s, x + n * vox * Cin, vox * Cin / 4, amax);
        if (!d->amax_dy) hipLaunchKernelGGL(absmax_kernel, dim3(c3_amax_grid(vox * Cout / 4)), dim3(256), 0, s, dy + n * vox * Cout, vox * Cout / 4, amax + 1);
        // guards and the K tail of every row read zeros: clear the planes' borders (the kernel writes all Vp positions of every row)
        for (int q = 0; q < 4; ++q) {
            half_t* pl = q == 0 ? xh : (q == 1 ? xl : (q == 2 ? yh : yl));
            const int C = q < 2 ? Cin : Cout;
            (void)hipMemsetAsync(pl, 0, l.guard * 2, s);
            (void)hipMemsetAsync(pl + l.guard + (size_t)C * l.ld, 0, l.guard * 2, s);
            if (l.ld > l.Vp) (void)hipMemset2DAsync(pl + l.guard + l.Vp, l.ld * 2, 0, (l.ld - l.Vp) * 2, C, s);
        }
        hipLaunchKernelGGL(split_cm_kernel, dim3((D + 2) * l.Hp, Cin / 32), dim3(256), 0, s, x + n * vox * Cin, D, H, W, Cin, ax, xh, xl, l.ld, (int)l.guard);
        hipLaunchKernelGGL(split_cm_kernel, dim3((D + 2) * l.Hp, Cout / 32), dim3(256), 0, s, dy + n * vox * Cout, D, H, W, Cout, ay, yh, yl, l.ld, (int)l.guard);
        // C[(t9, ci)][(plane, kx, co)] = sum_u XT[ci][u + sA(t9)] * dYT_plane[co][u - (kx - 1)].  Two launches for the three products: X_hi against
        // BOTH planes of dY (the hi and lo rows are six segments of one W operand: the X_hi tile is loaded once for two products), then X_lo
        // against dY_hi
        asd_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.M = M; a.K = (int)l.ld; a.lda = (int)l.ld; a.ldw = (int)l.ld;
        a.zero_page = zero_page; a.partials_only = 1;
        a.a_seg_rows = Cin; a.w_seg_rows = Cout;
        for (int t9 = 0; t9 < 9; ++t9) a.a_seg_off[t9] = (int)(((long long)l.guard + (long long)(t9 / 3 - 1) * l.Hp * l.Wp + (long long)(t9 % 3 - 1) * l.Wp) * 2);
        const long long lo_off = (long long)((char*)yl - (char*)yh);
        for (int q = 0; q < 6; ++q) a.w_seg_off[q] = (int)((q / 3) * lo_off + ((long long)l.guard - (q % 3 - 1)) * 2);
        float* slabs2 = slabs + (size_t)split1 * M * 2 * N;
        a.C = slabs;        // unused (partials_only) but must be non-null
        a.W = yh;
        // launch 1: X_hi . [dY_hi | dY_lo]
        static const int tile_env = getenv("ASD_C3_WGRAD_TILE") ? atoi(getenv("ASD_C3_WGRAD_TILE")) : 0;       // A/B hook (tools): 1-based tile configuration
        a.A = xh; a.N = 2 * N; a.ldc = 2 * N; a.split_k = split1; a.workspace = slabs; a.tile_cfg = tile_env ? tile_env : 8;      // 320 x 128 (6 Cout % 128 == 0): 4.24 vs 4.49 ms with 128 x 128 on 64 -> 64 @128^3 (tools/c3_wgrad_ab.py)
        ASD_PROBE_START(s);
        int rc = asd_gemm_f16(&a, stream);
        ASD_PROBE_STOP(s);
        if (rc != ASD_OK) return rc;
        // launch 2: X_lo . dY_hi
        a.A = xl; a.N = N; a.ldc = N; a.split_k = split2; a.workspace = slabs2; a.tile_cfg = N % 128 == 0 ? 2 : 1;
        rc = asd_gemm_f16(&a, stream);
        if (rc != ASD_OK) return rc;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(asd_div_up((size_t)M * N, 256)), dim3(256), 0, s, slabs, split1, slabs2, split2, Cin, Cout, ax, ay,
                           dw + (size_t)n * dw_sample_stride);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_layer_act_bwd(const float* dy, const float* y, const float* sub, const uint8_t* act_mask, int64_t rows, int32_t C, float gain, float clamp,
                      float* dz, float* d_bias, float* d_rowsum, uint32_t* amax_out, void* stream) {
    ASD_CHECK_ARG(dy && (y || act_mask) && dz && rows > 0 && C % 4 == 0 && C >= 4 && C <= 1024 && 256 % (C / 4) == 0, "C / 4 must divide 256");
    hipStream_t s = (hipStream_t)stream;
    if (d_bias) (void)hipMemsetAsync(d_bias, 0, (size_t)C * 4, s);
    if (d_rowsum && C / 4 > 64) (void)hipMemsetAsync(d_rowsum, 0, (size_t)rows * 4, s);
    const int rpb = 256 / (C / 4);
    int grid = asd_div_up(rows, rpb);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), (size_t)C * 4, s, dy, y, sub, act_mask, (size_t)rows, C, gain, clamp, dz, d_bias, d_rowsum, amax_out);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}


int asd_modulated_weights_fwd(const float* weight, const float* styles, int32_t N, int32_t Cout, int32_t Cin, int32_t K, float gain,
                              int32_t demodulate, float* wm, float* dcoef, void* stream) {
    ASD_CHECK_ARG(weight && styles && wm && dcoef && N > 0 && N <= 8 && Cout > 0 && Cin > 0 && K > 0, "bad argument (1 <= N <= 8)");
    hipLaunchKernelGGL(modw_fwd_kernel, dim3(Cout, N), dim3(256), 0, (hipStream_t)stream, weight, styles, Cin, K, gain, demodulate, wm, dcoef);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_modulated_weights_bwd(const float* d_wm, const float* wm, const float* weight, const float* styles, const float* dcoef, int32_t N,
                              int32_t Cout, int32_t Cin, int32_t K, float gain, int32_t demodulate, float* d_weight, float* d_styles, void* stream) {
    ASD_CHECK_ARG(d_wm && wm && weight && styles && dcoef && d_weight && d_styles && N > 0 && N <= 8 && Cout > 0 && Cin > 0, "bad argument (1 <= N <= 8)");
    ASD_CHECK_ARG(K == 1 || K == 27, "kernel volumes of 1 (toRGB) and 27 (3 x 3 x 3) taps");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(d_styles, 0, (size_t)N * Cin * 4, s);
    if (K == 27)
        hipLaunchKernelGGL((modw_bwd_kernel<27>), dim3(Cout), dim3(256), 0, s, d_wm, wm, weight, styles, dcoef, N, Cin, gain, demodulate, d_weight, d_styles);
    else
        hipLaunchKernelGGL((modw_bwd_kernel<1>), dim3(Cout), dim3(256), 0, s, d_wm, wm, weight, styles, dcoef, N, Cin, gain, demodulate, d_weight, d_styles);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_absmax_f32(const float* x, int64_t n, uint32_t* amax_out, void* stream) {
    ASD_CHECK_ARG(x && amax_out && n > 0 && n % 4 == 0, "n must be a positive multiple of 4");
    hipLaunchKernelGGL(absmax_kernel, dim3(c3_amax_grid((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n / 4, amax_out);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_torgb_fwd(const float* x, int64_t rows, int32_t Cin, const float* w, const float* bias, const float* add, float* y, uint32_t* amax_out, void* stream) {
    ASD_CHECK_ARG(x && w && y && rows > 0 && Cin > 0 && Cin % 32 == 0, "bad argument (Cin % 32 == 0, 32 output channels)");
    hipLaunchKernelGGL(torgb_fwd_kernel, dim3(asd_div_up(rows, 64)), dim3(256), 0, (hipStream_t)stream, x, (size_t)rows, Cin, w, bias, add, y, amax_out);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_torgb_bwd(const float* x, const float* dy, int64_t rows, int32_t Cin, const float* w, const float* dx_add, float* dx, float* dw, float* d_bias,
                  void* stream) {
    ASD_CHECK_ARG(x && dy && w && rows > 0 && Cin > 0 && Cin % 64 == 0, "bad argument (Cin % 64 == 0, 32 output channels)");
    hipStream_t s = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(torgb_dgrad_kernel, dim3(asd_div_up(rows, 64)), dim3(256), 0, s, dy, (size_t)rows, Cin, w, dx_add, dx);
    if (dw) {
        (void)hipMemsetAsync(dw, 0, (size_t)TORGB_O * Cin * 4, s);
        if (d_bias) (void)hipMemsetAsync(d_bias, 0, TORGB_O * 4, s);
        size_t rpb = (size_t)asd_div_up(rows, 2048);
        if (rpb < 64) rpb = 64;
        hipLaunchKernelGGL(torgb_wgrad_kernel, dim3(asd_div_up(rows, rpb)), dim3(256), 0, s, x, dy, (size_t)rows, Cin, rpb, dw, d_bias);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_upsample3d_fwd(const float* x, int32_t N, int32_t r, int32_t C, const asd_conv3d_epilogue* ep, const float* add, float* y, uint8_t* act_mask,
                       void* stream) {
    ASD_CHECK_ARG(x && y && N > 0 && r > 0 && C % 4 == 0 && C > 0, "bad argument");
    ASD_CHECK_ARG(!ep || (ep->act >= 0 && ep->act <= 1 && (!ep->noise || ep->noise_strength)), "bad epilogue");
    const size_t total = (size_t)N * 8 * r * r * r * (C / 4);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(c3_grid(total) * 2), dim3(256), 0, (hipStream_t)stream, x, N, r, C, ep ? ep->noise : nullptr,
                       ep ? ep->noise_strength : nullptr, ep ? ep->bias : nullptr, ep ? ep->act : 0, ep ? ep->gain : 1.f, ep ? ep->clamp : 0.f, add, y,
                       ep ? ep->amax_out : nullptr, act_mask);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_upsample3d_bwd(const float* dy, int32_t N, int32_t r, int32_t C, float* dx, float* ws /* 6 N r^3 C floats */, void* stream) {
    ASD_CHECK_ARG(dy && dx && ws && N > 0 && r > 0 && C % 4 == 0 && C > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t R = 2 * (size_t)r, c4 = C / 4;
    float* t1 = ws;                                     // [N][2r][2r][r][C]
    float* t2 = ws + (size_t)N * R * R * r * C;         // [N][2r][r][r][C]
    hipLaunchKernelGGL(upsample1d_bwd_kernel, dim3(c3_grid((size_t)N * R * R * r * c4) * 2), dim3(256), 0, s, dy, (size_t)N * R * R, r, c4, t1);
    hipLaunchKernelGGL(upsample1d_bwd_kernel, dim3(c3_grid((size_t)N * R * r * r * c4) * 2), dim3(256), 0, s, t1, (size_t)N * R, r, r * c4, t2);
    hipLaunchKernelGGL(upsample1d_bwd_kernel, dim3(c3_grid((size_t)N * r * r * r * c4) * 2), dim3(256), 0, s, t2, (size_t)N, r, (size_t)r * r * c4, dx);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
