// gemm_ws.hip — weight-streaming 3x3 convolution for the UNet's 8x8 level (tile configuration 25) on gfx950.
//
// The layers: ResBlock convolutions 1280 -> 1280 and 2560 -> 1280 on B <= 5 images of 8 x 8 pixels (M = 64 B <= 320 rows, K = 9 Cin =
// 11520 / 23040, 29.5 / 59 MB of weights that every launch streams from HBM once; openaimodel.py:163-275 ResBlock, :563-752 the level
// structure).  As an implicit GEMM on 64 x 64 tiles every (tile, split, stage) combination sits on a 27-33 us plateau
// (docs/DESIGN_LOG_r3.md): each of the 20 N tiles re-reads the im2col'd activations (9 x the raw tensor) through L2 -> LDS, 300 MB per
// launch, or — with wide N tiles — a 20-way split-K exchanges 33 MB of fp32 slabs.  This kernel cuts both:
//   * a block owns ALL M rows x 64 output channels x one channel slice (split_k slices of Cin / split_k channels, all nine taps), so
//     the weights are read exactly once per launch and the slab count stays at Cin / 128 .. Cin / 256;
//   * per 32-channel chunk the RAW activations of all images (B x 64 pixels, 64-byte rows) come into LDS once and serve the nine
//     taps as row shifts — 9 x less L2 -> LDS traffic than the im2col'd gather; pixels outside an image are masked to zero in the
//     fragment registers (no halo in LDS);
//   * the weight stream runs TWO chunks ahead (three-slot ring of [9 taps][64 channels][32 k] tiles, 36 KB each), the activations one
//     (two slots).  The first version had one chunk of both in flight and one vmcnt(0) per chunk: every chunk then cost a full
//     HBM round trip (2.3 us against 1.5 us of MFMAs; timing ablations without weights / activations / MFMAs / stores all landed
//     within 10 % — gpurun_out/r6e).  vmcnt counts a wave's loads in order, so a stream with a deep lead and one with a short deadline
//     cannot share a wave: waves 0-1 issue the weight tiles and wait with vmcnt(18) (the newest chunk stays in flight), waves 2-3 issue
//     the activations and drain; the block barrier behind both waits publishes the data to everyone.
// Four waves = 2 (halves of the row tiles) x 2 (32 output channels); a wave holds B row tiles of 32 x 32 (v_mfma_f32_32x32x16_f16, the
// weight fragment of a k-step is read once and serves its B MFMAs).  64-byte LDS rows, 16-byte chunk q of a row stored at
// q ^ (key & 3) with key = the image line (activations) / (channel >> 2) (weights): a ds_read_b128's 16-lane groups then fall on 16
// distinct bank quads for every tap.  Output: fp32 slabs workspace[split_k][M][N] — bias, time embedding, residual, GroupNorm follow in
// the split-K reduction kernels of gemm.hip.
#include <type_traits>

#include "gemm_tile.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

#define WS_A_BYTES(NB) ((NB) * 64 * 64)          // B x 64 pixels x 32 channels
#define WS_W_BYTES (9 * 64 * 64)
#define WS_PAD 1024                               // in front of and behind the activation slots: masked lanes read up to 9 rows outside
#define WS_LDS_BYTES(NB) (3 * WS_W_BYTES + 2 * WS_A_BYTES(NB) + 2 * WS_PAD)

template <int NB>
__global__ __launch_bounds__(256, 1) void conv3x3_ws_kernel(const asd_gemm_args p) {
    constexpr int A_BYTES = WS_A_BYTES(NB), W_BYTES = WS_W_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024;                    // wave-instructions (1 KB = 16 pixels each) per chunk: 4 NB
    constexpr int A_PER_WAVE = (A_INSTR + 1) / 2;              // two loader waves
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [W0 | W1 | W2 | pad | A0 | A1 | pad]
    char* const w_base = smem;
    char* const a_base = smem + 3 * W_BYTES + WS_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the roles branch, not mask
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / 64;
    const int tn = (int)blockIdx.x % tiles_n, kz = (int)blockIdx.x / tiles_n;
    const int n0 = tn * 64;
    const int cs = p.Cin / p.split_k, c0 = kz * cs, n_chunks = cs / 32;
    const bool w_loader = wave < 2;

    // ---- loaders -------------------------------------------------------------------------------------------------------------------------
    // activations (waves 2, 3): piece q covers pixels 16 q .. 16 q + 15 of the B x 64, lane -> (pixel lane / 4, 16-byte piece lane & 3 of its
    // 64-byte row, stored at (lane & 3) ^ (image line & 3))
    // (the image line of pixel 16 q + lane / 4 is 2 q + (lane >> 5): its low two bits depend on q's parity)
    const int a_px = lane >> 2;
    const char* a_lane = (const char*)p.A + ((size_t)a_px * p.Cin + c0) * 2;
    const int a_piece_even = ((lane & 3) ^ (a_px >> 3)) * 16, a_piece_odd = ((lane & 3) ^ (2 + (a_px >> 3))) * 16;
    // weights (waves 0, 1): piece t = 18 wave + j: tap t / 4, channels 16 (t & 3) .. + 15; lane -> (channel lane / 4, piece lane & 3 stored at
    // (lane & 3) ^ ((lane >> 4) & 3))
    const char* w_lane = (const char*)p.W + ((size_t)(n0 + (lane >> 2)) * p.ldw + c0) * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);

    auto issue_w = [&](int chunk) {           // 18 loads
        char* W_s = w_base + (chunk % 3) * W_BYTES;
#ifdef WS_ABL_NOW
        if (chunk >= 0) return;
#endif
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int t = wave * 18 + j;
            const size_t off = ((size_t)((t & 3) * 16) * p.ldw + (size_t)(t >> 2) * p.Cin + chunk * 32) * 2;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(w_lane + off), (LDS_AS void*)(W_s + t * 1024), 16, 0, 0);
        }
    };
    auto issue_a = [&](int chunk) {
        char* A_s = a_base + (chunk & 1) * A_BYTES;
#ifdef WS_ABL_NOA
        if (chunk >= 0) return;
#endif
#pragma unroll
        for (int j = 0; j < A_PER_WAVE; ++j) {
            const int q = (wave - 2) * A_PER_WAVE + j;
            if (q < A_INSTR)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(a_lane + ((size_t)q * 16 * p.Cin + chunk * 32) * 2 + ((q & 1) ? a_piece_odd : a_piece_even)),
                                                 (LDS_AS void*)(A_s + q * 1024), 16, 0, 0);
        }
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------------------------------
    // A operand (32 rows x 16 k): lane -> row lane & 31 = pixel (line yy = (lane & 31) >> 3 [+ 4 for the odd tile of an image], xx = lane & 7),
    // k half lane >> 5.  Under tap (ky, kx) it reads pixel (Y = line + ky - 1, X = xx + kx - 1): LDS row 32 mt + 8 (ky - 1) + (kx - 1) +
    // (8 yy + xx) of the chunk's slot, piece (2 j + half) ^ (Y & 3) for k-step j; lanes whose (Y, X) leaves the image read a neighbouring
    // row (the pads absorb the ends) and are ANDed to zero — a select would become a predicated load with its own wait.
    // Address = per-lane base(ky, j) [slot, wave's first row tile, -9 rows, swizzled piece] + immediate (2048 i + 64 (8 ky + kx)).
    const int l31 = lane & 31, kh = lane >> 5, yy = l31 >> 3, xx = l31 & 7;
    const unsigned m_x_first = xx == 0 ? 0u : 0xffffffffu, m_x_last = xx == 7 ? 0u : 0xffffffffu;
    const unsigned m_y_first = yy == 0 ? 0u : 0xffffffffu, m_y_last = yy == 3 ? 0u : 0xffffffffu;
    // row tile wm * NB + i is the upper half of its image when that number is even: per parity of i
    const bool top0 = ((wm * NB) & 1) == 0, top1 = ((wm * NB + 1) & 1) == 0;
    const unsigned m_ky0[2] = {top0 ? m_y_first : 0xffffffffu, top1 ? m_y_first : 0xffffffffu};
    const unsigned m_ky2[2] = {top0 ? 0xffffffffu : m_y_last, top1 ? 0xffffffffu : m_y_last};
    const unsigned a_lane_lds = (unsigned)(size_t)a_base + (unsigned)((wm * NB * 32 + yy * 8 + xx - 9) * 64);
    const unsigned w_lane_lds = (unsigned)(size_t)w_base + (unsigned)((wn * 32 + l31) * 64);
    const int wkey = (l31 >> 2) & 3;
    unsigned a_piece[3][2], w_piece[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w_piece[j] = (unsigned)(((2 * j + kh) ^ wkey) * 16);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) a_piece[ky][j] = (unsigned)((((2 * j + kh) ^ ((yy + ky - 1) & 3))) * 16);
    }

    floatx16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    struct Frag { uintx4 a[NB]; uintx4 w; };
#define WS_READ16(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(dst) : "v"(addr), "n"(OFF) : "memory")

    if (w_loader) {
        issue_w(0);
        if (n_chunks > 1) { issue_w(1); asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        issue_a(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();             // a bare barrier: __syncthreads() carries a fence that drains vmcnt — the loads in flight
    for (int c = 0; c < n_chunks; ++c) {
        if (w_loader) { if (c + 2 < n_chunks) issue_w(c + 2); }
        else if (c + 1 < n_chunks) issue_a(c + 1);
        const unsigned a_slot = a_lane_lds + (unsigned)((c & 1) * A_BYTES), w_slot = w_lane_lds + (unsigned)((c % 3) * W_BYTES);
        unsigned aaddr[3][2], waddr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            waddr[j] = w_slot + w_piece[j];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) aaddr[ky][j] = a_slot + a_piece[ky][j];
        }
        // the fragments of step s + 1 (tap (s + 1) / 2, k-step (s + 1) & 1: NB + 1 reads) fly while the NB MFMAs of step s run: two register sets,
        // counted lgkmcnt (left to itself the compiler keeps ONE read ahead and waits lgkmcnt(0) in front of every MFMA — one wave per SIMD
        // has nobody to hide the LDS latency behind)
        auto read_frags = [&](Frag& f, auto step_c) {
            constexpr int step = decltype(step_c)::value, tap = step >> 1, j = step & 1, ky = tap / 3, kx = tap - ky * 3;
            WS_READ16(f.w, waddr[j], tap * 4096);
#define WS_RA(I_) if constexpr (I_ < NB) WS_READ16(f.a[I_], aaddr[ky][j], I_ * 2048 + (8 * ky + kx) * 64)
            WS_RA(0); WS_RA(1); WS_RA(2); WS_RA(3); WS_RA(4);
#undef WS_RA
        };
        auto mma = [&](Frag& f, auto step_c) {
            constexpr int step = decltype(step_c)::value, tap = step >> 1, ky = tap / 3, kx = tap - ky * 3;
            const unsigned mx = kx == 0 ? m_x_first : (kx == 2 ? m_x_last : 0xffffffffu);
            const unsigned m0 = mx & (ky == 0 ? m_ky0[0] : (ky == 2 ? m_ky2[0] : 0xffffffffu));
            const unsigned m1 = mx & (ky == 0 ? m_ky0[1] : (ky == 2 ? m_ky2[1] : 0xffffffffu));
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (tap != 4) {
                    const unsigned m = (i & 1) ? m1 : m0;
                    f.a[i][0] &= m; f.a[i][1] &= m; f.a[i][2] &= m; f.a[i][3] &= m;
                }
#ifdef WS_ABL_NOMMA
                acc[i][0] += __uint_as_float(f.a[i][0]) * __uint_as_float(f.w[0]);
#else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, f.a[i]), __builtin_bit_cast(half8, f.w), acc[i], 0, 0, 0);
#endif
            }
        };
        auto wait_frags = [&](Frag& f, auto pending_c) {      // wait until at most `pending` LDS reads are outstanding; the fragment registers order the MFMAs behind it
            constexpr int pending = decltype(pending_c)::value;
            if constexpr (NB == 5) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.a[4]), "+v"(f.w) : "n"(pending));
            else if constexpr (NB == 4) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.w) : "n"(pending));
            else if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.w) : "n"(pending));
            else if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.w) : "n"(pending));
            else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[0]), "+v"(f.w) : "n"(pending));
        };
        Frag f0, f1;
        read_frags(f0, std::integral_constant<int, 0>{});
#define WS_STEP(S_, CUR_, NXT_)                                                                                   \
        if constexpr (S_ + 1 < 18) {                                                                              \
            read_frags(NXT_, std::integral_constant<int, (S_ + 1 < 18 ? S_ + 1 : 17)>{});                          \
            wait_frags(CUR_, std::integral_constant<int, NB + 1>{});                                              \
        } else wait_frags(CUR_, std::integral_constant<int, 0>{});                                                \
        mma(CUR_, std::integral_constant<int, S_>{});                                                             \
        __builtin_amdgcn_sched_barrier(0)
        WS_STEP(0, f0, f1); WS_STEP(1, f1, f0); WS_STEP(2, f0, f1); WS_STEP(3, f1, f0); WS_STEP(4, f0, f1); WS_STEP(5, f1, f0);
        WS_STEP(6, f0, f1); WS_STEP(7, f1, f0); WS_STEP(8, f0, f1); WS_STEP(9, f1, f0); WS_STEP(10, f0, f1); WS_STEP(11, f1, f0);
        WS_STEP(12, f0, f1); WS_STEP(13, f1, f0); WS_STEP(14, f0, f1); WS_STEP(15, f1, f0); WS_STEP(16, f0, f1); WS_STEP(17, f1, f0);
#undef WS_STEP
        // the next chunk's operands: every wave waits for its own loads (the weight loaders leave chunk c + 2 in flight), the barrier
        // publishes them — and tells the loaders that this chunk's slots are free (all fragment reads were waited for above)
        if (w_loader && c + 2 < n_chunks) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef WS_READ16

    // ---- partial sums -> slab kz: acc[i][4 q + r] = C[row 32 mt + 8 q + 4 kh + r][n0 + 32 wn + (lane & 31)] ----------------------------------
    float* slab = p.workspace + (size_t)kz * p.M * p.N + n0 + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int mt = wm * NB + i;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 32 + q * 8 + kh * 4 + r;
#ifdef WS_ABL_NOSTORE
                if (acc[i][q * 4 + r] == 12345.678f) slab[(size_t)m * p.N] = acc[i][q * 4 + r];
#else
                slab[(size_t)m * p.N] = acc[i][q * 4 + r];            // M == 64 NB: every row tile is whole
#endif
            }
    }
}

size_t asd_conv_ws_lds_bytes(int images) { return (size_t)WS_LDS_BYTES(images); }

// 3x3 stride-1 pad-1 convolution on B <= 5 images of 8 x 8 pixels, Cin % (32 split_k) == 0, N % 64 == 0, split_k >= 2 (the caller checks)
int asd_conv_ws_launch(const asd_gemm_args* a, hipStream_t s) {
    const int images = a->M / 64;
    const dim3 grid((a->N / 64) * a->split_k);
#define WS_LAUNCH(NB_)                                                                                                             \
    do {                                                                                                                          \
        static std::atomic<unsigned long long> attr_set_devmask{0};                                                               \
        const size_t lds = asd_conv_ws_lds_bytes(NB_);                                                                            \
        if (asd_attr_needed(attr_set_devmask))                                                                                    \
            (void)hipFuncSetAttribute((const void*)conv3x3_ws_kernel<NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((conv3x3_ws_kernel<NB_>), grid, dim3(256), lds, s, *a);                                                \
    } while (0)
    switch (images) {
        case 1: WS_LAUNCH(1); break;
        case 2: WS_LAUNCH(2); break;
        case 3: WS_LAUNCH(3); break;
        case 4: WS_LAUNCH(4); break;
        case 5: WS_LAUNCH(5); break;
        default: return ASD_ERR_ARG;
    }
#undef WS_LAUNCH
    return ASD_OK;
}
