// gemm_pp.hip — "ping-pong" LDS-window 3x3 convolution for gfx950: eight waves per block, two per SIMD, the two halves of the block
// staggered by one barrier so that on every SIMD one wave is in an MFMA segment while its partner reads fragments, issues the
// global->LDS tile loads and does the addressing of the next segment (MI355X_MICROARCH.md, "Two waves per SIMD"; the 8-phase GEMM
// template of cdna_hip_programming.md).  Same problem and C ABI as conv3x3_win_kernel of gemm.hip (stride-1 pad-1 3x3 convolution,
// NHWC fp16, weights packed [Cout][ky][kx][Cin]; replaces the cuDNN calls behind diffusers' ResnetBlock2D / the LDM Encoder:
// extern/mvdream/ldm/modules/diffusionmodules/openaimodel.py:163-275, model.py:452-543), different schedule:
//
//   * K-step = one filter tap x 32 input channels (64-byte LDS rows).  A block owns a (4 TM) x 16 pixel patch x BN = 32 TN channels;
//     wave (wm, wn) owns TM patch rows x 16 TN channels (TM x TN fragments of v_mfma_f32_16x16x32_f16).  A K-step is one or two
//     PHASES (the two halves of the wave's patch rows); a phase is
//         [fragment reads, tile loads of later K-steps, counted vmcnt]  s_barrier  [16-20 MFMAs at s_setprio 1]  s_barrier
//     and waves 4-7 run one barrier behind waves 0-3, so the bracketed halves of the two groups alternate.
//   * the input window ((4 TM + 2) x 18 pixels x 32 channels) is double buffered: the next 32-channel chunk arrives during taps 1-4
//     of the current one; weight tiles (BN x 32 per tap) go through a three-slot ring, two K-steps ahead.  Loads are never drained:
//     every wait is s_waitcnt vmcnt(n) with n = the loads issued after the tile that must have landed, one phase before its first read
//     (LDS-DMA data is ordered for other waves' ds_reads only by the issuing wave's vmcnt followed by a barrier).
//   * 64-byte rows, 16-byte chunk q of row r stored at chunk q ^ 2*bit2(r): ds_read_b128 of 16 consecutive rows is conflict-free for
//     every row offset (checked against the lane groups of the LDS table of the microarchitecture guide), so the nine taps are
//     immediate offsets from three per-lane bases (one per kx); window lines have a pitch of 24 rows (18 used) to keep bit 2 of
//     the row a function of the column alone.
//   * split-K slices the 32-channel chunks; epilogue, wide rows, GroupNorm records: tile_epilogue of gemm_tile.h.
#include "gemm_tile.h"

template <int N>
__device__ __forceinline__ void pp_vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wave-uniform n (scalar branches); n <= 7
__device__ __forceinline__ void pp_vm_wait_n(int n) {
    switch (n) {
        case 0: pp_vm_wait<0>(); break;
        case 1: pp_vm_wait<1>(); break;
        case 2: pp_vm_wait<2>(); break;
        case 3: pp_vm_wait<3>(); break;
        case 4: pp_vm_wait<4>(); break;
        case 5: pp_vm_wait<5>(); break;
        case 6: pp_vm_wait<6>(); break;
        default: pp_vm_wait<7>(); break;
    }
}
// accumulate IN PLACE.  Through the builtin hipcc gives most MFMAs of the unrolled tap loop a destination different from their
// accumulator operand (a rotating set of 40-60 extra registers: 227 VGPRs at 128 accumulators, spills inside the loop at 160).  No
// dependent MFMA follows within a phase (every accumulator is used once per phase), the epilogue reads them after two barriers.
__device__ __forceinline__ void pp_mfma(floatx4& c, const half8& w, const half8& x) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}
// A/B switches of tools/pp_variants.sh (results unchanged): -DASD_PP_NO_STAGGER runs the two halves of the block in lockstep (all eight
// waves read / load together, then multiply together), -DASD_PP_NO_PRIO drops the s_setprio around the MFMA segments
#ifdef ASD_PP_NO_PRIO
#define PP_PRIO(x) do { } while (0)
#else
#define PP_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifdef ASD_PP_NO_STAGGER
#define PP_STAGGER 0
#else
#define PP_STAGGER 1
#endif
#define PP_PIN() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER()                    \
    do {                                \
        PP_PIN();                       \
        __builtin_amdgcn_s_barrier();   \
        PP_PIN();                       \
    } while (0)

// -DASD_PP_PROFILE (tools/pp_profile.py): s_memtime accounting per wave — prologue, load/read segments, waits at the two barriers of a
// phase, MFMA segments, epilogue — written to p.workspace (split_k == 1 launches only)
#ifdef ASD_PP_PROFILE
#define PT_NOW() __builtin_amdgcn_s_memtime()
#define PT_ADD(acc) do { const unsigned long long n__ = PT_NOW(); acc += n__ - pt_last; pt_last = n__; } while (0)
#else
#define PT_ADD(acc) do { } while (0)
#endif

template <int TM, int TN>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(const asd_gemm_args p) {
    constexpr int WM = 4, WN = 2, PH = WM * TM, BN = WN * TN * 16, PITCH = 24;
    constexpr int WLINES = PH + 2, WIN_BYTES = WLINES * PITCH * 64, W_BYTES = BN * 64;
    constexpr int WSLABS = BN / 16, NWL = (WSLABS + 7) / 8;         // weight slabs (16 rows x 64 B) per K-step, per wave
    constexpr int NPIECE = 3 * WLINES / 2;                           // window pieces (1 KiB = 16 rows) per chunk: three per pair of lines
    constexpr int WROUNDS = (NPIECE + 7) / 8, PPT = (WROUNDS + 3) / 4;   // pieces per wave in each of taps 1..4
    constexpr int NPH = (TM * TN >= 24) ? 2 : 1;                     // phases per K-step
    static_assert(WLINES % 2 == 0 && (NPH == 1 || (TN > TM ? TN : TM) % 2 == 0), "window lines are loaded in pairs; the long side of the wave tile splits over the phases");
    static_assert(NWL + PPT <= 7, "pp_vm_wait_n covers 0..7");
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [window 0 | window 1 | weights 0 | 1 | 2]
    char* const win = smem;
    char* const wring = smem + 2 * WIN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;          // waves w and w + 4 share a SIMD: one of each group
    const int grp = wn;
    const int H = p.Hout, Wd = p.Wout;
    const int tiles_x = Wd / 16, tiles_y = H / PH;
    const int tiles_m = (p.M / (H * Wd)) * tiles_y * tiles_x;
    const int tiles_n = (p.N + BN - 1) / BN;
    int item, tm, tn_;
    if (!asd_xcd_item(blockIdx.x, tiles_m * tiles_n * p.split_k, item)) return;
    const int kz = item / (tiles_m * tiles_n);
    asd_grouped_tile(item - kz * tiles_m * tiles_n, tiles_m, tiles_n, p.group_m, p.group_n, tm, tn_);
    const int n0 = tn_ * BN;
    const int b = tm / (tiles_y * tiles_x), tr = tm - b * tiles_y * tiles_x;
    const int y0 = (tr / tiles_x) * PH, x0 = (tr - (tr / tiles_x) * tiles_x) * 16;
    const int n_chunks = p.Cin / 32;
    const int c_per = (n_chunks + p.split_k - 1) / p.split_k;
    const int c0 = kz * c_per, c1 = min(n_chunks, c0 + c_per);

    const char* img = (const char*)p.A + (size_t)b * H * Wd * p.Cin * 2;
    const char* const Wp = (const char*)p.W;

    // ---- loaders: lane -> (row rho = lane >> 2 of a 16-row slab, physical chunk lane & 3) -------------------------------------------
    const int rho = lane >> 2, pch = lane & 3;
    const bool wide = p.wide_rows != 0;                // weight rows in permuted order (tile_epilogue: 8 consecutive channels per lane)
    unsigned woff[NWL];                                // byte offset of this lane's 16 B inside W for tap 0 / chunk 0
    int wslab[NWL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        int sl = wave + 8 * j;
        if (sl >= WSLABS) sl -= 8;                     // BN = 320: the waves without a third slab repeat their second one
        wslab[j] = sl;
        const int R = sl * 16 + rho;                   // LDS row of the tile
        const int ch = wide ? ((R >> 5) * 32 + ((R >> 2) & 3) * 8 + ((R >> 4) & 1) * 4 + (R & 3)) : R;
        const int q = pch ^ (((rho >> 2) & 1) << 1);
        woff[j] = (unsigned)min(n0 + ch, p.N - 1) * (unsigned)(p.ldw * 2) + q * 16;
    }
    // scalar operand base + one 32-bit lane offset per load (global_load_lds saddr + voffset); the base is made opaque so that the
    // compiler does not fold it into per-lane 64-bit addresses (two registers per load, live across the whole loop)
    auto load_w = [&](int c, int t, int slot) __attribute__((always_inline)) {
        const char* base = Wp + ((size_t)t * p.Cin + (size_t)c * 32) * 2;
        asm volatile("" : "+s"(base));
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            unsigned o = woff[j];
            asm volatile("" : "+v"(o));                // keeps the zero-extension next to the load (saddr + 32-bit voffset is matched per block)
            load_slab(base + o, wring + slot * W_BYTES + wslab[j] * 1024);
        }
    };
    // window piece pc of chunk c: rows [16 k, 16 k + 16) of the 48-row pair of window lines pc / 3.  Rows outside the image (and the six
    // padding rows of a line) are not loaded: their lanes are masked off and the rows were zeroed once in the prologue (the validity of a
    // row does not depend on the chunk).  ZERO: that prologue pass over both buffers instead of the load.
    auto piece = [&](int pc, int c, char* wbuf, bool zero_pass) __attribute__((always_inline)) {
        const int pair = pc / 3, k = pc - pair * 3;
        int rho_o = rho;                               // opaque: the per-piece pixel offsets are recomputed at the call, not hoisted out of
        asm volatile("" : "+v"(rho_o));                // the chunk loop into live registers (the accumulators need them)
        const int o = k * 16 + rho_o;
        const int second = o >= PITCH ? 1 : 0;
        const int wy = 2 * pair + second, col = o - PITCH * second;
        const int yi = y0 - 1 + wy, xi = x0 - 1 + col;
        const bool ok = col < 18 && (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)Wd;
        char* const dst = wbuf + (pair * 48 + k * 16) * 64;
        if (zero_pass) {
            if (!ok) {
                *(uint4*)(dst + lane * 16) = uint4{0u, 0u, 0u, 0u};
                *(uint4*)(dst + WIN_BYTES + lane * 16) = uint4{0u, 0u, 0u, 0u};
            }
            return;
        }
        const int q = pch ^ (((col >> 2) & 1) << 1);
        const unsigned off = (unsigned)(yi * Wd + xi) * (unsigned)(p.Cin * 2) + q * 16;
        const char* base = img + (size_t)c * 64;
        asm volatile("" : "+s"(base));
        if (ok) load_slab(base + off, dst);
    };
    auto load_piece = [&](int pc, int c, char* wbuf) __attribute__((always_inline)) { piece(pc, c, wbuf, false); };

    // ---- fragment addressing: lane -> (row i = lane & 15 of a 16-row fragment, logical chunk fq = lane >> 4) --------------------------
    const int fi = lane & 15, fq = lane >> 4;
    int la[3];                                         // window: patch row 0 of the wave, tap column kx
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

#ifdef ASD_PP_PROFILE
    const unsigned long long pt_start = PT_NOW(), pt_wall0 = wall_clock64();     // wall_clock64: constant 100 MHz
    unsigned long long pt_last = pt_start, pt_pro = 0, pt_l = 0, pt_b1 = 0, pt_m = 0, pt_b2 = 0, pt_epi = 0;
#endif
    if (c0 < c1) {
        // ---- prologue: the first window, weight tiles of K-steps 0 and 1 ------------------------------------------------------------
        for (int pc = wave; pc < NPIECE; pc += 8) {
            piece(pc, c0, win, true);
            load_piece(pc, c0, win);
        }
        load_w(c0, 0, 0);
        load_w(c0, 1, 1);
        pp_vm_wait<0>();
        PP_BARRIER();
        if (PP_STAGGER && grp == 1) PP_BARRIER();      // the second group runs one barrier behind
        PT_ADD(pt_pro);

#pragma unroll 1
        for (int c = c0; c < c1; ++c) {
            const bool last = c + 1 >= c1;
            const int par = (c - c0) & 1;
            const char* const wcur = win + par * WIN_BYTES;
            char* const wnext = win + (par ^ 1) * WIN_BYTES;
            int nwin_prev = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - ky * 3;
                const char* const Ab = wcur + la[kx] + ky * PITCH * 64;
                const char* const Ws = wring + (t % 3) * W_BYTES + lb;
                const bool iss = !(last && t >= 7);    // K-step s + 2 exists
                // a K-step is split along the longer side of the wave tile: the fragments of the short side are read once and kept
                constexpr bool SPLIT_N = NPH == 2 && TN > TM;
                constexpr int NA = (NPH == 2 && !SPLIT_N) ? TM / 2 : TM, NB = SPLIT_N ? TN / 2 : TN;
                half8 B[NB], A[NA];
                // ---------------- phase 1 ---------------------------------------------------------------------------------------
#pragma unroll
                for (int j = 0; j < NB; ++j) B[j] = *(const half8*)(Ws + j * 1024);
#pragma unroll
                for (int a = 0; a < NA; ++a) A[a] = *(const half8*)(Ab + a * PITCH * 64);
                PP_PIN();
                int nwin = 0;
                if (NPH == 1 && !last && t >= 1 && t <= 4) {
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int pc = ((t - 1) * PPT + k) * 8 + wave;
                        if (pc < NPIECE) { load_piece(pc, c + 1, wnext); ++nwin; }
                    }
                }
                if (iss) {
                    if (t + 2 < 9) load_w(c, t + 2, (t + 2) % 3);
                    else load_w(c + 1, t + 2 - 9, (t + 2) % 3);
                    pp_vm_wait_n(NWL + (NPH == 1 ? nwin : nwin_prev));      // weights of K-step s + 1 have landed
                } else if (t == 7) {
                    pp_vm_wait<0>();
                }
                if (NPH == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the partner may refill the slot
                PT_ADD(pt_l);
                PP_BARRIER();
                PT_ADD(pt_b1);
                PP_PRIO(1);
#pragma unroll
                for (int a = 0; a < NA; ++a)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        pp_mfma(acc[a][j], B[j], A[a]);
                PP_PRIO(0);
                PT_ADD(pt_m);
                PP_BARRIER();
                PT_ADD(pt_b2);
                // ---------------- phase 2: the other half of the long side ------------------------------------------------------
                if constexpr (NPH == 2) {
                    if constexpr (SPLIT_N) {
#pragma unroll
                        for (int j = 0; j < NB; ++j) B[j] = *(const half8*)(Ws + (NB + j) * 1024);
                    } else {
#pragma unroll
                        for (int a = 0; a < NA; ++a) A[a] = *(const half8*)(Ab + (NA + a) * PITCH * 64);
                    }
                    PP_PIN();
                    if (!last && t >= 1 && t <= 4) {
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            const int pc = ((t - 1) * PPT + k) * 8 + wave;
                            if (pc < NPIECE) { load_piece(pc, c + 1, wnext); ++nwin; }
                        }
                    }
                    PT_ADD(pt_l);
                    PP_BARRIER();
                    PT_ADD(pt_b1);
                    PP_PRIO(1);
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            pp_mfma(SPLIT_N ? acc[a][NB + j] : acc[NA + a][j], B[j], A[a]);
                        }
                    PP_PRIO(0);
                    PT_ADD(pt_m);
                    PP_BARRIER();
                    PT_ADD(pt_b2);
                }
                nwin_prev = nwin;
            }
        }
        if (PP_STAGGER && grp == 0) PP_BARRIER();      // the first group waits for the second one's last segment
    }

    // acc[i][j][r] = C[pixel (y0 + wm*TM + i, x0 + (lane&15))][n0 + wn*TN*16 + j*16 + (lane>>4)*4 + r]
    const bool gn = p.gn_partials != nullptr && p.split_k == 1;     // block-uniform
    float* gn_lds = (float*)smem;
    if (gn) gn_tile_begin(gn_lds);
    tile_epilogue<TM, TN>(p, acc, n0 + wn * TN * 16, kz, b * H * Wd,
                          [&](int i) { return (b * H + y0 + wm * TM + i) * Wd + x0 + fi; }, gn, gn_lds);
    if (gn) gn_tile_end(p, gn_lds, tm * tiles_n + tn_);
#ifdef ASD_PP_PROFILE
    if (p.split_k == 1 && p.workspace && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PT_ADD(pt_epi);
        unsigned long long* o = (unsigned long long*)p.workspace + ((size_t)item * 8 + wave) * 16;
        o[0] = pt_start; o[1] = pt_last; o[2] = pt_pro; o[3] = pt_l; o[4] = pt_b1; o[5] = pt_m; o[6] = pt_b2; o[7] = pt_epi;
        o[8] = pt_wall0; o[9] = wall_clock64();
    }
#endif
}

// ---- host side: variants and launch (called by asd_gemm_f16 of gemm.hip) ----------------------------------------------------------
struct asd_pp_variant { int tm, tn; };
static const asd_pp_variant asd_pp_variants[] = {{8, 4}, {4, 8}, {4, 10}, {4, 4}, {4, 5}};

size_t asd_conv_pp_lds_bytes(int variant) {
    const int tm = asd_pp_variants[variant].tm, tn = asd_pp_variants[variant].tn;
    return (size_t)2 * (4 * tm + 2) * 24 * 64 + (size_t)3 * (2 * tn * 16) * 64;
}

int asd_conv_pp_launch(int variant, const asd_gemm_args* a, int blocks, hipStream_t s) {
    const size_t lds = asd_conv_pp_lds_bytes(variant);
#define PP_LAUNCH(TM_, TN_)                                                                                                       \
    do {                                                                                                                          \
        static std::atomic<unsigned long long> attr_set_devmask{0}; bool attr_set = !asd_attr_needed(attr_set_devmask);                                                                                             \
        if (!attr_set) {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)conv3x3_pp_kernel<TM_, TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv3x3_pp_kernel<TM_, TN_>), dim3(blocks), dim3(512), lds, s, *a);                                   \
    } while (0)
    switch (variant) {
        case 0: PP_LAUNCH(8, 4); break;
        case 1: PP_LAUNCH(4, 8); break;
        case 2: PP_LAUNCH(4, 10); break;
        case 3: PP_LAUNCH(4, 4); break;
        case 4: PP_LAUNCH(4, 5); break;
        default: return ASD_ERR_ARG;
    }
#undef PP_LAUNCH
    return ASD_OK;
}
