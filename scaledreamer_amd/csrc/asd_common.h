// asd_common.h — shared host/device helpers for libasd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/asd_hip.h"

#define ASD_WAVE 64

void asd_set_error(const char* fmt, ...);

#define ASD_CHECK_ARG(cond, msg)                          \
    do {                                                  \
        if (!(cond)) {                                    \
            asd_set_error("%s: %s", __func__, msg);       \
            return ASD_ERR_ARG;                           \
        }                                                 \
    } while (0)

#define ASD_LAUNCH_CHECK()                                                         \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            asd_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return ASD_ERR_LAUNCH;                                                 \
        }                                                                          \
    } while (0)

static inline int asd_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) must be issued once per kernel AND device (a process may drive several GPUs): one bit
// per device in a per-call-site mask; the call is idempotent, so two threads racing here both set it and nothing is lost
#include <atomic>
static inline bool asd_attr_needed(std::atomic<unsigned long long>& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_relaxed) & bit) return false;
    mask.fetch_or(bit, std::memory_order_relaxed);
    return true;
}

// Memory-bound grid sizing (guide G11): cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline int asd_grid_for(int64_t n, int block) {
    int g = asd_div_up(n, block);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return g;
}

#ifdef __HIPCC__
// ---- wave64 primitives -------------------------------------------------------------------------
__device__ __forceinline__ int asd_lane() { return (int)(threadIdx.x & 63); }

// sum over the 64 lanes, returned in every lane.  DPP row shifts / row broadcasts (VALU only; the __shfl_xor butterfly is six
// ds_bpermute round trips through the LDS pipe per call, and the field / background backward kernels make hundreds of calls per wave)
__device__ __forceinline__ float asd_wave_sum(float v) {
    int x;
#define ASD_DPP_ADD(CTRL, ROW_MASK)                                                                                      \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);                                    \
    v += __int_as_float(x)
    ASD_DPP_ADD(0x111, 0xf);   // row_shr:1
    ASD_DPP_ADD(0x112, 0xf);   // row_shr:2
    ASD_DPP_ADD(0x114, 0xf);   // row_shr:4
    ASD_DPP_ADD(0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds the row sum
    ASD_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    ASD_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef ASD_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum over each row of 16 lanes, valid in lane 15 of the row (four DPP row shifts, VALU only)
__device__ __forceinline__ float asd_row_sum15(float v) {
    int x;
#define ASD_DPP_ADD(CTRL)                                                                                                \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);                                         \
    v += __int_as_float(x)
    ASD_DPP_ADD(0x111);   // row_shr:1
    ASD_DPP_ADD(0x112);   // row_shr:2
    ASD_DPP_ADD(0x114);   // row_shr:4
    ASD_DPP_ADD(0x118);   // row_shr:8
#undef ASD_DPP_ADD
    return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ float asd_wave_incl_scan(float v) {
    const int lane = asd_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float asd_wave_incl_prod(float v) {
    const int lane = asd_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v *= u;
    }
    return v;
}
// number of set bits of a wave ballot strictly below this lane
__device__ __forceinline__ int asd_ballot_rank(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ float asd_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float asd_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float asd_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- hash grid ---------------------------------------------------------------------------------
// Table index of integer corner (cx,cy,cz).  Branch-free so that the 8 gathers of a level stay in one
// basic block: both the dense and the hashed index are formed and selected with a scalar condition.
// Inputs are clamped to [0,1] by the callers (asd_unit), so a dense index is < 2*size and one
// conditional subtract implements tcnn's `% hashmap_size`.
__device__ __forceinline__ uint32_t asd_grid_index(const asd_grid_meta& m, int l, uint32_t cx, uint32_t cy,
                                                   uint32_t cz) {
    const uint32_t res = m.resolution[l];
    const uint32_t size = m.size[l];
    uint32_t di = cx + cy * res + cz * res * res;
    di -= (di >= size) ? size : 0u;
    const uint32_t hi = ((cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u)) & (size - 1u);
    return m.dense[l] ? di : hi;
}
// out-of-range rule of this implementation: clamp to the unit cube (tcnn wraps through an unsigned cast)
__device__ __forceinline__ float asd_unit(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// Gather-interpolate all levels of one point. enc[2L] stays in VGPRs (L is a compile-time constant).
template <int L>
__device__ __forceinline__ void asd_encode(const asd_grid_meta& m, const float* __restrict__ params, float x,
                                           float y, float z, float (&enc)[2 * L]) {
    x = asd_unit(x); y = asd_unit(y); z = asd_unit(z);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float s = m.scale[l];
        const float px = fmaf(s, x, 0.5f), py = fmaf(s, y, 0.5f), pz = fmaf(s, z, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
        const float wx = px - fx, wy = py - fy, wz = pz - fz;
        const float2* __restrict__ tab = reinterpret_cast<const float2*>(params) + m.offset[l];
        float2 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)  // issue the 8 independent 8-byte gathers first
            v[c] = tab[asd_grid_index(m, l, cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1))];
        float f0 = 0.f, f1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float ax = (c & 1) ? wx : 1.f - wx;
            const float ay = (c & 2) ? wy : 1.f - wy;
            const float az = (c & 4) ? wz : 1.f - wz;
            const float wt = ax * ay * az;
            f0 = fmaf(wt, v[c].x, f0);
            f1 = fmaf(wt, v[c].y, f1);
        }
        enc[2 * l] = f0;
        enc[2 * l + 1] = f1;
        // keep at most 4 levels (32 gathers, 64 VGPRs) in flight: without this fence hipcc hoists all
        // 128 gathers of the 16 levels to the top and spills (256 VGPRs, occupancy 1)
        if ((l & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// Scatter-add d_enc into the table gradient (fp32 hardware atomics; build with -munsafe-fp-atomics).
template <int L>
__device__ __forceinline__ void asd_scatter(const asd_grid_meta& m, float* __restrict__ dparams, float x, float y,
                                            float z, const float (&denc)[2 * L]) {
    x = asd_unit(x); y = asd_unit(y); z = asd_unit(z);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float g0 = denc[2 * l], g1 = denc[2 * l + 1];
        if (g0 == 0.f && g1 == 0.f) continue;
        const float s = m.scale[l];
        const float px = fmaf(s, x, 0.5f), py = fmaf(s, y, 0.5f), pz = fmaf(s, z, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
        const float wx = px - fx, wy = py - fy, wz = pz - fz;
        float* __restrict__ tab = dparams + 2u * (size_t)m.offset[l];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float ax = (c & 1) ? wx : 1.f - wx;
            const float ay = (c & 2) ? wy : 1.f - wy;
            const float az = (c & 4) ? wz : 1.f - wz;
            const float wt = ax * ay * az;
            const uint32_t idx = asd_grid_index(m, l, cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1));
            atomicAdd(tab + 2u * (size_t)idx, wt * g0);
            atomicAdd(tab + 2u * (size_t)idx + 1, wt * g1);
        }
    }
}
// Scatter with wave-level run aggregation and request coalescing.
//
// What gfx950 rate-limits is not the fp32 atomic but the REQUEST: the lanes of one atomic instruction that fall into the same
// 64-byte block travel as one request, and the chip retires ~21 G requests/s whatever their width, scope or table size
// (tools/atomic_probe2.hip: 1 dword per block 21 G dword-atomics/s, 2 -> 42, 4 -> 84, 16 -> 311).  A table entry is two adjacent
// floats, and the x-neighbour corner is the adjacent entry on the dense levels and — when cx is even — on the hashed ones
// ((cx ^ h) and ((cx + 1) ^ h) differ in bit 0 only).  So:
//   * levels >= NAGG (no sharing between neighbouring samples): the work of a wave is transposed — lanes 4k..4k+3 of one
//     instruction carry (corner x0: f0, f1 | corner x1: f0, f1) of ONE source sample, 16 samples per instruction — so the 16
//     dword updates of a sample and level travel in 4..8 requests instead of 16;
//   * levels < NAGG: samples are packed ray-major / t-ascending, consecutive lanes fall into the same cell (run length ~ cell
//     size / march step: ~18 on level 0, ~1.4 on level 7).  The per-corner contributions of a run are first summed across its
//     lanes (segmented shuffle reduction on the run id); the run head then issues f0 while the run's second lane issues the f1 of
//     the same entry in the same instruction (one request); single-lane runs fall back to a second instruction.
// Must be called by all 64 lanes (inactive lanes pass active=false).
//   * levels < NPRIV (the coarsest, a few thousand entries that EVERY sample of the step updates: ~1 % of the atomics but ~10 % of
//     the kernel's time — same-line serialisation, tools/field_bwd_ab.py): each XCD adds into its own copy `priv + xcd * priv_stride`
//     (ASD_PRIV_COPIES copies, that many times fewer collisions per line); asd_priv_reduce_kernel folds the copies into dparams afterwards.
#ifndef ASD_PRIV_COPIES
#define ASD_PRIV_COPIES 8
#endif
template <int L, int NAGG, int NPRIV = 0, bool FINE = true>
__device__ __forceinline__ void asd_scatter_runs(const asd_grid_meta& m, float* __restrict__ dparams, float x, float y,
                                                 float z, const float (&denc)[2 * L], bool active, float* __restrict__ priv = nullptr,
                                                 uint32_t priv_stride = 0) {
    static_assert(NPRIV <= NAGG, "only run-aggregated levels have per-XCD copies");
    x = asd_unit(x); y = asd_unit(y); z = asd_unit(z);
    const int lane = asd_lane();
    if (NPRIV > 0) {
        unsigned xcc, copy;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        copy = xcc & 7u;
        if (ASD_PRIV_COPIES > 8) {   // more copies: the XCD's CUs are split by the low bits of their id (HW_ID[11:8])
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            copy = copy * (ASD_PRIV_COPIES / 8) + ((hwid >> 8) & (ASD_PRIV_COPIES / 8 - 1));
        }
        priv += (size_t)copy * priv_stride;
    }
#pragma unroll
    for (int l = 0; l < (NAGG < L ? NAGG : L); ++l) {
        const float g0 = active ? denc[2 * l] : 0.f, g1 = active ? denc[2 * l + 1] : 0.f;
        const float s = m.scale[l];
        const float px = fmaf(s, x, 0.5f), py = fmaf(s, y, 0.5f), pz = fmaf(s, z, 0.5f);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        const uint32_t cx = (uint32_t)(int32_t)fx, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
        const float wx = px - fx, wy = py - fy, wz = pz - fz;
        float* __restrict__ tab = (l < NPRIV ? priv : dparams) + 2u * (size_t)m.offset[l];
        const uint32_t res = m.resolution[l];
        const uint32_t key = active ? cx + (cy + cz * res) * res : 0xFFFFFFFFu - (uint32_t)lane;
        const uint32_t prev = __shfl_up(key, 1, 64);
        const bool head = lane == 0 || prev != key;
        const unsigned long long hm = __ballot(head);
        const int rid = __popcll(hm & (~0ull >> (63 - lane)));   // heads at or below this lane = run id
        float v[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wt = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
            v[2 * c] = wt * g0;
            v[2 * c + 1] = wt * g1;
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int prid = __shfl_down(rid, o, 64);
            const bool ok = (lane + o < 64) && prid == rid;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float u = __shfl_down(v[q], o, 64);
                v[q] += ok ? u : 0.f;
            }
        }
        // a run's second lane (not a head, its predecessor is) carries the head's f1; heads without such a lane issue it themselves
        const bool issue = head && active;
        const bool carrier = !head && ((hm >> (lane - 1)) & 1ull);           // lane >= 1 here (lane 0 is always a head)
        const bool self_f1 = issue && (lane == 63 || ((hm >> (lane + 1)) & 1ull));
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t idx = asd_grid_index(m, l, cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1));
            const float f1_prev = __shfl_up(v[2 * c + 1], 1, 64);
            // the carrier sits in the head's cell, so it forms the same idx
            const float val = issue ? v[2 * c] : f1_prev;
            float* dst = tab + 2u * (size_t)idx + (issue ? 0 : 1);
            if ((issue || carrier) && val != 0.f) atomicAdd(dst, val);
            if (self_f1 && v[2 * c + 1] != 0.f) atomicAdd(tab + 2u * (size_t)idx + 1, v[2 * c + 1]);
        }
    }
    if (NAGG >= L || !FINE) return;      // FINE = false: the levels >= NAGG go through the paged scatter (field_paged.hip)
    // ---- fine levels, transposed: lane = (source sample k = lane >> 2 of group q, x corner (lane >> 1) & 1, feature lane & 1)
    const int xb = (lane >> 1) & 1, ft = lane & 1;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int src = 16 * q + (lane >> 2);
        const float sx = __shfl(x, src, 64), sy = __shfl(y, src, 64), sz = __shfl(z, src, 64);
        const bool sact = __shfl((int)active, src, 64) != 0;
#pragma unroll
        for (int l = (NAGG < L ? NAGG : L); l < L; ++l) {
            const float ga = __shfl(denc[2 * l], src, 64), gb = __shfl(denc[2 * l + 1], src, 64);
            const float g = ft ? gb : ga;
            if (!sact || g == 0.f) continue;
            const float s = m.scale[l];
            const float px = fmaf(s, sx, 0.5f), py = fmaf(s, sy, 0.5f), pz = fmaf(s, sz, 0.5f);
            const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
            const uint32_t cx = (uint32_t)(int32_t)fx + (uint32_t)xb, cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
            const float wx = px - fx, wy = py - fy, wz = pz - fz;
            const float ax = xb ? wx : 1.f - wx;
            float* __restrict__ tab = dparams + 2u * (size_t)m.offset[l] + ft;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // same product order as the one-lane-per-sample form: (ax * ay) * az
                const float wt = ax * ((c & 1) ? wy : 1.f - wy) * ((c & 2) ? wz : 1.f - wz);
                const uint32_t idx = asd_grid_index(m, l, cx, cy + (c & 1), cz + ((c >> 1) & 1));
                atomicAdd(tab + 2u * (size_t)idx, wt * g);
            }
        }
    }
}

// measurement hook (asd_probe_events): events recorded around an entry point's dominant kernel
extern hipEvent_t g_asd_probe_start, g_asd_probe_stop;
#define ASD_PROBE_START(s) do { if (g_asd_probe_start) (void)hipEventRecord(g_asd_probe_start, (s)); } while (0)
#define ASD_PROBE_STOP(s) do { if (g_asd_probe_stop) (void)hipEventRecord(g_asd_probe_stop, (s)); } while (0)

#endif  // __HIPCC__
