// field_paged.hip — the hash-grid gradient's fine levels without global atomics: items binned by 64 KB table page, one workgroup per
// page accumulating in LDS (design: field_paged.h).  Replaces the levels >= ASD_FIELD_NAGG part of asd_scatter_runs for the training
// backward of `implicit-volume` / `Hyper-iNGP` (tcnn kernel_grid_backward semantics, reference call site threestudio/models/networks.py:55-64).
//
// Roofline: HBM.  Algorithmic bytes per row and fine level: 8 corner entries x 8 B read-modify-write = 64 B if every update went to HBM
// once (SURVEY.md section 8d counts 1024 B per 16-level encode); this path streams 4 items x 16 B twice (write + read) = 128 B per row and
// level through HBM and touches the table itself once per PAGE (64 KB read + write per page and launch).
#include "field_paged.h"

#define PG_P1 2654435761u
#define PG_P2 805459861u

struct pg_level {      // what a row needs of one level
    uint32_t cx, hy0, hy1, hz0, hz1;
    float wx, wy, wz;
};
__device__ __forceinline__ pg_level pg_locate(const asd_grid_meta& m, int l, float x, float y, float z) {
    // the arithmetic of asd_encode (asd_common.h): same cell, same weights
    const float s = m.scale[l];
    const float px = fmaf(s, x, 0.5f), py = fmaf(s, y, 0.5f), pz = fmaf(s, z, 0.5f);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const uint32_t cy = (uint32_t)(int32_t)fy, cz = (uint32_t)(int32_t)fz;
    pg_level r;
    r.cx = (uint32_t)(int32_t)fx;
    r.hy0 = cy * PG_P1; r.hy1 = (cy + 1u) * PG_P1;
    r.hz0 = cz * PG_P2; r.hz1 = (cz + 1u) * PG_P2;
    r.wx = px - fx; r.wy = py - fy; r.wz = pz - fz;
    return r;
}
__device__ __forceinline__ uint32_t pg_hash_yz(const pg_level& r, int combo) {
    return ((combo & 1) ? r.hy1 : r.hy0) ^ ((combo & 2) ? r.hz1 : r.hz0);
}

// row rr = pt * n + i of the chunk is live iff i < nn
__device__ __forceinline__ bool pg_row_live(int64_t rr, int64_t rows, int n, int nn) {
    return rr < rows && (int)(rr % n) < nn;
}

// ---- pass 1: the items.  Every bin owns a fixed range of `stride` slots (1.25x its share of the rows' items + slack: the hash spreads
// (cy, cz) pairs evenly over a level's pages, a 4 M-row chunk fills a bin to 80 % +- 0.1 %).  A block takes ONE rank per item from an LDS
// histogram (returning LDS atomic; ranks parked in registers), reserves one contiguous range per touched bin with one global atomic, and
// writes the items.  An item that finds its bin full — rows crowded into a few cells — is added with global atomics right here. ------
template <int NF>
__global__ __launch_bounds__(256) void pg_fill_kernel(const asd_grid_meta m, const asd_paged_plan plan, const float* __restrict__ upos,
                                                      const float* __restrict__ g, int64_t row0, int64_t rows, int n,
                                                      const int* __restrict__ n_dev, uint32_t stride, uint32_t* __restrict__ cursor,
                                                      uint4* __restrict__ items, float* __restrict__ d_grid) {
    __shared__ uint32_t hist[ASD_PG_MAX_BINS];
    const int nn = n_dev ? min(*n_dev, n) : n;
    const int64_t rr = row0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = pg_row_live(rr, rows, n, nn);
    if (__syncthreads_count(live) == 0) return;
    for (int q = threadIdx.x; q < plan.bins; q += 256) hist[q] = 0u;
    __syncthreads();
    float x = 0.f, y = 0.f, z = 0.f;
    constexpr int NFP = (NF + 1) / 2 * 2;       // levels per row of g (= ASD_PG_NF_PAD)
    float gv[2 * NFP];
    uint32_t rk[2 * NF];            // ranks inside the block's range of a bin: two 16-bit ranks per word (<= 1024 items per block and bin)
#pragma unroll
    for (int q = 0; q < 2 * NFP; ++q) gv[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 2 * NF; ++q) rk[q] = 0u;
    if (live) {
        x = upos[3 * rr]; y = upos[3 * rr + 1]; z = upos[3 * rr + 2];
        const float4* src = reinterpret_cast<const float4*>(g + rr * (2 * NFP));
#pragma unroll
        for (int q = 0; q < NFP / 2; ++q) {
            const float4 v = src[q];
            gv[4 * q] = v.x; gv[4 * q + 1] = v.y; gv[4 * q + 2] = v.z; gv[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (gv[2 * f] == 0.f && gv[2 * f + 1] == 0.f) continue;
            const pg_level r = pg_locate(m, plan.first_level + f, x, y, z);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t rank = atomicAdd(&hist[f * plan.pages_per_level + ((pg_hash_yz(r, c) & plan.mask) >> ASD_PG_SHIFT)], 1u);
                rk[2 * f + (c >> 1)] |= rank << (16 * (c & 1));
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < plan.bins; q += 256) {
        const uint32_t c = hist[q];
        hist[q] = c ? atomicAdd(&cursor[q], c) : 0u;         // from here on: the block's first slot in bin q
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const float g0 = gv[2 * f], g1 = gv[2 * f + 1];
        if (g0 == 0.f && g1 == 0.f) continue;
        const int l = plan.first_level + f;
        const pg_level r = pg_locate(m, l, x, y, z);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t h = pg_hash_yz(r, c);
            const uint32_t pgi = (h & plan.mask) >> ASD_PG_SHIFT;
            const int bin = f * plan.pages_per_level + (int)pgi;
            const uint32_t slot = hist[bin] + ((rk[2 * f + (c >> 1)] >> (16 * (c & 1))) & 0xFFFFu);
            const float wyz = ((c & 1) ? r.wy : 1.f - r.wy) * ((c & 2) ? r.wz : 1.f - r.wz);
            const uint32_t e0 = (r.cx ^ h) & (ASD_PG_ENTRIES - 1), e1 = ((r.cx + 1u) ^ h) & (ASD_PG_ENTRIES - 1);
            if (slot < stride) {
                items[(size_t)bin * stride + slot] = make_uint4(e0 | (e1 << 16), __float_as_uint(wyz * g0), __float_as_uint(wyz * g1), __float_as_uint(r.wx));
            } else {
                float* tab = d_grid + 2u * ((size_t)m.offset[l] + (size_t)pgi * ASD_PG_ENTRIES);
                const float a = 1.f - r.wx;
                atomicAdd(tab + 2u * e0, a * (wyz * g0)); atomicAdd(tab + 2u * e0 + 1, a * (wyz * g1));
                atomicAdd(tab + 2u * e1, r.wx * (wyz * g0)); atomicAdd(tab + 2u * e1 + 1, r.wx * (wyz * g1));
            }
        }
    }
}

// ---- pass 2: one workgroup per page ------------------------------------------------------------------------------------------------------
// LDS float atomics retire ~0.3 lanes per clock and CU on gfx950 (the first form of this kernel, 4 ds_add_f32 per item: 543 us for the 22.5 M
// items of a 562 k-sample step — more than the global atomics it replaced), so the image is updated with PLAIN 8-byte read-modify-writes
// under a tag arbitration: every thread announces the two entries of its item in a 16-bit tag image (tag[e] = thread id), the workgroup
// meets at ONE barrier, and whoever reads its own id back owns that entry for the round and adds its pair; losers (several items of the
// round on one entry) keep the half they lost for the next round (8192 announcements into 8192 entries: 63 % win the first round, the
// rest within three more).  One barrier per round is enough:
// an announcement of round r + 1 that overtakes a slow wave's read of round r can only turn a winner into a loser (nobody adds, both
// retry), and nobody reads a tag of round r + 1 before every wave has finished its adds of round r behind that round's barrier.
#define PG_CHUNK 4       // items per thread and chunk: 8 announcements per thread and round
#define PG_ROUNDS 4      // 8192 announcements into 8192 entries: 3014, ~500, ~15, ~0 left after rounds 1..4
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void pg_accum_kernel(const asd_grid_meta m, const asd_paged_plan plan, const uint4* __restrict__ items,
                                                        uint32_t stride, const uint32_t* __restrict__ cursor, float* __restrict__ d_grid) {
    __shared__ __attribute__((aligned(16))) float2 page[ASD_PG_ENTRIES];     // 64 KB
    __shared__ uint16_t tag[ASD_PG_ENTRIES];                                 // 16 KB
    const int bin = blockIdx.x, tid = threadIdx.x;
    const uint32_t cnt = min(cursor[bin], stride);
    if (cnt == 0u) return;
    for (int q = tid; q < ASD_PG_ENTRIES / 2; q += 1024) reinterpret_cast<float4*>(page)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint4* src = items + (size_t)bin * stride;
    // a chunk = PG_CHUNK items per thread, all announced together; the next chunk's loads are in flight while this one is resolved (one
    // item per thread and round left every round waiting ~2 us for a load issued one round earlier: 335 us for 22.5 M items)
    uint4 nxt[PG_CHUNK];
    uint32_t k = tid;
#pragma unroll
    for (int j = 0; j < PG_CHUNK; ++j) nxt[j] = k + j * 1024u < cnt ? src[k + j * 1024u] : make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (uint32_t c0 = 0; c0 < cnt; c0 += PG_CHUNK * 1024u) {
        uint32_t ee[PG_CHUNK];
        float v0[PG_CHUNK], v1[PG_CHUNK], wx[PG_CHUNK];
        uint32_t pend = 0u;
#pragma unroll
        for (int j = 0; j < PG_CHUNK; ++j) {
            ee[j] = nxt[j].x; v0[j] = __uint_as_float(nxt[j].y); v1[j] = __uint_as_float(nxt[j].z); wx[j] = __uint_as_float(nxt[j].w);
            if (c0 + k + j * 1024u < cnt) pend |= 3u << (2 * j);
        }
        const uint32_t kn = c0 + PG_CHUNK * 1024u + k;
#pragma unroll
        for (int j = 0; j < PG_CHUNK; ++j) nxt[j] = kn + j * 1024u < cnt ? src[kn + j * 1024u] : make_uint4(0u, 0u, 0u, 0u);
        // PG_ROUNDS rounds, then whatever is still pending (a hot entry: many items of one chunk on it) goes in with LDS atomics.  A fixed
        // count instead of a workgroup-wide "anything left?" keeps the kernel at exactly 80 KB of LDS: two workgroups per CU.
#pragma unroll 1
        for (int round = 0; round < PG_ROUNDS; ++round) {
#pragma unroll
            for (int j = 0; j < PG_CHUNK; ++j) {
                if (pend & (1u << (2 * j))) tag[ee[j] & 0xFFFFu] = (uint16_t)tid;
                if (pend & (2u << (2 * j))) tag[ee[j] >> 16] = (uint16_t)tid;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PG_CHUNK; ++j) {
                const uint32_t e0 = ee[j] & 0xFFFFu, e1 = ee[j] >> 16;
                if ((pend & (1u << (2 * j))) && tag[e0] == (uint16_t)tid) {
                    const float a = 1.f - wx[j];
                    float2 t = page[e0];
                    t.x = fmaf(a, v0[j], t.x); t.y = fmaf(a, v1[j], t.y);
                    page[e0] = t;
                    pend &= ~(1u << (2 * j));
                }
                if ((pend & (2u << (2 * j))) && tag[e1] == (uint16_t)tid) {
                    float2 t = page[e1];
                    t.x = fmaf(wx[j], v0[j], t.x); t.y = fmaf(wx[j], v1[j], t.y);
                    page[e1] = t;
                    pend &= ~(2u << (2 * j));
                }
            }
        }
        __syncthreads();            // every plain add of the last round has landed
        if (pend) {
#pragma unroll
            for (int j = 0; j < PG_CHUNK; ++j) {
                const uint32_t e0 = ee[j] & 0xFFFFu, e1 = ee[j] >> 16;
                if (pend & (1u << (2 * j))) { atomicAdd(&page[e0].x, (1.f - wx[j]) * v0[j]); atomicAdd(&page[e0].y, (1.f - wx[j]) * v1[j]); }
                if (pend & (2u << (2 * j))) { atomicAdd(&page[e1].x, wx[j] * v0[j]); atomicAdd(&page[e1].y, wx[j] * v1[j]); }
            }
        }
        // (the next chunk's first adds sit behind its first barrier, i.e. behind these atomics)
    }
    __syncthreads();
    const int l = plan.first_level + bin / plan.pages_per_level, pg = bin % plan.pages_per_level;
    float4* dst = reinterpret_cast<float4*>(d_grid + 2u * ((size_t)m.offset[l] + (size_t)pg * ASD_PG_ENTRIES));
    for (int q = tid; q < ASD_PG_ENTRIES / 2; q += 1024) {
        const float4 a = reinterpret_cast<const float4*>(page)[q];
        if (a.x == 0.f && a.y == 0.f && a.z == 0.f && a.w == 0.f) continue;
        float4 d = dst[q];
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
        dst[q] = d;
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------
int asd_paged_plan_init(const asd_grid_meta* m, int first_level, asd_paged_plan* plan) {
    const int L = (int)m->n_levels;
    if (first_level >= L || L - first_level != ASD_PG_NF) return 0;
    const uint32_t size = m->size[first_level];
    if (size < (uint32_t)ASD_PG_ENTRIES || (size & (size - 1u)) != 0u) return 0;
    for (int l = first_level; l < L; ++l)
        if (m->dense[l] || m->size[l] != size || m->resolution[l] >= (uint32_t)ASD_PG_ENTRIES) return 0;
    plan->first_level = first_level;
    plan->n_levels = L - first_level;
    plan->pages_per_level = (int)(size >> ASD_PG_SHIFT);
    plan->bins = plan->n_levels * plan->pages_per_level;
    plan->mask = size - 1u;
    return plan->bins <= ASD_PG_MAX_BINS;
}

static uint32_t pg_stride(int64_t chunk_rows, int pages_per_level) {
    // slots per bin: 1.25x the bin's share of 4 items per row and level, + slack for small launches; a multiple of 4 items
    const int64_t share = (chunk_rows * 4 + pages_per_level - 1) / pages_per_level;
    return (uint32_t)((share + share / 4 + 1024 + 3) & ~(int64_t)3);
}

int64_t asd_paged_workspace_floats(int64_t rows) {
    const int64_t chunk = rows < ASD_PG_CHUNK_ROWS ? rows : ASD_PG_CHUNK_ROWS;
    // sized for the 2^19-entry levels of the shipped grids (64 pages per level); asd_paged_scatter re-derives the stride from the plan
    // and falls back to fewer rows per pass if a grid with smaller tables needs more slots than this
    return (int64_t)ASD_PG_NF * 64 * pg_stride(chunk, 64) * 4 + (ASD_PG_MAX_BINS + 32) + 16;
}

int asd_paged_scatter(const asd_grid_meta* m, const asd_paged_plan* plan, const float* upos, const float* g, int32_t n, int32_t n_pts,
                      const int32_t* n_dev, float* d_grid, float* ws, hipStream_t s) {
    const int64_t rows = (int64_t)n * n_pts;
    uint32_t* cursor = reinterpret_cast<uint32_t*>(ws);
    uint4* items = reinterpret_cast<uint4*>((reinterpret_cast<uintptr_t>(cursor + ASD_PG_MAX_BINS + 32) + 15) & ~(uintptr_t)15);
    const int64_t item_cap = (asd_paged_workspace_floats(rows) - (ASD_PG_MAX_BINS + 32) - 16) / 4;
    int64_t chunk_rows = rows < ASD_PG_CHUNK_ROWS ? rows : ASD_PG_CHUNK_ROWS;
    while (chunk_rows > 256 && (int64_t)plan->bins * pg_stride(chunk_rows, plan->pages_per_level) > item_cap) chunk_rows /= 2;
    const uint32_t stride = pg_stride(chunk_rows, plan->pages_per_level);
    if ((int64_t)plan->bins * stride > item_cap) {
        asd_set_error("asd_paged_scatter: workspace too small for this grid");
        return ASD_ERR_ARG;
    }
    for (int64_t row0 = 0; row0 < rows; row0 += chunk_rows) {
        const int64_t end = row0 + chunk_rows < rows ? row0 + chunk_rows : rows;
        if (hipMemsetAsync(cursor, 0, (size_t)(ASD_PG_MAX_BINS + 32) * sizeof(uint32_t), s) != hipSuccess) {
            asd_set_error("asd_paged_scatter: hipMemsetAsync failed");
            return ASD_ERR_LAUNCH;
        }
        const dim3 grid((unsigned)((end - row0 + 255) / 256));
        hipLaunchKernelGGL((pg_fill_kernel<ASD_PG_NF>), grid, dim3(256), 0, s, *m, *plan, upos, g, row0, end, n, n_dev, stride, cursor, items, d_grid);
        hipLaunchKernelGGL(pg_accum_kernel, dim3(plan->bins), dim3(1024), 0, s, *m, *plan, items, stride, cursor, d_grid);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}
