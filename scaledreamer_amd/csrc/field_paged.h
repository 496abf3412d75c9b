// field_paged.h — paged scatter of the hash-grid gradient's fine levels (internal to libasd_hip.so; kernels in field_paged.hip).
//
// tcnn kernel_grid_backward (reference call site threestudio/models/networks.py:55-64) adds every sample's 8 corner contributions per
// level into the table gradient with global atomics.  On gfx950 a float atomic leaves the L2 as one fabric request whatever its scope
// (~21 G requests/s for the chip, tools/atomic_probe2.hip), and on the hashed levels neighbouring samples share no entry, so nothing
// can be aggregated in the wave: 433 k samples x 10 levels x 4..8 requests = 1.1 ms per headline step at 2.3x the algorithmic traffic.
// Here the fine levels never touch a global atomic:
//   * a hashed index is (cx ^ cy * P1 ^ cz * P2) & (size - 1) with cx <= resolution < 2^13, so bits >= 13 — the 8192-entry PAGE — depend
//     on (cy, cz) only: the two x corners of a (dy, dz) pair always land in the same 64 KB page of their level;
//   * pass 1 (pg_fill) turns every (row, level, dy, dz) into a 16-byte ITEM {both in-page entries, g0 w_yz, g1 w_yz, w_x} and appends it
//     to the list of its page (levels x pages bins of fixed capacity, block-local reservations; an overflowing item is added with global
//     atomics on the spot);
//   * pass 2 (pg_accum) gives every page to ONE workgroup that sums its items into a 64 KB LDS image — plain read-modify-writes under
//     a tag arbitration, LDS float atomics being slower than the global ones they would replace — and adds the image to the table
//     gradient with plain 16-byte read-modify-writes — the only HBM traffic besides the streamed items.
#pragma once
#include "asd_common.h"

#define ASD_PG_SHIFT 13                    // log2(entries per page): 8192 entries x 2 floats = 64 KB of LDS
#define ASD_PG_ENTRIES (1 << ASD_PG_SHIFT)
#define ASD_PG_MAX_BINS 1024               // levels x pages per level (one block scans them)
#ifndef ASD_PG_NF
#define ASD_PG_NF 11                       // hashed levels of the 16-level grid (levels >= ASD_FIELD_NAGG = 5: all of them have 2^19 entries)
#endif
#define ASD_PG_NF_PAD ((ASD_PG_NF + 1) / 2 * 2)   // levels per row of `g` (24 floats: 16-byte aligned rows; the pad pair is zero)
#define ASD_PG_CHUNK_ROWS (2 << 20)        // rows binned per pass (item slots: 800 B per row)

struct asd_paged_plan {
    int first_level, n_levels;             // fine levels [first_level, first_level + n_levels)
    int pages_per_level, bins;
    uint32_t mask;                         // size - 1 of a fine level
};

// 1 when every level >= first_level is hashed, has the same power-of-two size >= one page, and a resolution below 2^ASD_PG_SHIFT
int asd_paged_plan_init(const asd_grid_meta* m, int first_level, asd_paged_plan* plan);
// floats behind the row buffers: items of one chunk + counters
int64_t asd_paged_workspace_floats(int64_t rows);
// upos [rows, 3] unit-cube positions, g [rows, 2 * ASD_PG_NF_PAD] gradients w.r.t. the fine levels' features; row = pt * n + i is live iff
// i < min(*n_dev, n).  d_grid += the fine levels' scatter.  ws: asd_paged_workspace_floats(rows) floats.
int asd_paged_scatter(const asd_grid_meta* m, const asd_paged_plan* plan, const float* upos, const float* g, int32_t n, int32_t n_pts,
                      const int32_t* n_dev, float* d_grid, float* ws, hipStream_t s);
