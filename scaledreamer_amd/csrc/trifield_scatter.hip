// trifield_scatter.hip — the tri-plane field's feature gradient onto the planes (backward of sample_from_planes,
// custom/amortized/models/geometry/utils.py:81-93) for the ROWS the fused field leaves (denc [R][96], points [R][3] in grid_sample coordinates).
//
// A tri-plane is 3 x 64 x 64 texels of 32 channels: every texel receives thousands of contributions per step, so the scatter is bound by fp32
// atomics on a few thousand hot addresses (asd_triplane_sample_bwd: 0.64 ms per 1 M rows even when consecutive rows of one cell are summed in
// registers first).  Here the rows are SORTED by cell first — a counting sort per plane over (H + 1)(W + 1) cells with block-private LDS
// histograms: three light passes over the 12-byte points — and a wave then walks 256 consecutive entries of one plane's sorted list: the rows of
// a cell arrive together, are summed in registers (lane = channel x tap row), and leave as ONE atomic per tap and channel when the cell changes.
// The pass reads every gradient row once (128 B per row and plane, whole cache lines) and issues ~1 atomic per 50-250 rows.
// Within a cell the order of the rows is the order the fill pass's atomics resolved in: sums are reordered run to run exactly as with atomics.
#include "trifield_common.h"
#include "trifield_mfma.h"

#define TS_SEG 256             // sorted entries per wave of the reduction

__device__ __forceinline__ int ts_cell(const float* __restrict__ pts, size_t row, int plane, int H, int W, int& x0, int& y0, float& fx, float& fy) {
    float u, v;
    tf_plane_uv(pts[3 * row], pts[3 * row + 1], pts[3 * row + 2], plane, u, v);
    tf_axis(u, W, x0, fx); tf_axis(v, H, y0, fy);
    if (x0 < -1 || x0 >= W || y0 < -1 || y0 >= H) return -1;             // no tap inside the plane: the row does not contribute
    return (y0 + 1) * (W + 1) + (x0 + 1);
}

// pass 1: cnt[plane][cell] += rows of the block's slab (LDS histogram, one global atomic per touched cell and block)
__global__ __launch_bounds__(256) void ts_hist_kernel(const float* __restrict__ pts, int R, int H, int W, int* __restrict__ cnt) {
    extern __shared__ int hist[];                   // [3][cells]
    const int cells = (H + 1) * (W + 1);
    for (int q = threadIdx.x; q < 3 * cells; q += 256) hist[q] = 0;
    __syncthreads();
    const int per = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            int x0, y0;
            float fx, fy;
            const int key = ts_cell(pts, r, pl, H, W, x0, y0, fx, fy);
            if (key >= 0) atomicAdd(&hist[pl * cells + key], 1);
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 3 * cells; q += 256)
        if (hist[q]) atomicAdd(&cnt[q], hist[q]);
}

// pass 2: off[plane][cell] = exclusive prefix of cnt within the plane; cursor = off; len[plane] = total (one block per plane)
__global__ __launch_bounds__(1024) void ts_scan_kernel(const int* __restrict__ cnt, int cells, int* __restrict__ off, int* __restrict__ cursor, int* __restrict__ len) {
    __shared__ int part[1024];
    const int pl = blockIdx.x, tid = threadIdx.x;
    const int per = (cells + 1023) / 1024, q0 = tid * per, q1 = min(cells, q0 + per);
    int s = 0;
    for (int q = q0; q < q1; ++q) s += cnt[pl * cells + q];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {            // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    if (tid == 1023) len[pl] = part[1023];
    for (int q = q0; q < q1; ++q) { off[pl * cells + q] = run; cursor[pl * cells + q] = run; run += cnt[pl * cells + q]; }
}

// pass 3: sorted[plane][position] = row; a block reserves, per touched cell, a range for its slab's rows, then ranks them in LDS
__global__ __launch_bounds__(256) void ts_fill_kernel(const float* __restrict__ pts, int R, int H, int W, int* __restrict__ cursor, int* __restrict__ sorted) {
    extern __shared__ int lds[];                    // hist [3][cells] | base [3][cells]
    const int cells = (H + 1) * (W + 1);
    int* hist = lds;
    int* base = lds + 3 * cells;
    for (int q = threadIdx.x; q < 3 * cells; q += 256) hist[q] = 0;
    __syncthreads();
    const int per = (R + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(R, r0 + per);
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            int x0, y0;
            float fx, fy;
            const int key = ts_cell(pts, r, pl, H, W, x0, y0, fx, fy);
            if (key >= 0) atomicAdd(&hist[pl * cells + key], 1);
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 3 * cells; q += 256) {
        const int c = hist[q];
        base[q] = c ? atomicAdd(&cursor[q], c) : 0;
        hist[q] = 0;
    }
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            int x0, y0;
            float fx, fy;
            const int key = ts_cell(pts, r, pl, H, W, x0, y0, fx, fy);
            if (key >= 0) sorted[(size_t)pl * R + base[pl * cells + key] + atomicAdd(&hist[pl * cells + key], 1)] = r;
        }
    }
}

// pass 4: a wave per TS_SEG sorted entries of one plane; lane = (channel c = l & 31, tap row t = l >> 5)
__global__ __launch_bounds__(256) void ts_reduce_kernel(const float* __restrict__ denc, const float* __restrict__ pts, const int* __restrict__ sorted,
                                                        const int* __restrict__ len, int R, int H, int W, float* __restrict__ d_planes) {
    const int lane = threadIdx.x & 63, c = lane & 31, t = lane >> 5;
    const int nseg = (R + TS_SEG - 1) / TS_SEG;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= 3 * nseg) return;
    const int pl = w / nseg, seg = w - pl * nseg;
    const int start = seg * TS_SEG, end = min(len[pl], start + TS_SEG);
    if (start >= end) return;
    float a0 = 0.f, a1 = 0.f;                       // taps (x0, y0 + t) and (x0 + 1, y0 + t) of the current cell
    int cur = -1, cx = 0, cy = 0;
    auto flush = [&]() {
        if (cur < 0) return;
        const int y = cy + t;
        if (y >= 0 && y < H) {
            float* dst = d_planes + ((size_t)(pl * H + y) * W) * 32 + c;
            if (cx >= 0 && a0 != 0.f) atomicAdd(dst + (size_t)cx * 32, a0);
            if (cx + 1 < W && a1 != 0.f) atomicAdd(dst + (size_t)(cx + 1) * 32, a1);
        }
        a0 = 0.f; a1 = 0.f;
    };
    for (int b0 = start; b0 < end; b0 += 64) {
        // lane i: entry b0 + i of the list
        int row = 0, key = -1, x0 = 0, y0 = 0;
        float fx = 0.f, fy = 0.f;
        if (b0 + lane < end) {
            row = sorted[(size_t)pl * R + b0 + lane];
            key = ts_cell(pts, row, pl, H, W, x0, y0, fx, fy);
        }
        const int m = min(64, end - b0);
        for (int j0 = 0; j0 < m; j0 += 8) {
            float g8[8];                                // the gradient loads of eight rows fly together (a row's address comes out of a readlane)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rj = __builtin_amdgcn_readlane(row, (j0 + k) & 63);
                g8[k] = j0 + k < m ? denc[(size_t)rj * TF_NIN + pl * 32 + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + k;
                if (j >= m) break;                      // wave-uniform
                const int kj = __builtin_amdgcn_readlane(key, j);
                const float fxj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fx), j));
                const float fyj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fy), j));
                if (kj != cur) {                        // wave-uniform
                    flush();
                    cur = kj;
                    cx = __builtin_amdgcn_readlane(x0, j); cy = __builtin_amdgcn_readlane(y0, j);
                }
                const float wy = t ? fyj : 1.f - fyj;
                a0 = fmaf((1.f - fxj) * wy, g8[k], a0);
                a1 = fmaf(fxj * wy, g8[k], a1);
            }
        }
    }
    flush();
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
int64_t tfs_work_ints(int rows, int H, int W) { return (int64_t)3 * rows + (int64_t)9 * (H + 1) * (W + 1) + 64; }
bool tfs_supported(int H, int W) { return (size_t)6 * (H + 1) * (W + 1) * sizeof(int) <= 150 * 1024; }

int tfs_scatter(const float* denc, const float* pts, int R, int H, int W, float* d_planes, int* work, hipStream_t s) {
    const int cells = (H + 1) * (W + 1);
    int* cnt = work;                                // [3][cells]
    int* off = cnt + 3 * cells;
    int* cursor = off + 3 * cells;
    int* len = cursor + 3 * cells;                  // [3] (+ pad)
    int* sorted = len + 64;                         // [3][R]
    static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)ts_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)ts_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr = true;
    }
    (void)hipMemsetAsync(cnt, 0, (size_t)3 * cells * sizeof(int), s);
    int blocks = asd_div_up(R, 2048);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ts_hist_kernel, dim3(blocks), dim3(256), (size_t)3 * cells * sizeof(int), s, pts, R, H, W, cnt);
    hipLaunchKernelGGL(ts_scan_kernel, dim3(3), dim3(1024), 0, s, cnt, cells, off, cursor, len);
    hipLaunchKernelGGL(ts_fill_kernel, dim3(blocks), dim3(256), (size_t)6 * cells * sizeof(int), s, pts, R, H, W, cursor, sorted);
    const int waves = 3 * asd_div_up(R, TS_SEG);
    hipLaunchKernelGGL(ts_reduce_kernel, dim3(asd_div_up(waves, 4)), dim3(256), 0, s, denc, pts, sorted, len, R, H, W, d_planes);
    return ASD_OK;
}
