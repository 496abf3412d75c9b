// amortized.hip — kernels of the multi-prompt (amortized) render path on gfx950:
//   * importance resampling of ray intervals, transmittance cdf, sorted merge   (ImportanceEstimator.sampling,
//     threestudio/models/estimators.py:62-101 -> nerfacc.pdf.importance_sampling / volrend.render_transmittance_from_density)
//   * trilinear voxel and tri-plane feature sampling + their scatter backward   (get_trilinear_feature / sample_from_planes,
//     custom/amortized/models/geometry/utils.py:67-110: F.grid_sample bilinear, zeros padding, align_corners=False)
//   * NCDHW <-> channel-last relayout of the generator's feature volume.
// All of it is HBM/L2-bound gather-scatter work: feature vectors are stored channel-LAST so that one corner of one sample
// is one contiguous C*4-byte read (128 B = one cache line for C = 32), served by C/4 adjacent lanes with 16-byte loads.
#include "asd_common.h"

// ---- importance resampling ------------------------------------------------------------------------------------------
// One thread per output edge.  u_j = (j + jitter[r]) / (n_out + 1) (stratified) or j / n_out; p = last index <= e_in-2 with
// cdf[p] <= u (binary search; the cdf of a ray is non-decreasing); linear interpolation inside the segment.
__global__ __launch_bounds__(256) void importance_resample_kernel(const float* __restrict__ vals, const float* __restrict__ cdfs,
                                                                  int n_rays, int e_in, int n_out, const float* __restrict__ jitter,
                                                                  float* __restrict__ out) {
    const int per = n_out + 1;
    const long long total = (long long)n_rays * per;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int r = (int)(q / per), j = (int)(q - (long long)r * per);
        const float* v = vals + (size_t)r * e_in;
        const float* c = cdfs + (size_t)r * e_in;
        const float u = jitter ? ((float)j + jitter[r]) / (float)(n_out + 1) : (float)j / (float)n_out;
        int lo = 0, hi = e_in - 2;   // invariant: answer in [lo, hi]; c[lo] <= u or lo == 0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (c[mid] <= u) lo = mid; else hi = mid - 1;
        }
        const float c0 = c[lo], c1 = c[lo + 1];
        const float w = c1 > c0 ? fminf(fmaxf((u - c0) / (c1 - c0), 0.f), 1.f) : 0.f;
        out[q] = fmaf(w, v[lo + 1] - v[lo], v[lo]);
    }
}

// cdf[r, j] = 1 - exp(-sum_{k<j} sigma_k dt_k), cdf[r, S] = 1.  One thread per ray: the running sum is a sequential fmaf
// chain (same rounding as the oracle); n_rays * S is a few MB, the kernel is latency-trivial.
__global__ __launch_bounds__(256) void transmittance_cdf_kernel(const float* __restrict__ t_edges, const float* __restrict__ sigma,
                                                                int n_rays, int S, float* __restrict__ cdf) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    const float* t = t_edges + (size_t)r * (S + 1);
    const float* sg = sigma + (size_t)r * S;
    float* o = cdf + (size_t)r * (S + 1);
    float acc = 0.f, t0 = t[0];
    for (int j = 0; j < S; ++j) {
        o[j] = 1.f - expf(-acc);
        const float t1 = t[j + 1];
        acc = fmaf(sg[j], t1 - t0, acc);
        t0 = t1;
    }
    o[S] = 1.f;
}

// merge of two sorted lists per ray: out position of a[i] = i + #{b < a[i]}, of b[j] = j + #{a <= b[j]}  (ties: a first)
__global__ __launch_bounds__(256) void merge_sorted_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                           int n_rays, float* __restrict__ out) {
    const int per = na + nb;
    const long long total = (long long)n_rays * per;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int r = (int)(q / per), e = (int)(q - (long long)r * per);
        const float* x = a + (size_t)r * na;
        const float* y = b + (size_t)r * nb;
        float v;
        int pos;
        if (e < na) {
            v = x[e];
            int lo = 0, hi = nb;   // first index with y[idx] >= v
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (y[mid] < v) lo = mid + 1; else hi = mid; }
            pos = e + lo;
        } else {
            const int j = e - na;
            v = y[j];
            int lo = 0, hi = na;   // first index with x[idx] > v
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (x[mid] <= v) lo = mid + 1; else hi = mid; }
            pos = j + lo;
        }
        out[(size_t)r * per + pos] = v;
    }
}

// ---- grid_sample helpers ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gs_axis(float x, int size, int& i0, float& w1) {
    const float ix = ((x + 1.f) * (float)size - 1.f) * 0.5f;   // align_corners = False
    const float f = floorf(ix);
    i0 = (int)f;
    w1 = ix - f;
}

// voxel_cl [B, D, H, W, C], points [B*M, 3] -> out [B*M, C].  LPP = C/4 lanes per point (float4 each).
template <int LPP>
__global__ __launch_bounds__(256) void voxel_sample_fwd_kernel(const float* __restrict__ voxel, int B, int D, int H, int W,
                                                               const float* __restrict__ points, int M, float* __restrict__ out) {
    constexpr int C = LPP * 4;
    const long long total = (long long)B * M;
    const int sub = threadIdx.x % LPP;
    for (long long q = ((long long)blockIdx.x * 256 + threadIdx.x) / LPP; q < total; q += (long long)gridDim.x * 256 / LPP) {
        const int b = (int)(q / M);
        const float px = points[3 * q], py = points[3 * q + 1], pz = points[3 * q + 2];
        int x0, y0, z0;
        float fx, fy, fz;
        gs_axis(px, W, x0, fx); gs_axis(py, H, y0, fy); gs_axis(pz, D, z0, fz);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const int x = x0 + dx, y = y0 + dy, z = z0 + dz;
            if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
            const float4 v = *reinterpret_cast<const float4*>(voxel + ((((size_t)b * D + z) * H + y) * W + x) * C + sub * 4);
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
        *reinterpret_cast<float4*>(out + (size_t)q * C + sub * 4) = acc;
    }
}

// Scatter backward.  Samples arrive ray-major and the three finite-difference offset points of a sample are adjacent, so
// consecutive points mostly fall into the same cell (cell = 2/128 of the cube, sample spacing and offsets are smaller).  A lane
// group therefore walks RUN = 8 consecutive points, keeps the 8 corner sums of the current cell in registers and issues its
// atomics only when the cell changes: the kernel is bound by the fp32 atomic rate, and this divides the atomic count by the
// average run length (3-6 on the amortized configs).
#ifndef SCATTER_RUN
#define SCATTER_RUN 8
#endif
// Request-coalesced scatter (tools/atomic_probe2.hip: the atomic units retire ~21 G REQUESTS/s, and the lanes of one instruction that
// fall into the same 64-byte block are one request): LPP lanes share a point and lane `sub` owns the channels sub, sub + LPP,
// sub + 2 LPP ...  — so one atomic instruction of the group covers LPP CONSECUTIVE floats of a corner's channel-last feature row
// (64 B = one request at LPP = 16) instead of every fourth one.
template <int LPP, int CPL>
__global__ __launch_bounds__(256) void voxel_sample_bwd_kernel(const float* __restrict__ d_out, int B, int D, int H, int W,
                                                               const float* __restrict__ points, int M, float* __restrict__ d_voxel, int run) {
    constexpr int C = LPP * CPL;
    const long long total = (long long)B * M;
    const long long chunks = (total + run - 1) / run;
    const int sub = threadIdx.x % LPP;
    for (long long ch = ((long long)blockIdx.x * 256 + threadIdx.x) / LPP; ch < chunks; ch += (long long)gridDim.x * 256 / LPP) {
        float acc[8][CPL];
        long long cur = -1;      // linear index of the current cell's (x0, y0, z0) corner (may be "outside": handled per corner)
        int cb = 0, cx = 0, cy = 0, cz = 0;
        auto flush = [&]() {
            if (cur < 0) return;
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int x = cx + (corner & 1), y = cy + ((corner >> 1) & 1), z = cz + (corner >> 2);
                if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue;
                float* dst = d_voxel + ((((size_t)cb * D + z) * H + y) * W + x) * C + sub;
#pragma unroll
                for (int k = 0; k < CPL; ++k) atomicAdd(dst + k * LPP, acc[corner][k]);
            }
        };
        const long long q_end = min(total, (ch + 1) * run);
        for (long long q = ch * run; q < q_end; ++q) {
            const int b = (int)(q / M);
            int x0, y0, z0;
            float fx, fy, fz;
            gs_axis(points[3 * q], W, x0, fx); gs_axis(points[3 * q + 1], H, y0, fy); gs_axis(points[3 * q + 2], D, z0, fz);
            const long long key = ((((long long)b * (D + 2) + (z0 + 1)) * (H + 2) + (y0 + 1)) * (W + 2) + (x0 + 1));
            if (key != cur) {
                flush();
                cur = key; cb = b; cx = x0; cy = y0; cz = z0;
#pragma unroll
                for (int corner = 0; corner < 8; ++corner)
#pragma unroll
                    for (int k = 0; k < CPL; ++k) acc[corner][k] = 0.f;
            }
            float g[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) g[k] = d_out[(size_t)q * C + sub + k * LPP];
#pragma unroll
            for (int corner = 0; corner < 8; ++corner) {
                const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
                const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
#pragma unroll
                for (int k = 0; k < CPL; ++k) acc[corner][k] = fmaf(w, g[k], acc[corner][k]);
            }
        }
        flush();
    }
}

// planes_cl [B, 3, H, W, C], points [B*M, 3] -> out [B*M, 3*C]; projections (x,y), (x,z), (z,y); first coordinate -> W.
__device__ __forceinline__ void plane_uv(float x, float y, float z, int plane, float& u, float& v) {
    if (plane == 0) { u = x; v = y; } else if (plane == 1) { u = x; v = z; } else { u = z; v = y; }
}

template <int LPP, bool BWD>
__global__ __launch_bounds__(256) void triplane_sample_kernel(const float* __restrict__ src /*planes | d_out*/, int B, int H, int W,
                                                              const float* __restrict__ points, int M, float coord_scale,
                                                              float* __restrict__ dst /*out | d_planes*/) {
    constexpr int C = LPP * 4;
    const long long total = (long long)B * M * 3;
    const int sub = threadIdx.x % LPP;
    for (long long t = ((long long)blockIdx.x * 256 + threadIdx.x) / LPP; t < total; t += (long long)gridDim.x * 256 / LPP) {
        const long long q = t / 3;
        const int pl = (int)(t - q * 3), b = (int)(q / M);
        float u, v, fx, fy;
        int x0, y0;
        plane_uv(points[3 * q] * coord_scale, points[3 * q + 1] * coord_scale, points[3 * q + 2] * coord_scale, pl, u, v);
        gs_axis(u, W, x0, fx); gs_axis(v, H, y0, fy);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = acc;
        if (BWD) g = *reinterpret_cast<const float4*>(src + (size_t)t * C + sub * 4);
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
            const int dx = corner & 1, dy = corner >> 1;
            const int x = x0 + dx, y = y0 + dy;
            if (x < 0 || x >= W || y < 0 || y >= H) continue;
            const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
            const size_t cell = ((((size_t)b * 3 + pl) * H + y) * W + x) * C + sub * 4;
            if (BWD) {
                atomicAdd(dst + cell, w * g.x); atomicAdd(dst + cell + 1, w * g.y);
                atomicAdd(dst + cell + 2, w * g.z); atomicAdd(dst + cell + 3, w * g.w);
            } else {
                const float4 s = *reinterpret_cast<const float4*>(src + cell);
                acc.x = fmaf(w, s.x, acc.x); acc.y = fmaf(w, s.y, acc.y); acc.z = fmaf(w, s.z, acc.z); acc.w = fmaf(w, s.w, acc.w);
            }
        }
        if (!BWD) *reinterpret_cast<float4*>(dst + (size_t)t * C + sub * 4) = acc;
    }
}

// tri-plane scatter backward with the same run accumulation: a lane group owns one plane of SCATTER_RUN consecutive points
template <int LPP, int CPL>
__global__ __launch_bounds__(256) void triplane_sample_bwd_kernel(const float* __restrict__ d_out, int B, int H, int W,
                                                                  const float* __restrict__ points, int M, float coord_scale,
                                                                  float* __restrict__ d_planes, int run) {
    constexpr int C = LPP * CPL;     // lane `sub` of a group owns channels sub + k * LPP (request-coalesced scatter, see voxel_sample_bwd_kernel)
    const long long total = (long long)B * M;
    const long long chunks = (total + run - 1) / run;
    const int sub = threadIdx.x % LPP;
    for (long long t = ((long long)blockIdx.x * 256 + threadIdx.x) / LPP; t < chunks * 3; t += (long long)gridDim.x * 256 / LPP) {
        const long long ch = t / 3;
        const int pl = (int)(t - ch * 3);
        float acc[4][CPL];
        long long cur = -1;
        int cb = 0, cx = 0, cy = 0;
        auto flush = [&]() {
            if (cur < 0) return;
#pragma unroll
            for (int corner = 0; corner < 4; ++corner) {
                const int x = cx + (corner & 1), y = cy + (corner >> 1);
                if (x < 0 || x >= W || y < 0 || y >= H) continue;
                float* dst = d_planes + ((((size_t)cb * 3 + pl) * H + y) * W + x) * C + sub;
#pragma unroll
                for (int k = 0; k < CPL; ++k) atomicAdd(dst + k * LPP, acc[corner][k]);
            }
        };
        const long long q_end = min(total, (ch + 1) * run);
        for (long long q = ch * run; q < q_end; ++q) {
            const int b = (int)(q / M);
            float u, v, fx, fy;
            int x0, y0;
            plane_uv(points[3 * q] * coord_scale, points[3 * q + 1] * coord_scale, points[3 * q + 2] * coord_scale, pl, u, v);
            gs_axis(u, W, x0, fx); gs_axis(v, H, y0, fy);
            const long long key = (((long long)b * (H + 2) + (y0 + 1)) * (W + 2) + (x0 + 1));
            if (key != cur) {
                flush();
                cur = key; cb = b; cx = x0; cy = y0;
#pragma unroll
                for (int corner = 0; corner < 4; ++corner)
#pragma unroll
                    for (int k = 0; k < CPL; ++k) acc[corner][k] = 0.f;
            }
            float g[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) g[k] = d_out[((size_t)q * 3 + pl) * C + sub + k * LPP];
#pragma unroll
            for (int corner = 0; corner < 4; ++corner) {
                const float w = ((corner & 1) ? fx : 1.f - fx) * ((corner >> 1) ? fy : 1.f - fy);
#pragma unroll
                for (int k = 0; k < CPL; ++k) acc[corner][k] = fmaf(w, g[k], acc[corner][k]);
            }
        }
        flush();
    }
}

// [B, C, S] <-> [B, S, C] through a 32 x 33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void relayout_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.y * rows * cols;
    const int tiles_c = (cols + 31) / 32;
    const int c0 = (int)(blockIdx.x % tiles_c) * 32, r0 = (int)(blockIdx.x / tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * k][tx] = x[base + (size_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) y[base + (size_t)c * rows + r] = tile[tx][ty + 8 * k];
    }
}

extern "C" {

int asd_importance_resample(const float* vals, const float* cdfs, int32_t n_rays, int32_t e_in, int32_t n_out,
                            const float* jitter, float* out, void* stream) {
    if (n_rays == 0) return ASD_OK;
    ASD_CHECK_ARG(vals && cdfs && out && n_rays > 0, "null argument");
    ASD_CHECK_ARG(e_in >= 2 && n_out >= 1, "need at least one input interval and one output interval");
    hipLaunchKernelGGL(importance_resample_kernel, dim3(asd_grid_for((int64_t)n_rays * (n_out + 1), 256)), dim3(256), 0,
                       (hipStream_t)stream, vals, cdfs, n_rays, e_in, n_out, jitter, out);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_transmittance_cdf(const float* t_edges, const float* sigma, int32_t n_rays, int32_t n_samples, float* cdf, void* stream) {
    if (n_rays == 0) return ASD_OK;
    ASD_CHECK_ARG(t_edges && sigma && cdf && n_rays > 0 && n_samples > 0, "bad argument");
    hipLaunchKernelGGL(transmittance_cdf_kernel, dim3(asd_div_up(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, t_edges, sigma,
                       n_rays, n_samples, cdf);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_merge_sorted(const float* a, int32_t na, const float* b, int32_t nb, int32_t n_rays, float* out, void* stream) {
    if (n_rays == 0) return ASD_OK;
    ASD_CHECK_ARG(a && b && out && na > 0 && nb > 0 && n_rays > 0, "bad argument");
    hipLaunchKernelGGL(merge_sorted_kernel, dim3(asd_grid_for((int64_t)n_rays * (na + nb), 256)), dim3(256), 0, (hipStream_t)stream,
                       a, na, b, nb, n_rays, out);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

#define ASD_LPP_DISPATCH(C_, CALL)                      \
    switch (C_) {                                       \
        case 4: { constexpr int LPP = 1; CALL; } break;   \
        case 8: { constexpr int LPP = 2; CALL; } break;   \
        case 16: { constexpr int LPP = 4; CALL; } break;  \
        case 32: { constexpr int LPP = 8; CALL; } break;  \
        case 64: { constexpr int LPP = 16; CALL; } break; \
        default: asd_set_error("feature channels must be 4, 8, 16, 32 or 64 (got %d)", C_); return ASD_ERR_UNSUPPORTED; \
    }

// scatter kernels: up to 16 lanes per point, lane `sub` owning channels sub + k * LPP
#define ASD_SCATTER_DISPATCH(C_, CALL)                                   \
    switch (C_) {                                                        \
        case 4: { constexpr int LPP = 4, CPL = 1; CALL; } break;         \
        case 8: { constexpr int LPP = 8, CPL = 1; CALL; } break;         \
        case 16: { constexpr int LPP = 16, CPL = 1; CALL; } break;       \
        case 32: { constexpr int LPP = 16, CPL = 2; CALL; } break;       \
        case 64: { constexpr int LPP = 16, CPL = 4; CALL; } break;       \
        default: asd_set_error("feature channels must be 4, 8, 16, 32 or 64 (got %d)", C_); return ASD_ERR_UNSUPPORTED; \
    }

int asd_voxel_sample_fwd(const float* voxel_cl, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, const float* points,
                         int32_t M, float* out, void* stream) {
    if ((int64_t)B * M == 0) return ASD_OK;
    ASD_CHECK_ARG(voxel_cl && points && out && B > 0 && D > 0 && H > 0 && W > 0 && M > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    ASD_LPP_DISPATCH(C, hipLaunchKernelGGL((voxel_sample_fwd_kernel<LPP>), dim3(asd_grid_for((int64_t)B * M * LPP, 256)), dim3(256), 0, s,
                                           voxel_cl, B, D, H, W, points, M, out));
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

static int voxel_sample_bwd_run(const float* d_out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, const float* points, int32_t M, float* d_voxel_cl,
                                int run, void* stream) {
    if ((int64_t)B * M == 0) return ASD_OK;
    ASD_CHECK_ARG(d_out && points && d_voxel_cl && B > 0 && D > 0 && H > 0 && W > 0 && M > 0 && run > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    ASD_SCATTER_DISPATCH(C, hipLaunchKernelGGL((voxel_sample_bwd_kernel<LPP, CPL>), dim3(asd_grid_for(asd_div_up((int64_t)B * M, run) * LPP, 256)),
                                           dim3(256), 0, s, d_out, B, D, H, W, points, M, d_voxel_cl, run));
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_voxel_sample_bwd(const float* d_out, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, const float* points, int32_t M,
                         float* d_voxel_cl, void* stream) {
    return voxel_sample_bwd_run(d_out, B, D, H, W, C, points, M, d_voxel_cl, SCATTER_RUN, stream);
}

// rows in ray order (the fused voxel field's feature-gradient rows: stencil points of a sample adjacent, samples along a ray): a lane group keeps
// summing in registers while consecutive rows stay in one cell, so long runs cut the atomics (C4 step 54.6 ms at 8 rows, 51.6 at 128)
int asd_voxel_sample_bwd_rows(const float* d_out, int32_t D, int32_t H, int32_t W, int32_t C, const float* points, int32_t rows, float* d_voxel_cl,
                              int32_t run, void* stream) {
    return voxel_sample_bwd_run(d_out, 1, D, H, W, C, points, rows, d_voxel_cl, run, stream);
}

int asd_triplane_sample_fwd(const float* planes_cl, int32_t B, int32_t H, int32_t W, int32_t C, const float* points, int32_t M,
                            float coord_scale, float* out, void* stream) {
    if ((int64_t)B * M == 0) return ASD_OK;
    ASD_CHECK_ARG(planes_cl && points && out && B > 0 && H > 0 && W > 0 && M > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    ASD_LPP_DISPATCH(C, hipLaunchKernelGGL((triplane_sample_kernel<LPP, false>), dim3(asd_grid_for((int64_t)B * M * 3 * LPP, 256)), dim3(256),
                                           0, s, planes_cl, B, H, W, points, M, coord_scale, out));
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// `run`: consecutive rows one lane group walks, accumulating in registers while they stay in one cell of its plane (atomics only when the cell changes)
static int triplane_sample_bwd_run(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t C, const float* points, int32_t M, float coord_scale,
                                   float* d_planes_cl, int run, void* stream) {
    if ((int64_t)B * M == 0) return ASD_OK;
    ASD_CHECK_ARG(d_out && points && d_planes_cl && B > 0 && H > 0 && W > 0 && M > 0 && run > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    ASD_SCATTER_DISPATCH(C, hipLaunchKernelGGL((triplane_sample_bwd_kernel<LPP, CPL>), dim3(asd_grid_for(asd_div_up((int64_t)B * M, run) * 3 * LPP, 256)),
                                           dim3(256), 0, s, d_out, B, H, W, points, M, coord_scale, d_planes_cl, run));
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_triplane_sample_bwd(const float* d_out, int32_t B, int32_t H, int32_t W, int32_t C, const float* points, int32_t M,
                            float coord_scale, float* d_planes_cl, void* stream) {
    return triplane_sample_bwd_run(d_out, B, H, W, C, points, M, coord_scale, d_planes_cl, SCATTER_RUN, stream);
}

// rows in ray order (the fused tri-plane field's feature-gradient rows: four stencil points per sample, samples along a ray): long runs
int asd_triplane_sample_bwd_rows(const float* d_out, int32_t H, int32_t W, int32_t C, const float* points, int32_t rows, float* d_planes_cl, int32_t run,
                                 void* stream) {
    return triplane_sample_bwd_run(d_out, 1, H, W, C, points, rows, 1.f, d_planes_cl, run, stream);
}

int asd_relayout_f32(const float* x, int32_t batch, int32_t rows, int32_t cols, float* y, void* stream) {
    if ((int64_t)batch * rows * cols == 0) return ASD_OK;
    ASD_CHECK_ARG(x && y && batch > 0 && rows > 0 && cols > 0 && batch <= 65535, "bad argument");
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)asd_div_up(cols, 32) * (unsigned)asd_div_up(rows, 32), batch), dim3(256), 0,
                       (hipStream_t)stream, x, rows, cols, y);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
