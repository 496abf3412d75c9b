// render.hip — occupancy-grid ray marching, visibility pruning, wave-level compaction and alpha
// compositing for gfx950.  One 64-lane wavefront owns one ray: the per-ray exclusive scans
// (transmittance), ballots (active-sample compaction) and reductions (accumulate_along_rays) are
// wave-wide shuffles / ballots, no LDS round trip and no atomics.
//
// Replaces (include/asd_hip.h cites the call sites):
//   nerfacc.OccGridEstimator.sampling -> traverse_grids + render_visibility_from_density
//   nerfacc.render_weight_from_density / render_weight_from_alpha + accumulate_along_rays
//   and the glue of threestudio/models/renderers/nerf_volume_renderer.py:126-180,269-279,312-364.
// Roofline: bandwidth-trivial (32 B/sample in, 36 B/ray out — SURVEY.md §8d); these kernels are
// latency/launch bound, so the design goal is FEW launches, not bytes.
#include "asd_common.h"

#define RAYS_PER_BLOCK 4  // 256 threads = 4 waves = 4 rays

__device__ __forceinline__ bool ray_aabb(const float o[3], const float d[3], const float* aabb, float& tmin,
                                         float& tmax) {
    float t0 = -INFINITY, t1 = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.f) {
            if (o[a] < aabb[a] || o[a] > aabb[3 + a]) return false;
            continue;
        }
        const float inv = 1.f / d[a];
        float ta = (aabb[a] - o[a]) * inv, tb = (aabb[3 + a] - o[a]) * inv;
        if (ta > tb) { const float s = ta; ta = tb; tb = s; }
        if (ta > t0) t0 = ta;
        if (tb < t1) t1 = tb;
    }
    tmin = t0;
    tmax = t1;
    return t1 >= t0;
}

__device__ __forceinline__ int cell_of(const asd_march_cfg& c, float px, float py, float pz) {
    const float p[3] = {px, py, pz};
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float u = (p[a] - c.aabb[a]) / (c.aabb[3 + a] - c.aabb[a]);
        if (!(u >= 0.f && u <= 1.f)) return -1;
        int ci = (int)floorf(u * (float)c.resolution);
        if (ci > c.resolution - 1) ci = c.resolution - 1;
        idx[a] = ci;
    }
    return (idx[0] * c.resolution + idx[1]) * c.resolution + idx[2];
}

// Shared body of the count / write passes.  WRITE=false: only count[r]; WRITE=true: emit samples.
template <bool WRITE>
__global__ __launch_bounds__(256) void march_kernel(const asd_march_cfg c, const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d, int n_rays,
                                                    const uint32_t* __restrict__ occ_bits,
                                                    const float* __restrict__ jitter, int* __restrict__ count,
                                                    const int* __restrict__ offset, int* __restrict__ ray_idx,
                                                    float* __restrict__ t_start, float* __restrict__ t_end,
                                                    float* __restrict__ points) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = asd_lane();
    const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    float tmin, tmax;
    int total = 0;
    if (ray_aabb(o, d, c.aabb, tmin, tmax)) {
        const float near_eff = jitter ? fmaf(jitter[r], c.step, c.near_plane) : c.near_plane;
        const float t_begin = fmaxf(tmin, near_eff);
        const float t_exit = fminf(tmax, c.far_plane);
        const int base = WRITE ? offset[r] : 0;
        for (int k0 = 0; k0 < c.max_steps; k0 += 64) {
            const int k = k0 + lane;
            const float t0 = fmaf((float)k, c.step, t_begin);
            const float t1 = fmaf((float)(k + 1), c.step, t_begin);
            const float tm = (t0 + t1) / 2.0f;
            const bool in_range = (k < c.max_steps) && (tm <= t_exit);
            // lane 0 holds the smallest t of the chunk: if it is past the exit, every later one is too
            if (!__shfl((int)in_range, 0, 64)) break;
            // = t_origins + t_dirs * (t0+t1)/2 exactly as the reference forms it (separate mul and add)
            const float px = o[0] + d[0] * tm, py = o[1] + d[1] * tm, pz = o[2] + d[2] * tm;
            bool emit = false;
            if (in_range) {
                const int cell = cell_of(c, px, py, pz);
                emit = cell >= 0 && ((occ_bits[cell >> 5] >> (cell & 31)) & 1u);
            }
            const unsigned long long mask = __ballot(emit);
            if (WRITE && emit) {
                const int dst = base + total + asd_ballot_rank(mask);
                ray_idx[dst] = r;
                t_start[dst] = t0;
                t_end[dst] = t1;
                if (points) {
                    points[3 * (size_t)dst] = px;
                    points[3 * (size_t)dst + 1] = py;
                    points[3 * (size_t)dst + 2] = pz;
                }
            }
            total += __popcll(mask);
        }
    }
    if (!WRITE && lane == 0) count[r] = total;
}

// exclusive scan of n int32 in one block of 1024 threads (n_rays is 4 096 .. 262 144)
__global__ __launch_bounds__(1024) void scan_i32_kernel(const int* __restrict__ count, int n, int* __restrict__ offset,
                                                        int* __restrict__ total) {
    __shared__ int wave_sums[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int b = tid * per, e = min(b + per, n);
    int local = 0;
    for (int i = b; i < e; ++i) local += count[i];
    int v = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    if (lane == 63) wave_sums[wid] = v;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; ++w) { const int t = wave_sums[w]; wave_sums[w] = acc; acc += t; }
        carry_s = acc;
    }
    __syncthreads();
    int run = wave_sums[wid] + v - local;
    for (int i = b; i < e; ++i) { offset[i] = run; run += count[i]; }
    if (tid == 0 && total) total[0] = carry_s;
}

// visibility pruning: keep flags + per-ray kept count
__global__ __launch_bounds__(256) void prune_kernel(const float* __restrict__ sigma, const float* __restrict__ t_start,
                                                    const float* __restrict__ t_end, const int* __restrict__ offset,
                                                    const int* __restrict__ count, int n_rays, float early_stop_eps,
                                                    float alpha_thre, uint8_t* __restrict__ keep,
                                                    int* __restrict__ kept_count) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = asd_lane();
    const int b = offset[r], cnt = count[r];
    float carry = 0.f;
    int kept = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < cnt;
        const float sd = valid ? sigma[b + j] * (t_end[b + j] - t_start[b + j]) : 0.f;
        const float incl = asd_wave_incl_scan(sd);
        const float excl = carry + (incl - sd);
        const float T = expf(-excl);
        const float alpha = 1.f - expf(-sd);
        const bool k = valid && (T >= early_stop_eps) && (alpha >= alpha_thre);
        if (valid) keep[b + j] = (uint8_t)k;
        kept += __popcll(__ballot(k));
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) kept_count[r] = kept;
}

__global__ __launch_bounds__(256) void compact_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                      int n_rays, const int* __restrict__ offset,
                                                      const int* __restrict__ count, const uint8_t* __restrict__ keep,
                                                      const float* __restrict__ t_start,
                                                      const float* __restrict__ t_end,
                                                      const int* __restrict__ kept_offset,
                                                      int64_t* __restrict__ ray_idx_out, float* __restrict__ t0_out,
                                                      float* __restrict__ t1_out, float* __restrict__ points_out,
                                                      float* __restrict__ dirs_out, const float* __restrict__ c_sigma,
                                                      const float* __restrict__ c_feats, const float* __restrict__ c_enc,
                                                      float* __restrict__ sigma_out, float* __restrict__ feats_out, float* __restrict__ enc_out) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = asd_lane();
    const int b = offset[r], cnt = count[r];
    int dst0 = kept_offset[r];
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        const bool k = (j < cnt) && (keep ? keep[b + j] != 0 : true);
        const unsigned long long mask = __ballot(k);
        if (k) {
            const int dst = dst0 + asd_ballot_rank(mask);
            const float t0 = t_start[b + j], t1 = t_end[b + j];
            // positions = o + d * (t0+t1)/2   (nerf_volume_renderer.py:276-278)
            const float tm = (t0 + t1) / 2.0f;
            ray_idx_out[dst] = r;
            t0_out[dst] = t0;
            t1_out[dst] = t1;
            if (points_out) {
                points_out[3 * (size_t)dst] = ox + dx * tm;
                points_out[3 * (size_t)dst + 1] = oy + dy * tm;
                points_out[3 * (size_t)dst + 2] = oz + dz * tm;
            }
            if (dirs_out) {
                dirs_out[3 * (size_t)dst] = dx;
                dirs_out[3 * (size_t)dst + 1] = dy;
                dirs_out[3 * (size_t)dst + 2] = dz;
            }
            if (c_sigma) {      // the field was evaluated on the candidates: its outputs move with the sample
                sigma_out[dst] = c_sigma[b + j];
                feats_out[3 * (size_t)dst] = c_feats[3 * (size_t)(b + j)];
                feats_out[3 * (size_t)dst + 1] = c_feats[3 * (size_t)(b + j) + 1];
                feats_out[3 * (size_t)dst + 2] = c_feats[3 * (size_t)(b + j) + 2];
            }
        }
        if (c_enc) {
            // the 128-byte encoding rows: eight lanes per row (16 bytes each), eight kept samples of this trip per pass
            const int my_src = k ? b + j : -1, my_dst = k ? dst0 + asd_ballot_rank(mask) : 0;
            for (int s0 = 0; s0 < 64; s0 += 8) {
                if (!((mask >> s0) & 0xffull)) continue;                     // wave-uniform
                const int sl = s0 + (lane >> 3);
                const int src = __shfl(my_src, sl), dd = __shfl(my_dst, sl);
                if (src >= 0) {
                    const float4 v = reinterpret_cast<const float4*>(c_enc + (size_t)src * 32)[lane & 7];
                    reinterpret_cast<float4*>(enc_out + (size_t)dd * 32)[lane & 7] = v;
                }
            }
        }
        dst0 += __popcll(mask);
    }
}

// ---------------------------------------------------------------------------------------------------
// compositing
// ---------------------------------------------------------------------------------------------------
// colour of sample i, channel k: `rgb` holds activated colours, or — rgb_act == 1 — the raw features of a material that is
// colour = sigmoid(features) (NoMaterial, no_material.py:41-54): the activation and its gradient then ride in these kernels
__device__ __forceinline__ float comp_colour(const float* __restrict__ rgb, size_t i, int k, int rgb_act) {
    const float v = rgb[3 * i + k];
    return rgb_act == 1 ? 1.f / (1.f + expf(-v)) : v;
}

template <int MODE>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ t_start, const float* __restrict__ t_end,
    const float* __restrict__ rgb, const int* __restrict__ offset, const int* __restrict__ count, int n_rays,
    const float* __restrict__ bg, float* __restrict__ weights, float* __restrict__ opacity, float* __restrict__ depth,
    float* __restrict__ rgb_fg, float* __restrict__ z_var, float* __restrict__ comp_rgb, int rgb_act = 0) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = asd_lane();
    const int b = offset[r], cnt = count[r];
    // MODE 0: density in (alpha = 1 - exp(-sigma dt)); MODE 1 / 2: alpha in.  MODE 2 differs from 1 only in z_var: the VolSDF
    // renderer's plain second moment sum_i w_i (t_i - depth)^2 (generative_space_volsdf_volume_renderer.py:380-385) instead of the
    // opacity-normalised, opacity-masked variance of the NeRF renderer (nerf_volume_renderer.py:335-349)
    float carry = MODE == 0 ? 0.f : 1.f;
    float op = 0.f, dp = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < cnt;
        const int i = b + j;
        float T, alpha;
        if (MODE == 0) {
            const float sd = valid ? sigma[i] * (t_end[i] - t_start[i]) : 0.f;
            const float incl = asd_wave_incl_scan(sd);
            T = expf(-(carry + (incl - sd)));
            alpha = 1.f - expf(-sd);
            carry += __shfl(incl, 63, 64);
        } else {
            alpha = valid ? sigma[i] : 0.f;
            const float om = 1.f - alpha;
            const float incl = asd_wave_incl_prod(om);
            // exclusive product: shift the inclusive scan by one lane
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            T = carry * excl;
            carry *= __shfl(incl, 63, 64);
        }
        if (valid) {
            const float w = T * alpha, t = (t_start[i] + t_end[i]) * 0.5f;
            weights[i] = w;
            op += w;
            dp = fmaf(w, t, dp);
            c0 = fmaf(w, comp_colour(rgb, i, 0, rgb_act), c0);
            c1 = fmaf(w, comp_colour(rgb, i, 1, rgb_act), c1);
            c2 = fmaf(w, comp_colour(rgb, i, 2, rgb_act), c2);
        }
    }
    op = asd_wave_sum(op); dp = asd_wave_sum(dp);
    c0 = asd_wave_sum(c0); c1 = asd_wave_sum(c1); c2 = asd_wave_sum(c2);
    const float m = MODE == 2 ? 1.f : fmaxf(op, 1e-5f), zm = dp / m;
    float zv = 0.f;
    for (int j = lane; j < cnt; j += 64) {
        const int i = b + j;
        const float t = (t_start[i] + t_end[i]) * 0.5f;
        zv = fmaf(weights[i] / m, (t - zm) * (t - zm), zv);
    }
    zv = asd_wave_sum(zv);
    if (lane == 0) {
        opacity[r] = op;
        depth[r] = dp;
        z_var[r] = (MODE == 2 || op > 0.5f) ? zv : 0.f;
        rgb_fg[3 * (size_t)r] = c0; rgb_fg[3 * (size_t)r + 1] = c1; rgb_fg[3 * (size_t)r + 2] = c2;
        const float k = 1.f - op;
        comp_rgb[3 * (size_t)r] = c0 + bg[3 * (size_t)r] * k;
        comp_rgb[3 * (size_t)r + 1] = c1 + bg[3 * (size_t)r + 1] * k;
        comp_rgb[3 * (size_t)r + 2] = c2 + bg[3 * (size_t)r + 2] * k;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ sigma, const float* __restrict__ t_start, const float* __restrict__ t_end,
    const float* __restrict__ rgb, const int* __restrict__ offset, const int* __restrict__ count, int n_rays,
    const float* __restrict__ bg, const float* __restrict__ weights, const float* __restrict__ opacity,
    const float* __restrict__ depth, const float* __restrict__ d_comp_rgb, const float* __restrict__ d_rgb_fg,
    const float* __restrict__ d_opacity, const float* __restrict__ d_depth, const float* __restrict__ d_z_var,
    const float* __restrict__ d_weights, float* __restrict__ d_sigma, float* __restrict__ d_rgb,
    float* __restrict__ d_bg, int rgb_act = 0) {
    const int r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const int lane = asd_lane();
    const int b = offset[r], cnt = count[r];
    const float op = opacity[r], m = MODE == 2 ? 1.f : fmaxf(op, 1e-5f), zm = depth[r] / m;
    float G[3], gop = d_opacity ? d_opacity[r] : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gc = d_comp_rgb ? d_comp_rgb[3 * (size_t)r + k] : 0.f;
        G[k] = gc + (d_rgb_fg ? d_rgb_fg[3 * (size_t)r + k] : 0.f);
        gop -= gc * bg[3 * (size_t)r + k];
        if (d_bg && lane == 0) d_bg[3 * (size_t)r + k] = gc * (1.f - op);
    }
    const float gdp = d_depth ? d_depth[r] : 0.f;
    const float gzv = (d_z_var && (MODE == 2 || op > 0.5f)) ? d_z_var[r] : 0.f;
    float zvu = 0.f;
    if (gzv != 0.f && MODE != 2) {
        for (int j = lane; j < cnt; j += 64) {
            const int i = b + j;
            const float t = (t_start[i] + t_end[i]) * 0.5f;
            zvu = fmaf(weights[i] / m, (t - zm) * (t - zm), zvu);
        }
        zvu = asd_wave_sum(zvu);
    }
    // pass 1: total = sum_i w_i gw_i
    float tot = 0.f;
    for (int j = lane; j < cnt; j += 64) {
        const int i = b + j;
        const float t = (t_start[i] + t_end[i]) * 0.5f;
        float gw = gop + gdp * t + (d_weights ? d_weights[i] : 0.f);
        gw = fmaf(G[0], comp_colour(rgb, i, 0, rgb_act), gw);
        gw = fmaf(G[1], comp_colour(rgb, i, 1, rgb_act), gw);
        gw = fmaf(G[2], comp_colour(rgb, i, 2, rgb_act), gw);
        if (gzv != 0.f) gw += MODE == 2 ? gzv * ((t - zm) * (t - zm) - 2.f * t * zm * (1.f - op)) : gzv * ((t - zm) * (t - zm) - zvu) / m;
        tot = fmaf(weights[i], gw, tot);
    }
    tot = asd_wave_sum(tot);
    // pass 2: suffix sums S_i = tot - inclusive_prefix_i and the transmittances
    float carry_s = 0.f;                       // prefix of w*gw
    float carry_t = MODE == 0 ? 0.f : 1.f;     // prefix of sigma*dt (mode 0) / product of (1-alpha) (mode 1)
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < cnt;
        const int i = b + j;
        float t = 0.f, dt = 0.f, w = 0.f, gw = 0.f, sv = 0.f;
        float col[3] = {0.f, 0.f, 0.f};
        if (valid) {
            t = (t_start[i] + t_end[i]) * 0.5f;
            dt = t_end[i] - t_start[i];
            w = weights[i];
            sv = sigma[i];
            gw = gop + gdp * t + (d_weights ? d_weights[i] : 0.f);
#pragma unroll
            for (int k = 0; k < 3; ++k) { col[k] = comp_colour(rgb, i, k, rgb_act); gw = fmaf(G[k], col[k], gw); }
            if (gzv != 0.f) gw += MODE == 2 ? gzv * ((t - zm) * (t - zm) - 2.f * t * zm * (1.f - op)) : gzv * ((t - zm) * (t - zm) - zvu) / m;
        }
        const float wg = w * gw;
        const float incl_s = asd_wave_incl_scan(wg);
        const float S = tot - (carry_s + incl_s);
        carry_s += __shfl(incl_s, 63, 64);
        float ds;
        if (MODE == 0) {
            const float sd = sv * dt;
            const float incl_t = asd_wave_incl_scan(sd);
            const float Tnext = expf(-(carry_t + incl_t));
            carry_t += __shfl(incl_t, 63, 64);
            ds = dt * (Tnext * gw - S);
        } else {
            const float om = valid ? 1.f - sv : 1.f;
            const float incl_t = asd_wave_incl_prod(om);
            float excl = __shfl_up(incl_t, 1, 64);
            if (lane == 0) excl = 1.f;
            const float T = carry_t * excl;
            carry_t *= __shfl(incl_t, 63, 64);
            ds = T * gw - S / fmaxf(1.f - sv, 1e-10f);
        }
        if (valid) {
            d_sigma[i] = ds;
#pragma unroll
            for (int k = 0; k < 3; ++k) d_rgb[3 * (size_t)i + k] = w * G[k] * (rgb_act == 1 ? col[k] * (1.f - col[k]) : 1.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// occupancy grid maintenance
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void occ_ema_kernel(float* __restrict__ occs, const int* __restrict__ cell_idx,
                                                      const float* __restrict__ occ_new, int n_update, float decay) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_update) return;
    const int c = cell_idx[i];
    occs[c] = fmaxf(occs[c] * decay, occ_new[i]);
}
__global__ __launch_bounds__(1024) void occ_mean_kernel(const float* __restrict__ occs, int n_cells,
                                                        float* __restrict__ scratch) {
    __shared__ float ws[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_cells; i += 1024) acc += occs[i];
    acc = asd_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += ws[w];
        scratch[0] = t / (float)n_cells;
    }
}
__global__ __launch_bounds__(256) void occ_threshold_kernel(const float* __restrict__ occs, int n_cells,
                                                            const float* __restrict__ scratch, float occ_thre,
                                                            uint32_t* __restrict__ occ_bits,
                                                            uint8_t* __restrict__ binaries) {
    // one thread per 32-cell word
    const int wi = blockIdx.x * 256 + threadIdx.x;
    if (wi * 32 >= n_cells) return;
    const float thre = fminf(scratch[0], occ_thre);
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int i = wi * 32 + k;
        if (i >= n_cells) break;
        const bool b = occs[i] > thre;
        if (binaries) binaries[i] = (uint8_t)b;
        if (b) bits |= 1u << k;
    }
    occ_bits[wi] = bits;
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
// camera rays from the per-camera parameters (threestudio/utils/ops.py:183-269): one thread per pixel, same operation order as the
// oracle (orc_generate_rays) so that both agree bit for bit
__global__ __launch_bounds__(256) void generate_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ focal, int B, int H, int W,
                                                            int normalize, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const size_t total = (size_t)B * H * W;
    const float cx = (float)W / 2.0f, cy = (float)H / 2.0f;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const int i = (int)(p % W), j = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        const float* m = c2w + (size_t)b * 16;
        const float f = focal[b];
        const float d0 = (((float)i + 0.5f) - cx) / f, d1 = -(((float)j + 0.5f) - cy) / f, d2 = -1.0f;
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = d0 * m[4 * k] + d1 * m[4 * k + 1] + d2 * m[4 * k + 2];
        if (normalize) {
            float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            len = len > 1e-12f ? len : 1e-12f;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = v[k] / len;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { rays_d[p * 3 + k] = v[k]; rays_o[p * 3 + k] = m[4 * k + 3]; }
    }
}

extern "C" {

int asd_march_count(const asd_march_cfg* cfg, const float* rays_o, const float* rays_d, int32_t n_rays,
                    const uint32_t* occ_bits, const float* jitter, int32_t* count, void* stream) {
    ASD_CHECK_ARG(cfg && rays_o && rays_d && occ_bits && count && n_rays >= 0, "null argument");
    if (n_rays == 0) return ASD_OK;
    hipLaunchKernelGGL((march_kernel<false>), dim3(asd_div_up(n_rays, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       *cfg, rays_o, rays_d, n_rays, occ_bits, jitter, count, nullptr, nullptr, nullptr, nullptr, nullptr);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_scan_i32(const int32_t* count, int32_t n, int32_t* offset, int32_t* total, void* stream) {
    ASD_CHECK_ARG(count && offset && n >= 0, "null argument");
    hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, count, n, offset, total);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_march_write(const asd_march_cfg* cfg, const float* rays_o, const float* rays_d, int32_t n_rays,
                    const uint32_t* occ_bits, const float* jitter, const int32_t* offset, int32_t* ray_idx,
                    float* t_start, float* t_end, float* points, void* stream) {
    // the sample arrays may be NULL when the marcher found nothing (zero-sized tensors)
    ASD_CHECK_ARG(cfg && rays_o && rays_d && occ_bits && offset && n_rays >= 0, "null argument");
    if (n_rays == 0) return ASD_OK;
    hipLaunchKernelGGL((march_kernel<true>), dim3(asd_div_up(n_rays, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream,
                       *cfg, rays_o, rays_d, n_rays, occ_bits, jitter, nullptr, offset, ray_idx, t_start, t_end, points);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_prune_count(const float* sigma, const float* t_start, const float* t_end, const int32_t* offset,
                    const int32_t* count, int32_t n_rays, float early_stop_eps, float alpha_thre, uint8_t* keep,
                    int32_t* kept_count, void* stream) {
    ASD_CHECK_ARG(offset && count && kept_count && n_rays >= 0, "null argument");
    if (n_rays == 0) return ASD_OK;
    hipLaunchKernelGGL(prune_kernel, dim3(asd_div_up(n_rays, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, sigma,
                       t_start, t_end, offset, count, n_rays, early_stop_eps, alpha_thre, keep, kept_count);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_compact(const float* rays_o, const float* rays_d, int32_t n_rays, const int32_t* offset, const int32_t* count,
                const uint8_t* keep, const float* t_start, const float* t_end, const int32_t* kept_offset,
                int64_t* ray_idx_out, float* t_start_out, float* t_end_out, float* points_out, float* dirs_out,
                void* stream) {
    ASD_CHECK_ARG(rays_o && rays_d && offset && count && kept_offset && n_rays >= 0, "null argument");
    if (n_rays == 0) return ASD_OK;
    hipLaunchKernelGGL(compact_kernel, dim3(asd_div_up(n_rays, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, rays_o,
                       rays_d, n_rays, offset, count, keep, t_start, t_end, kept_offset, ray_idx_out, t_start_out,
                       t_end_out, points_out, dirs_out, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_composite_fwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end, const float* rgb,
                      const int32_t* offset, const int32_t* count, int32_t n_rays, const float* bg, float* weights,
                      float* opacity, float* depth, float* rgb_fg, float* z_var, float* comp_rgb, void* stream) {
    ASD_CHECK_ARG(offset && count && bg && opacity && depth && rgb_fg && z_var && comp_rgb && n_rays >= 0,
                  "null argument");
    ASD_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (density), 1 (alpha) or 2 (alpha, VolSDF z-variance)");
    if (n_rays == 0) return ASD_OK;
    const dim3 g(asd_div_up(n_rays, RAYS_PER_BLOCK)), blk(256);
#define COMPOSITE_FWD(M) hipLaunchKernelGGL((composite_fwd_kernel<M>), g, blk, 0, (hipStream_t)stream, sigma, t_start, t_end, rgb, offset, \
                           count, n_rays, bg, weights, opacity, depth, rgb_fg, z_var, comp_rgb)
    if (mode == 0) COMPOSITE_FWD(0); else if (mode == 1) COMPOSITE_FWD(1); else COMPOSITE_FWD(2);
#undef COMPOSITE_FWD
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_composite_bwd(int32_t mode, const float* sigma, const float* t_start, const float* t_end, const float* rgb,
                      const int32_t* offset, const int32_t* count, int32_t n_rays, const float* bg,
                      const float* weights, const float* opacity, const float* depth, const float* d_comp_rgb,
                      const float* d_rgb_fg, const float* d_opacity, const float* d_depth, const float* d_z_var,
                      const float* d_weights, float* d_sigma, float* d_rgb, float* d_bg, void* stream) {
    ASD_CHECK_ARG(offset && count && bg && opacity && depth && n_rays >= 0, "null argument");
    ASD_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (density), 1 (alpha) or 2 (alpha, VolSDF z-variance)");
    if (n_rays == 0) return ASD_OK;
    const dim3 g(asd_div_up(n_rays, RAYS_PER_BLOCK)), blk(256);
#define COMPOSITE_BWD(M) hipLaunchKernelGGL((composite_bwd_kernel<M>), g, blk, 0, (hipStream_t)stream, sigma, t_start, t_end, rgb, offset, \
                           count, n_rays, bg, weights, opacity, depth, d_comp_rgb, d_rgb_fg, d_opacity, d_depth, d_z_var, \
                           d_weights, d_sigma, d_rgb, d_bg)
    if (mode == 0) COMPOSITE_BWD(0); else if (mode == 1) COMPOSITE_BWD(1); else COMPOSITE_BWD(2);
#undef COMPOSITE_BWD
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The renderer's training pass as ONE entry point each way (NeRFVolumeRenderer.forward, nerf_volume_renderer.py:118-428, with the
// occupancy-grid estimator, a fused ImplicitVolume field and a colour = activation(features) material): march -> candidate densities ->
// visibility pruning -> compaction -> field at the kept samples -> compositing, enqueued from here with every count left on the device.
// All buffers live in one caller-owned workspace (asd_render_layout_init gives the offsets), sized by `capacity` >= n_rays * max_steps.
// ---------------------------------------------------------------------------------------------------------------------------------
static inline int64_t rl_al(int64_t v) { return (v + 255) & ~(int64_t)255; }

int asd_render_layout_init(int32_t n_rays, int32_t capacity, asd_render_layout* L) {
    ASD_CHECK_ARG(L && n_rays > 0 && capacity > 0, "bad argument");
    int64_t o = 0;
    auto take = [&](int64_t bytes) { const int64_t at = o; o += rl_al(bytes); return at; };
    const int64_t nr = n_rays, cap = capacity;
    L->count = take(nr * 4); L->offset = take(nr * 4); L->total = take(4);
    L->c_ray_idx = take(cap * 4); L->c_t0 = take(cap * 4); L->c_t1 = take(cap * 4); L->c_pts = take(cap * 12); L->c_sigma = take(cap * 4);
    L->keep = take(cap); L->kept = take(nr * 4); L->koff = take(nr * 4); L->n_kept = take(4);
    L->ray_idx = take(cap * 8); L->t0 = take(cap * 4); L->t1 = take(cap * 4); L->pts = take(cap * 12); L->dirs = take(cap * 12);
    L->sigma = take(cap * 4); L->feats = take(cap * 12); L->enc = take(cap * 128); L->weights = take(cap * 4);
    L->opacity = take(nr * 4); L->depth = take(nr * 4); L->z_var = take(nr * 4); L->rgb_fg = take(nr * 12); L->comp_rgb = take(nr * 12);
    L->c_feats = take(cap * 12); L->c_enc = take(cap * 128);       // (behind the per-ray outputs: the tail a caller copies out stays short)
    L->total_bytes = o;
    return ASD_OK;
}

static int render_check(const asd_render_params* p) {
    ASD_CHECK_ARG(p && p->meta && p->field && p->rays_o && p->rays_d && p->occ_bits && p->grid && p->w1d && p->w2d && p->w1f && p->w2f && p->bg, "null argument");
    ASD_CHECK_ARG(p->n_rays > 0 && (int64_t)p->capacity >= (int64_t)p->n_rays * p->march.max_steps, "capacity must cover n_rays * max_steps candidates");
    ASD_CHECK_ARG(p->field->n_feature_dims == 3 && p->field->field_mode == ASD_FIELD_DENSITY, "the fused pass renders a density field with 3 feature dims");
    ASD_CHECK_ARG(p->color_act == 0 || p->color_act == 1, "color_act: 0 (features are colours) or 1 (sigmoid)");
    return ASD_OK;
}

int asd_render_fwd(const asd_render_params* p, void* workspace, void* stream) {
    if (render_check(p) != ASD_OK) return ASD_ERR_ARG;
    ASD_CHECK_ARG(workspace, "null workspace");
    asd_render_layout L;
    asd_render_layout_init(p->n_rays, p->capacity, &L);
    char* w = (char*)workspace;
#define AT(T, f) ((T*)(w + L.f))
    const int nr = p->n_rays, cap = p->capacity;
    int rc;
#define STEP(call) do { rc = (call); if (rc != ASD_OK) return rc; } while (0)
    STEP(asd_march_count(&p->march, p->rays_o, p->rays_d, nr, p->occ_bits, p->jitter, AT(int32_t, count), stream));
    STEP(asd_scan_i32(AT(int32_t, count), nr, AT(int32_t, offset), AT(int32_t, total), stream));
    STEP(asd_march_write(&p->march, p->rays_o, p->rays_d, nr, p->occ_bits, p->jitter, AT(int32_t, offset), AT(int32_t, c_ray_idx), AT(float, c_t0), AT(float, c_t1),
                         AT(float, c_pts), stream));
    const int32_t *k_off, *k_cnt, *n_kept;
    // ASD_RENDER_CANDIDATE_FIELD=0 (A/B): densities of the candidates, then the whole field again at the kept samples (the reference's order:
    // nerfacc's sigma_fn inside the sampling, then geometry(positions)).  Default: the field ONCE, on the candidates — the hash-grid gathers
    // are what a sample costs, the feature head on the pruned candidates is cheap beside a second encode of the kept ones — and the
    // compaction carries sigma / features / encoding rows along (bit-identical values: same function of the same positions).
    static const bool candidate_field = !(getenv("ASD_RENDER_CANDIDATE_FIELD") && getenv("ASD_RENDER_CANDIDATE_FIELD")[0] == '0');
    const bool fused = p->prune && candidate_field;
    if (p->prune) {
        if (fused)
            STEP(asd_field_fwd(p->meta, p->field, p->grid, p->w1d, p->w2d, p->w1f, p->w2f, AT(float, c_pts), cap, AT(int32_t, total), AT(float, c_sigma),
                               AT(float, c_feats), nullptr, nullptr, AT(float, c_enc), stream));
        else
            STEP(asd_field_density(p->meta, p->field, p->grid, p->w1d, p->w2d, AT(float, c_pts), cap, AT(int32_t, total), AT(float, c_sigma), stream));
        STEP(asd_prune_count(AT(float, c_sigma), AT(float, c_t0), AT(float, c_t1), AT(int32_t, offset), AT(int32_t, count), nr, p->early_stop_eps, p->alpha_thre,
                             AT(uint8_t, keep), AT(int32_t, kept), stream));
        STEP(asd_scan_i32(AT(int32_t, kept), nr, AT(int32_t, koff), AT(int32_t, n_kept), stream));
        if (fused) {
            hipLaunchKernelGGL(compact_kernel, dim3(asd_div_up(nr, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, p->rays_o, p->rays_d, nr, AT(int32_t, offset),
                               AT(int32_t, count), AT(uint8_t, keep), AT(float, c_t0), AT(float, c_t1), AT(int32_t, koff), AT(int64_t, ray_idx), AT(float, t0),
                               AT(float, t1), AT(float, pts), AT(float, dirs), AT(float, c_sigma), AT(float, c_feats), AT(float, c_enc), AT(float, sigma),
                               AT(float, feats), AT(float, enc));
        } else {
            STEP(asd_compact(p->rays_o, p->rays_d, nr, AT(int32_t, offset), AT(int32_t, count), AT(uint8_t, keep), AT(float, c_t0), AT(float, c_t1), AT(int32_t, koff),
                             AT(int64_t, ray_idx), AT(float, t0), AT(float, t1), AT(float, pts), AT(float, dirs), stream));
        }
        k_off = AT(int32_t, koff); k_cnt = AT(int32_t, kept); n_kept = AT(int32_t, n_kept);
    } else {
        STEP(asd_compact(p->rays_o, p->rays_d, nr, AT(int32_t, offset), AT(int32_t, count), nullptr, AT(float, c_t0), AT(float, c_t1), AT(int32_t, offset),
                         AT(int64_t, ray_idx), AT(float, t0), AT(float, t1), AT(float, pts), AT(float, dirs), stream));
        (void)hipMemcpyAsync(AT(int32_t, koff), AT(int32_t, offset), (size_t)nr * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        (void)hipMemcpyAsync(AT(int32_t, kept), AT(int32_t, count), (size_t)nr * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        (void)hipMemcpyAsync(AT(int32_t, n_kept), AT(int32_t, total), 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        k_off = AT(int32_t, koff); k_cnt = AT(int32_t, kept); n_kept = AT(int32_t, n_kept);
    }
    if (!fused)
        STEP(asd_field_fwd(p->meta, p->field, p->grid, p->w1d, p->w2d, p->w1f, p->w2f, AT(float, pts), cap, n_kept, AT(float, sigma), AT(float, feats), nullptr,
                           nullptr, AT(float, enc), stream));
    hipLaunchKernelGGL((composite_fwd_kernel<0>), dim3(asd_div_up(nr, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, AT(float, sigma), AT(float, t0), AT(float, t1),
                       AT(float, feats), k_off, k_cnt, nr, p->bg, AT(float, weights), AT(float, opacity), AT(float, depth), AT(float, rgb_fg), AT(float, z_var),
                       AT(float, comp_rgb), p->color_act);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_render_bwd_workspace(const asd_render_params* p, int64_t* n_floats) {
    ASD_CHECK_ARG(p && p->field && n_floats, "null argument");
    int64_t nf = 0;
    const int rc = asd_field_bwd_workspace(p->field, p->capacity, 0, &nf);
    if (rc != ASD_OK) return rc;
    *n_floats = nf + (int64_t)4 * p->capacity + 64;
    return ASD_OK;
}

int asd_render_bwd(const asd_render_params* p, void* workspace, const float* d_comp_rgb, const float* d_rgb_fg, const float* d_opacity, const float* d_depth,
                   const float* d_z_var, float* d_grid, float* dw1d, float* dw2d, float* dw1f, float* dw2f, float* d_bg, float* bwd_workspace, void* stream) {
    if (render_check(p) != ASD_OK) return ASD_ERR_ARG;
    ASD_CHECK_ARG(workspace && bwd_workspace && d_grid && dw1d && dw2d && dw1f && dw2f, "null argument");
    asd_render_layout L;
    asd_render_layout_init(p->n_rays, p->capacity, &L);
    char* w = (char*)workspace;
    const int nr = p->n_rays, cap = p->capacity;
    float* d_sigma = bwd_workspace;
    float* d_feats = d_sigma + cap;
    float* fws = d_feats + (int64_t)3 * cap + 32;
    hipLaunchKernelGGL((composite_bwd_kernel<0>), dim3(asd_div_up(nr, RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, AT(float, sigma), AT(float, t0), AT(float, t1),
                       AT(float, feats), AT(int32_t, koff), AT(int32_t, kept), nr, p->bg, AT(float, weights), AT(float, opacity), AT(float, depth), d_comp_rgb, d_rgb_fg,
                       d_opacity, d_depth, d_z_var, (const float*)nullptr, d_sigma, d_feats, d_bg, p->color_act);
    const int rc = asd_field_bwd(p->meta, p->field, p->grid, p->w1d, p->w2d, p->w1f, p->w2f, AT(float, pts), AT(float, enc), AT(float, sigma), cap, AT(int32_t, n_kept),
                                 d_sigma, d_feats, nullptr, nullptr, d_grid, dw1d, dw2d, dw1f, dw2f, fws, stream);
    if (rc != ASD_OK) return rc;
    ASD_LAUNCH_CHECK();
    return ASD_OK;
#undef AT
#undef STEP
}

int asd_occgrid_update(float* occs, int32_t n_cells, const int32_t* cell_idx, const float* occ_new, int32_t n_update,
                       float decay, float occ_thre, uint32_t* occ_bits, uint8_t* binaries, float* scratch,
                       void* stream) {
    ASD_CHECK_ARG(occs && occ_bits && scratch && n_cells > 0 && n_update >= 0, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (n_update > 0) {
        ASD_CHECK_ARG(cell_idx && occ_new, "null update arrays");
        hipLaunchKernelGGL(occ_ema_kernel, dim3(asd_div_up(n_update, 256)), dim3(256), 0, s, occs, cell_idx, occ_new,
                           n_update, decay);
    }
    hipLaunchKernelGGL(occ_mean_kernel, dim3(1), dim3(1024), 0, s, occs, n_cells, scratch);
    hipLaunchKernelGGL(occ_threshold_kernel, dim3(asd_div_up(asd_div_up(n_cells, 32), 256)), dim3(256), 0, s, occs, n_cells,
                       scratch, occ_thre, occ_bits, binaries);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_generate_rays(const float* c2w, const float* focal, int32_t B, int32_t H, int32_t W, int32_t normalize, float* rays_o,
                      float* rays_d, void* stream) {
    ASD_CHECK_ARG(c2w && focal && rays_o && rays_d && B > 0 && H > 0 && W > 0, "bad argument");
    hipLaunchKernelGGL(generate_rays_kernel, dim3(asd_grid_for((int64_t)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, c2w, focal, B, H, W,
                       normalize, rays_o, rays_d);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
