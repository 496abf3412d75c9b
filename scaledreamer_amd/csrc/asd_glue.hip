// asd_glue.hip — the latent-space arithmetic of the ASD guidance between the renderer's image and the scalar loss, as six small
// fused kernels (everything here is elementwise or a per-sample reduction over 4*64*64 values: one pass each, no torch glue):
//   asd_image_prep_fwd/bwd   rgb[B,h,w,3] -> bilinear (align_corners=False) resize to HxW, *2-1, NHWC fp16 padded to 32 channels
//                            (stable_diffusion_asd_guidance.py:196-209 get_latents, :171-175 encode_images' imgs*2-1) and its adjoint
//   asd_latents_fwd          posterior sample * 0.18215 (:176-177), add_noise at t and at t+ (:242-246), written straight into the
//                            UNet's staging buffers in the batch layout of get_eps (:377-394): n_rep copies of x_t, then x_{t+}
//   asd_score_fwd            CFG + Perp-Neg combine (:404-428, threestudio/utils/ops.py:501-511 perpendicular_component), w(t)
//                            (:263-271), nan_to_num (:274), optional clamp (:276-277); grad, loss = 0.5*sum(grad^2)/B (:281-283:
//                            MSE against the detached target), grad_norm
//   asd_latents_bwd          d loss / d moments through the posterior sample (gradient reaches the VAE encoder, :225)
//   asd_prompt_context       view-dependent / Perp-Neg prompt selection (prompt_processors/base.py:82-167, 262-294) written into the
//                            UNet's context buffer in get_eps' batch order, with the Perp-Neg weights
// MVDream (mvdream_asd_guidance.py:181-304) is the same with n_rep = 2, n_neg = 0 and one t per 4-view group.
#include "asd_common.h"

typedef _Float16 half_t;

namespace {

// F.interpolate(mode="bilinear", align_corners=False): src = max(0, scale * (dst + 0.5) - 0.5), scale = in / out in fp32
__device__ __forceinline__ void bilinear_src(int dst, float scale, int n_in, int* i0, int* i1, float* l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int a = (int)s;
    *i0 = a;
    *i1 = a + (a < n_in - 1 ? 1 : 0);
    *l1 = s - (float)a;
}

__global__ __launch_bounds__(256) void image_prep_fwd_kernel(const float* __restrict__ rgb, int B, int h, int w, int H, int W,
                                                             half_t* __restrict__ x) {
    const size_t total = (size_t)B * H * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / ((size_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(Y, sy, h, &y0, &y1, &ly);
        bilinear_src(X, sx, w, &x0, &x1, &lx);
        const float* p = rgb + (size_t)b * h * w * 3;
        half_t o[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] = (half_t)0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = p[((size_t)y0 * w + x0) * 3 + c], v01 = p[((size_t)y0 * w + x1) * 3 + c];
            const float v10 = p[((size_t)y1 * w + x0) * 3 + c], v11 = p[((size_t)y1 * w + x1) * 3 + c];
            // ATen's order: (1-ly) * ((1-lx) v00 + lx v01) + ly * ((1-lx) v10 + lx v11)
            const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            o[c] = (half_t)(v * 2.0f - 1.0f);
        }
        uint4* dst = reinterpret_cast<uint4*>(x + i * 32);
        const uint4* src = reinterpret_cast<const uint4*>(o);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = src[q];
    }
}

// adjoint as a gather: one wave per source pixel; its lanes share the destination pixels whose footprint contains it (up to
// ~2/scale + 2 rows x columns: 18 x 18 at 64 -> 512), each lane reads the three channels of a destination pixel with one 8-byte load
__global__ __launch_bounds__(256) void image_prep_bwd_kernel(const half_t* __restrict__ dx, int B, int h, int w, int H, int W,
                                                             float* __restrict__ d_rgb) {
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pix >= B * h * w) return;
    const int xs = pix % w, ys = (pix / w) % h, b = pix / (w * h);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    // destination rows / columns that can touch source index s: src in (s-1, s+1)  =>  dst in ((s-0.5)/scale-0.5 .. (s+1.5)/scale-0.5)
    int Y0 = (int)floorf(((float)ys - 0.5f) / sy - 0.5f) - 1, Y1 = (int)ceilf(((float)ys + 1.5f) / sy - 0.5f) + 1;
    int X0 = (int)floorf(((float)xs - 0.5f) / sx - 0.5f) - 1, X1 = (int)ceilf(((float)xs + 1.5f) / sx - 0.5f) + 1;
    Y0 = Y0 < 0 ? 0 : Y0; X0 = X0 < 0 ? 0 : X0; Y1 = Y1 > H - 1 ? H - 1 : Y1; X1 = X1 > W - 1 ? W - 1 : X1;
    const int nx = X1 - X0 + 1, n = nx * (Y1 - Y0 + 1);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = lane; k < n; k += 64) {
        const int Y = Y0 + k / nx, X = X0 + k % nx;
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_src(Y, sy, h, &y0, &y1, &ly);
        bilinear_src(X, sx, w, &x0, &x1, &lx);
        const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
        const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
        const float wt = wy * wx;
        if (wt == 0.f) continue;
        const uint2 raw = *reinterpret_cast<const uint2*>(dx + (((size_t)b * H + Y) * W + X) * 32);
        const half_t* v = reinterpret_cast<const half_t*>(&raw);
        a0 = fmaf(wt, (float)v[0], a0); a1 = fmaf(wt, (float)v[1], a1); a2 = fmaf(wt, (float)v[2], a2);
    }
    a0 = asd_wave_sum(a0); a1 = asd_wave_sum(a1); a2 = asd_wave_sum(a2);
    if (lane == 0) {
        float* o = d_rgb + (size_t)pix * 3;
        o[0] = 2.0f * a0; o[1] = 2.0f * a1; o[2] = 2.0f * a2;     // d/d rgb of (2 rgb - 1)
    }
}

// one thread per latent pixel (b, y, x): all C channels
template <int C>
__global__ __launch_bounds__(256) void latents_fwd_kernel(const float* __restrict__ moments /*[B,hw,2C]*/, const float* __restrict__ post_noise /*[B,C,hw]*/,
                                                          const float* __restrict__ noise /*[B,C,hw]*/, const int64_t* __restrict__ t,
                                                          const int64_t* __restrict__ t_plus, const float* __restrict__ alphas, int B, int hw,
                                                          float scaling, int n_rep, float* __restrict__ latents /*[B,C,hw]*/,
                                                          half_t* __restrict__ unet_x /*[(n_rep+1)B,hw,32]*/, float* __restrict__ unet_t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < (n_rep + 1) * B) unet_t[i] = (float)(i < n_rep * B ? t[i % B] : t_plus[i - n_rep * B]);
    if (i >= B * hw) return;
    const int b = i / hw, p = i - b * hw;
    const float a0 = alphas[t[b]], a1 = alphas[t_plus[b]];
    const float sa0 = sqrtf(a0), sn0 = sqrtf(1.f - a0), sa1 = sqrtf(a1), sn1 = sqrtf(1.f - a1);
    half_t x0[32], x1[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { x0[c] = (half_t)0.f; x1[c] = (half_t)0.f; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float mean = moments[(size_t)i * 2 * C + c];
        float lv = moments[(size_t)i * 2 * C + C + c];
        lv = fminf(fmaxf(lv, -30.f), 20.f);
        const size_t q = ((size_t)b * C + c) * hw + p;
        const float z = (mean + expf(0.5f * lv) * post_noise[q]) * scaling;
        latents[q] = z;
        const float e = noise[q];
        x0[c] = (half_t)(sa0 * z + sn0 * e);
        x1[c] = (half_t)(sa1 * z + sn1 * e);
    }
    const uint4* s0 = reinterpret_cast<const uint4*>(x0);
    const uint4* s1 = reinterpret_cast<const uint4*>(x1);
    for (int r = 0; r <= n_rep; ++r) {
        uint4* dst = reinterpret_cast<uint4*>(unet_x + ((size_t)(r * B + b) * hw + p) * 32);
        const uint4* src = r < n_rep ? s0 : s1;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = src[q];
    }
}

__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
    v = asd_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k];
    return s;
}

__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.f;
    if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return v;
}

// one block of 1024 threads per sample.  eps [(n_rep+1)B, hw, C] in the batch order text | uncond | neg (2 per sample, interleaved) | second
template <int C>
__global__ __launch_bounds__(1024) void score_fwd_kernel(const float* __restrict__ eps, int B, int hw, int n_neg, const float* __restrict__ neg_w,
                                                         float guidance_scale, const int64_t* __restrict__ t, const float* __restrict__ alphas,
                                                         int weighting, float grad_clip, float* __restrict__ grad /*[B,C,hw]*/,
                                                         float* __restrict__ sumsq /*[B]*/) {
    __shared__ float sh[16];
    const int b = blockIdx.x, n = hw * C;
    const float* e_text = eps + (size_t)b * n;
    const float* e_unc = eps + (size_t)(B + b) * n;
    const float* e_sec = eps + (size_t)((2 + n_neg) * B + b) * n;
    float dot_pp = 0.f, dot_np[2] = {0.f, 0.f};
    if (n_neg > 0) {
        for (int i = threadIdx.x; i < n; i += 1024) {
            const float u = e_unc[i], p = e_text[i] - u;
            dot_pp = fmaf(p, p, dot_pp);
            for (int k = 0; k < n_neg; ++k) dot_np[k] = fmaf(eps[(size_t)(2 * B + b * n_neg + k) * n + i] - u, p, dot_np[k]);
        }
        dot_pp = block_sum_1024(dot_pp, sh);
        for (int k = 0; k < n_neg; ++k) dot_np[k] = block_sum_1024(dot_np[k], sh);
    }
    const float a = alphas[t[b]];
    const float wt = weighting == 0 ? 1.f - a : (weighting == 1 ? 1.f : sqrtf(a) * (1.f - a));
    float coef[2] = {0.f, 0.f}, wk[2] = {0.f, 0.f};
    for (int k = 0; k < n_neg; ++k) {
        coef[k] = dot_np[k] / fmaxf(dot_pp, 1e-6f);
        wk[k] = neg_w[b * n_neg + k];
    }
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float u = e_unc[i], p = e_text[i] - u;
        float acc = 0.f;
        for (int k = 0; k < n_neg; ++k) {
            const float en = eps[(size_t)(2 * B + b * n_neg + k) * n + i] - u;
            acc += wk[k] * (en - coef[k] * p);               // w_k * perpendicular_component(eps_neg_k, eps_pos)
        }
        float g = nan_to_num(wt * ((p + acc) * guidance_scale + u - e_sec[i]));
        if (grad_clip > 0.f) g = fminf(fmaxf(g, -grad_clip), grad_clip);
        const int pix = i / C, c = i - pix * C;              // eps is NHWC, grad is NCHW like the latents
        grad[((size_t)b * C + c) * hw + pix] = g;
        ss = fmaf(g, g, ss);
    }
    ss = block_sum_1024(ss, sh);
    if (threadIdx.x == 0) sumsq[b] = ss;
}

__global__ void score_reduce_kernel(const float* __restrict__ sumsq, int B, float* __restrict__ out2) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += sumsq[b];
    out2[0] = 0.5f * s / (float)B;      // 0.5 * mse_loss(latents, (latents - grad).detach(), "sum") / batch_size
    out2[1] = sqrtf(s);                 // grad.norm()
}

template <int C>
__global__ __launch_bounds__(256) void latents_bwd_kernel(const float* __restrict__ grad /*[B,C,hw]*/, const float* __restrict__ moments,
                                                          const float* __restrict__ post_noise, const float* __restrict__ upstream, int B, int hw,
                                                          float scaling, float* __restrict__ d_moments /*[B,hw,2C]*/) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * hw) return;
    const int b = i / hw, p = i - b * hw;
    const float up = (upstream ? upstream[0] : 1.f) / (float)B * scaling;    // d loss / d z = grad / B
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const size_t q = ((size_t)b * C + c) * hw + p;
        const float g = grad[q] * up;
        const float lv = moments[(size_t)i * 2 * C + C + c];
        d_moments[(size_t)i * 2 * C + c] = g;
        d_moments[(size_t)i * 2 * C + C + c] = (lv > -30.f && lv < 20.f) ? g * post_noise[q] * 0.5f * expf(0.5f * lv) : 0.f;
    }
}


// t+ = clamp(t + (long)(u * clamp(plus_ratio * (t - min_step), 0, T - 1 - t)), 1, T - 1), u = 1 without plus_random
// (stable_diffusion_asd_guidance.py:294-316): the tensor-op form's float32 arithmetic — the product in float32, clamp as min(max(x, lo), hi),
// truncation toward zero — in one launch instead of a dozen one-element ones
__global__ void timestep_plus_kernel(const long long* __restrict__ t, const float* __restrict__ u, int n, long long min_step, long long T, float plus_ratio,
                                     long long* __restrict__ t_plus) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long ti = t[i];
    float room = plus_ratio * (float)(ti - min_step);
    room = fminf(fmaxf(room, 0.f), (float)(T - ti - 1));
    if (u) room = room * u[i];
    long long tp = ti + (long long)room;
    tp = tp < 1 ? 1 : (tp > T - 1 ? T - 1 : tp);
    t_plus[i] = tp;
}

// ---- loss assembly: weighted scalar terms + the per-ray regularisers, one launch forward and one backward --------------------------
// scaledreamer.py:62-126 (training_step): loss = sum_j lambda_j loss_j + lambda_sparsity mean(sqrt(opacity^2 + 0.01))
//   + lambda_opaque BCE(clamp(opacity, 1e-3, 1 - 1e-3)) (threestudio/utils/ops.py:365-369, the clamped opacity as input AND target, both
//   differentiated) + lambda_z_variance mean(z_variance[opacity > 0.5]).  As tensor ops this is ~25 launches of a few hundred bytes between
//   the score and the VAE backward, issued while the host has nothing queued ahead (the autograd pass starts here): the device waits.
#define ASD_LOSS_MAX_TERMS 8
struct LossTerms {
    const float* value[ASD_LOSS_MAX_TERMS];
    float weight[ASD_LOSS_MAX_TERMS];
    int n;
};

// out[5] = {total, sparsity, opaque, z_variance, number of rays with opacity > 0.5}; a term with lambda <= 0 is not evaluated (0)
__global__ __launch_bounds__(1024) void loss_tail_fwd_kernel(LossTerms terms, const float* __restrict__ opacity, const float* __restrict__ z_var,
                                                              long long n, float lam_s, float lam_o, float lam_z, float* __restrict__ out) {
    __shared__ float red[16];
    float ss = 0.f, so = 0.f, sz = 0.f, cnt = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float o = opacity[i];
        if (lam_s > 0.f) ss += sqrtf(o * o + 0.01f);
        if (lam_o > 0.f) {
            const float x = fminf(fmaxf(o, 1.0e-3f), 1.0f - 1.0e-3f);
            so -= x * logf(x) + (1.f - x) * logf(1.f - x);
        }
        if (lam_z > 0.f && o > 0.5f) { sz += z_var[i]; cnt += 1.f; }
    }
    ss = block_sum_1024(ss, red);
    so = block_sum_1024(so, red);
    sz = block_sum_1024(sz, red);
    cnt = block_sum_1024(cnt, red);
    if (threadIdx.x == 0) {
        const float inv_n = 1.f / (float)n;
        const float sparsity = lam_s > 0.f ? ss * inv_n : 0.f, opaque = lam_o > 0.f ? so * inv_n : 0.f;
        const float zv = lam_z > 0.f ? sz / cnt : 0.f;          // no ray above 0.5: 0 / 0 = NaN, the mean of an empty selection
        float total = 0.f;
        for (int j = 0; j < terms.n; ++j) total += terms.weight[j] * terms.value[j][0];
        if (lam_s > 0.f) total += sparsity * lam_s;
        if (lam_o > 0.f) total += opaque * lam_o;
        if (lam_z > 0.f) total += zv * lam_z;
        out[0] = total; out[1] = sparsity; out[2] = opaque; out[3] = zv; out[4] = cnt;
    }
}

__global__ __launch_bounds__(256) void loss_tail_bwd_kernel(LossTerms terms, const float* __restrict__ upstream, const float* __restrict__ opacity,
                                                             long long n, float lam_s, float lam_o, float lam_z, const float* __restrict__ fwd_out,
                                                             float* __restrict__ d_terms, float* __restrict__ d_opacity, float* __restrict__ d_zvar) {
    const float up = upstream ? upstream[0] : 1.f;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < terms.n) d_terms[i] = up * terms.weight[i];
    if (i >= n) return;
    const float o = opacity[i], inv_n = 1.f / (float)n;
    float g = 0.f;
    if (lam_s > 0.f) g += lam_s * inv_n * o / sqrtf(o * o + 0.01f);
    if (lam_o > 0.f && o >= 1.0e-3f && o <= 1.0f - 1.0e-3f) g += lam_o * inv_n * (logf(1.f - o) - logf(o));
    d_opacity[i] = up * g;
    if (d_zvar) d_zvar[i] = (lam_z > 0.f && o > 0.5f) ? up * lam_z / fwd_out[4] : 0.f;
}

// ---- view-dependent prompt selection, written straight into the UNet's context buffer ---------------------------------------------
// prompt_processors/base.py:82-167 (get_text_embeddings_perp_neg) and :262-294 (direction by thresholds: side < front < back <
// overhead), as one launch: a block owns one output token row.  Row blends keep the reference's fp32 arithmetic
// (r * a + (1 - r) * b: two products and a sum, no fma) before the fp16 rounding of the UNet input.
struct PromptParams {
    float overhead_thr, front_thr, back_thr;
    float f_sb[3], f_fsb[3], f_fs[3], f_sf[3];
};

__device__ __forceinline__ float shifted_azimuth(float az) {      // (az + 180) % 360 - 180 with Python's sign convention
    float r = fmodf(az + 180.f, 360.f);
    if (r != 0.f && r < 0.f) r += 360.f;
    return r - 180.f;
}
__device__ __forceinline__ float exp_decay(const float* f, float r) { return f[0] * expf(-f[1] * r) + f[2]; }

// segments of the output batch: layout 0 = [text | uncond | text], layout 1 = [pos | uncond | neg1_0, neg2_0, neg1_1, ... | pos]
__global__ __launch_bounds__(128) void prompt_context_kernel(const float* __restrict__ text_vd, const float* __restrict__ uncond_vd, int n_dir,
                                                             int n_tok, int dim, const float* __restrict__ elevation,
                                                             const float* __restrict__ azimuth, int B, int layout, PromptParams pp,
                                                             float neg_scale, half_t* __restrict__ ctx, int ctx_stride,
                                                             float* __restrict__ neg_w) {
    const int tok = blockIdx.x % n_tok, slot = blockIdx.x / n_tok;          // slot: sample of the UNet batch
    int b, kind;                                                             // kind 0 text / pos, 1 uncond, 2 neg1, 3 neg2
    if (layout == 0) { kind = slot / B == 1 ? 1 : 0; b = slot % B; }
    else if (slot < 2 * B) { kind = slot / B; b = slot % B; }
    else if (slot < 4 * B) { b = (slot - 2 * B) >> 1; kind = 2 + ((slot - 2 * B) & 1); }
    else { kind = 0; b = slot - 4 * B; }
    const float azi = shifted_azimuth(azimuth[b]);
    int idx = 0;
    if (n_dir > 1) {
        if (azi > -pp.front_thr && azi < pp.front_thr) idx = 1;
        if (azi > 180.f - pp.back_thr || azi < -180.f + pp.back_thr) idx = 2;
        if (elevation[b] > pp.overhead_thr) idx = 3;
    }
    const float a = fabsf(azi);
    const bool over = idx == 3, is_front = a < 90.f;
    const float r_f = 1.f - a / 90.f, r_b = 2.0f - a / 90.f;
    // row = wa * A + wb * Bm (wb == 0: plain copy of A)
    const float* A;
    const float* Bm = nullptr;
    float wa = 1.f, wb = 0.f;
    const size_t row = (size_t)tok * dim, tab = (size_t)n_tok * dim;
    if (kind == 1 || (over && kind >= 2)) A = uncond_vd + idx * tab;
    else if (layout == 0 || n_dir == 1) A = text_vd + idx * tab;
    else if (kind == 0) {
        if (over) A = text_vd + 3 * tab;
        else if (is_front) { A = text_vd + 1 * tab; Bm = text_vd; wa = r_f; wb = 1.f - r_f; }            // front <-> side
        else { A = text_vd; Bm = text_vd + 2 * tab; wa = r_b; wb = 1.f - r_b; }                           // side <-> back
    } else if (kind == 2) A = is_front ? text_vd + 1 * tab : text_vd;
    else A = is_front ? text_vd : text_vd + 1 * tab;
    half_t* dst = ctx + ((size_t)slot * ctx_stride + tok) * dim;
    for (int c = threadIdx.x * 4; c < dim; c += 128 * 4) {
        const float4 va = *(const float4*)(A + row + c);
        float4 v = va;
        if (Bm) {
            const float4 vb = *(const float4*)(Bm + row + c);
            v.x = wa * va.x + wb * vb.x; v.y = wa * va.y + wb * vb.y; v.z = wa * va.z + wb * vb.z; v.w = wa * va.w + wb * vb.w;
        }
        half_t o[4] = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
        *(uint2*)(dst + c) = *(const uint2*)o;
    }
    if (neg_w && layout == 1 && tok == 0 && kind == 0 && slot < B && threadIdx.x == 0) {
        float w1 = 0.f, w2 = 0.f;
        if (!over) {
            w1 = is_front ? -exp_decay(pp.f_fs, r_f) : -exp_decay(pp.f_sb, r_b);
            w2 = is_front ? -exp_decay(pp.f_sf, 1.f - r_f) : -exp_decay(pp.f_fsb, r_b);
        }
        neg_w[2 * b] = w1 * neg_scale;
        neg_w[2 * b + 1] = w2 * neg_scale;
    }
}

}  // namespace

extern "C" {

int asd_image_prep_fwd(const float* rgb, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, void* x_nhwc32, void* stream) {
    ASD_CHECK_ARG(rgb && x_nhwc32 && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad argument");
    hipLaunchKernelGGL(image_prep_fwd_kernel, dim3(asd_grid_for((int64_t)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, rgb, B, h, w, H, W,
                       (half_t*)x_nhwc32);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_image_prep_bwd(const void* dx_nhwc32, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, float* d_rgb, void* stream) {
    ASD_CHECK_ARG(dx_nhwc32 && d_rgb && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "bad argument");
    hipLaunchKernelGGL(image_prep_bwd_kernel, dim3(asd_div_up((int64_t)B * h * w, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)dx_nhwc32, B, h, w, H, W, d_rgb);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_latents_fwd(const float* moments_nhwc, const float* post_noise, const float* noise, const int64_t* t, const int64_t* t_plus,
                    const float* alphas_cumprod, int32_t B, int32_t C, int32_t hl, int32_t wl, float scaling, int32_t n_rep, float* latents,
                    void* unet_x, float* unet_t, void* stream) {
    ASD_CHECK_ARG(moments_nhwc && post_noise && noise && t && t_plus && alphas_cumprod && latents && unet_x && unet_t, "null argument");
    ASD_CHECK_ARG(C == 4 && B > 0 && hl > 0 && wl > 0 && n_rep >= 1, "latents: 4 channels, n_rep >= 1");
    const int n = B * hl * wl > (n_rep + 1) * B ? B * hl * wl : (n_rep + 1) * B;
    hipLaunchKernelGGL((latents_fwd_kernel<4>), dim3(asd_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, moments_nhwc, post_noise, noise, t,
                       t_plus, alphas_cumprod, B, hl * wl, scaling, n_rep, latents, (half_t*)unet_x, unet_t);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_score_fwd(const float* eps_nhwc, int32_t B, int32_t C, int32_t hw, int32_t n_neg, const float* neg_w, float guidance_scale,
                  const int64_t* t, const float* alphas_cumprod, int32_t weighting, float grad_clip, float* grad, float* sumsq,
                  float* loss_and_norm, void* stream) {
    ASD_CHECK_ARG(eps_nhwc && t && alphas_cumprod && grad && sumsq && loss_and_norm, "null argument");
    ASD_CHECK_ARG(C == 4 && B > 0 && hw > 0 && n_neg >= 0 && n_neg <= 2 && (n_neg == 0 || neg_w) && weighting >= 0 && weighting <= 2,
                  "score: 4 channels, at most 2 negative prompts per sample");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((score_fwd_kernel<4>), dim3(B), dim3(1024), 0, s, eps_nhwc, B, hw, n_neg, neg_w, guidance_scale, t, alphas_cumprod, weighting,
                       grad_clip, grad, sumsq);
    hipLaunchKernelGGL(score_reduce_kernel, dim3(1), dim3(1), 0, s, sumsq, B, loss_and_norm);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_latents_bwd(const float* grad, const float* moments_nhwc, const float* post_noise, const float* upstream, int32_t B, int32_t C, int32_t hl,
                    int32_t wl, float scaling, float* d_moments_nhwc, void* stream) {
    ASD_CHECK_ARG(grad && moments_nhwc && post_noise && d_moments_nhwc && C == 4 && B > 0 && hl > 0 && wl > 0, "bad argument");
    hipLaunchKernelGGL((latents_bwd_kernel<4>), dim3(asd_div_up(B * hl * wl, 256)), dim3(256), 0, (hipStream_t)stream, grad, moments_nhwc, post_noise,
                       upstream, B, hl * wl, scaling, d_moments_nhwc);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_prompt_context(const float* text_vd, const float* uncond_vd, int32_t n_dir, int32_t n_tok, int32_t dim, const float* elevation,
                       const float* azimuth, int32_t batch, int32_t layout, const float* params15, float neg_scale, void* context_f16,
                       int32_t ctx_stride, float* neg_w, void* stream) {
    ASD_CHECK_ARG(text_vd && uncond_vd && elevation && azimuth && params15 && context_f16, "null argument");
    ASD_CHECK_ARG((n_dir == 1 || n_dir == 4) && n_tok > 0 && dim > 0 && dim % 4 == 0 && batch > 0 && ctx_stride >= n_tok, "bad sizes");
    ASD_CHECK_ARG(layout == 0 || (layout == 1 && n_dir == 4 && neg_w), "layout 1 (Perp-Neg) needs the four view-dependent embeddings and neg_w");
    PromptParams pp;
    pp.overhead_thr = params15[0]; pp.front_thr = params15[1]; pp.back_thr = params15[2];
    for (int i = 0; i < 3; ++i) { pp.f_sb[i] = params15[3 + i]; pp.f_fsb[i] = params15[6 + i]; pp.f_fs[i] = params15[9 + i]; pp.f_sf[i] = params15[12 + i]; }
    const int slots = (layout == 0 ? 3 : 5) * batch;
    hipLaunchKernelGGL(prompt_context_kernel, dim3(slots * n_tok), dim3(128), 0, (hipStream_t)stream, text_vd, uncond_vd, n_dir, n_tok, dim, elevation,
                       azimuth, batch, layout, pp, neg_scale, (half_t*)context_f16, ctx_stride, neg_w);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_timestep_plus(const int64_t* t, const float* u, int32_t n, int64_t min_step, int64_t num_train_timesteps, float plus_ratio, int64_t* t_plus,
                      void* stream) {
    ASD_CHECK_ARG(t && t_plus && n > 0 && num_train_timesteps > 1 && plus_ratio >= 0.f, "bad argument");
    hipLaunchKernelGGL(timestep_plus_kernel, dim3(asd_div_up(n, 64)), dim3(64), 0, (hipStream_t)stream, (const long long*)t, u, n, (long long)min_step,
                       (long long)num_train_timesteps, plus_ratio, (long long*)t_plus);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_loss_tail_fwd(const float* const* terms, const float* weights, int32_t n_terms, const float* opacity, const float* z_variance,
                      int64_t n_rays, float lambda_sparsity, float lambda_opaque, float lambda_z_variance, float* out5, void* stream) {
    ASD_CHECK_ARG(opacity && out5 && n_rays > 0 && n_terms >= 0 && n_terms <= ASD_LOSS_MAX_TERMS, "bad argument");
    ASD_CHECK_ARG(n_terms == 0 || (terms && weights), "null term table");
    ASD_CHECK_ARG(!(lambda_z_variance > 0.f) || z_variance, "lambda_z_variance > 0 needs z_variance");
    LossTerms t;
    t.n = n_terms;
    for (int j = 0; j < ASD_LOSS_MAX_TERMS; ++j) { t.value[j] = j < n_terms ? terms[j] : nullptr; t.weight[j] = j < n_terms ? weights[j] : 0.f; }
    for (int j = 0; j < n_terms; ++j) ASD_CHECK_ARG(terms[j], "null term");
    hipLaunchKernelGGL(loss_tail_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, t, opacity, z_variance, (long long)n_rays, lambda_sparsity,
                       lambda_opaque, lambda_z_variance, out5);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_loss_tail_bwd(const float* upstream, const float* weights, int32_t n_terms, const float* opacity, int64_t n_rays, float lambda_sparsity,
                      float lambda_opaque, float lambda_z_variance, const float* out5, float* d_terms, float* d_opacity, float* d_z_variance,
                      void* stream) {
    ASD_CHECK_ARG(opacity && out5 && d_opacity && n_rays > 0 && n_terms >= 0 && n_terms <= ASD_LOSS_MAX_TERMS, "bad argument");
    ASD_CHECK_ARG(n_terms == 0 || (weights && d_terms), "null term table");
    LossTerms t;
    t.n = n_terms;
    for (int j = 0; j < ASD_LOSS_MAX_TERMS; ++j) { t.value[j] = nullptr; t.weight[j] = j < n_terms ? weights[j] : 0.f; }
    hipLaunchKernelGGL(loss_tail_bwd_kernel, dim3(asd_div_up((int)n_rays, 256)), dim3(256), 0, (hipStream_t)stream, t, upstream, opacity,
                       (long long)n_rays, lambda_sparsity, lambda_opaque, lambda_z_variance, out5, d_terms, d_opacity, d_z_variance);
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
