// optim.hip — the optimizer step of the ASD loop as multi-tensor fused kernels:
//   asd_adamw_f32   torch.optim.AdamW / Adam as the reference configures them (asd_sd_nerf.yaml:110-125: AdamW, betas (0, 0.99),
//                   eps 1e-15, five parameter groups addressed by attribute path, threestudio/systems/utils.py:25-53)
//   asd_adan_f32    Adan with the update rule of threestudio/systems/optimizers.py:200-315 (triplane config)
// One launch updates up to ASD_OPT_MAX_TENSORS tensors of any sizes and per-tensor hyper-parameters (the table travels in the
// kernel arguments, like torch's multi_tensor_apply): each block walks 4096-element chunks of the flattened tensor list.  Every
// element is read and written exactly once (p, g, m, v [, d, g_prev]): 28 B per AdamW element, HBM-bound, nothing else to tune.
// Dense on purpose: with weight decay and second-moment decay the reference's update touches every entry of the hash table every
// step (an entry with zero gradient still shrinks by lr*wd and its v decays), so a touched-entries-only update would not be the
// reference's optimizer.
#include "asd_common.h"

namespace {

constexpr int CHUNK = 4096;

// torch's lerp(a, b, w): a + w (b - a) for w < 0.5, b - (b - a)(1 - w) otherwise (exact at w = 1: beta1 = 0 gives m = g)
__device__ __forceinline__ float lerp_t(float a, float b, float w) {
    const float d = b - a;
    return w < 0.5f ? fmaf(w, d, a) : fmaf(-d, 1.f - w, b);
}

struct AdamTable {
    asd_opt_tensor t[ASD_OPT_MAX_TENSORS];
    int32_t chunk_start[ASD_OPT_MAX_TENSORS + 1];   // prefix sums of ceil(n / CHUNK)
    int32_t n_tensors;
};

__device__ __forceinline__ int find_tensor(const int32_t* start, int n, int chunk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// mode 0: AdamW (decoupled decay p *= 1 - lr wd), mode 1: Adam (L2: g += wd p)
__global__ __launch_bounds__(256) void adamw_kernel(const AdamTable tab, float beta1, float beta2, float eps, int adam_l2) {
    const int total = tab.chunk_start[tab.n_tensors];
    for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int ti = find_tensor(tab.chunk_start, tab.n_tensors, chunk);
        const asd_opt_tensor& t = tab.t[ti];
        const int64_t base = (int64_t)(chunk - tab.chunk_start[ti]) * CHUNK;
        const float lr = t.lr, wd = t.weight_decay, bc1 = t.bias_correction1, bc2_sqrt = t.bias_correction2_sqrt;
        const float step_size = lr / bc1, decay = (float)(1.0 - (double)lr * (double)wd), w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
        for (int k = threadIdx.x * 4; k < CHUNK; k += 256 * 4) {
            const int64_t i = base + k;
            if (i >= t.n) break;
            if (i + 3 < t.n) {
                float4 p = *(float4*)(t.p + i), m = *(float4*)(t.m + i), v = *(float4*)(t.v + i);
                const float4 g4 = *(const float4*)(t.g + i);
                float* pp = &p.x; float* mm = &m.x; float* vv = &v.x; const float* gg = &g4.x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g = gg[r];
                    // explicit fused multiply-adds where ATen's kernels contract (add(alpha), lerp, addcmul, addcdiv): with g ~ -wd p the
                    // L2 term cancels and an unfused rounding would show up as a percent-level difference of that element's step
                    if (adam_l2) g = fmaf(wd, pp[r], g); else pp[r] *= decay;
                    mm[r] = lerp_t(mm[r], g, w1);
                    vv[r] = fmaf(w2 * g, g, beta2 * vv[r]);
                    const float denom = sqrtf(vv[r]) / bc2_sqrt + eps;
                    pp[r] = fmaf(-step_size, mm[r] / denom, pp[r]);
                }
                *(float4*)(t.p + i) = p; *(float4*)(t.m + i) = m; *(float4*)(t.v + i) = v;
            } else {
                for (int64_t j = i; j < t.n; ++j) {
                    float g = t.g[j], p = t.p[j];
                    if (adam_l2) g = fmaf(wd, p, g); else p *= decay;
                    const float m = lerp_t(t.m[j], g, w1);
                    const float v = fmaf(w2 * g, g, beta2 * t.v[j]);
                    t.m[j] = m; t.v[j] = v;
                    t.p[j] = fmaf(-step_size, m / (sqrtf(v) / bc2_sqrt + eps), p);
                }
            }
        }
    }
}

// Adan; t.m = exp_avg, t.v = exp_avg_diff, t.n2 = exp_avg_sq, t.prev = neg_pre_grad (= -(previous clipped gradient))
__global__ __launch_bounds__(256) void adan_kernel(const AdamTable tab, float b1, float b2, float b3, float eps, float clip, int no_prox) {
    const int total = tab.chunk_start[tab.n_tensors];
    for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int ti = find_tensor(tab.chunk_start, tab.n_tensors, chunk);
        const asd_opt_tensor& t = tab.t[ti];
        const int64_t base = (int64_t)(chunk - tab.chunk_start[ti]) * CHUNK;
        const float lr = t.lr, wd = t.weight_decay;
        const float c1 = lr / t.bias_correction1, c2 = lr * b2 / t.bias_correction2, bc3_sqrt = t.bias_correction2_sqrt;
        for (int k = threadIdx.x; k < CHUNK; k += 256) {
            const int64_t i = base + k;
            if (i >= t.n) break;
            const float g = t.g[i] * clip;
            const float d = t.prev[i] + g;                    // g - g_prev
            const float m = t.m[i] * b1 + (1.f - b1) * g;
            const float v = t.v[i] * b2 + (1.f - b2) * d;
            const float u = d * b2 + g;                       // g + b2 d
            const float n = t.n2[i] * b3 + (1.f - b3) * u * u;
            const float den = sqrtf(n) / bc3_sqrt + eps;
            float p = t.p[i];
            if (no_prox) p *= 1.f - lr * wd;
            p = p - c1 * (m / den);
            p = p - c2 * (v / den);
            if (!no_prox) p = p / (1.f + lr * wd);
            t.g[i] = g;                                       // the reference clips the gradient in place
            t.p[i] = p; t.m[i] = m; t.v[i] = v; t.n2[i] = n; t.prev[i] = -g;
        }
    }
}

int fill(AdamTable& tab, const asd_opt_tensor* t, int n) {
    tab.n_tensors = n;
    tab.chunk_start[0] = 0;
    for (int i = 0; i < n; ++i) {
        tab.t[i] = t[i];
        tab.chunk_start[i + 1] = tab.chunk_start[i] + (int32_t)((t[i].n + CHUNK - 1) / CHUNK);
    }
    return tab.chunk_start[n];
}

}  // namespace

extern "C" {

int asd_adamw_f32(const asd_opt_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float eps, int32_t adam_l2, void* stream) {
    ASD_CHECK_ARG(tensors && n_tensors > 0, "null argument");
    for (int i0 = 0; i0 < n_tensors; i0 += ASD_OPT_MAX_TENSORS) {
        const int n = n_tensors - i0 < ASD_OPT_MAX_TENSORS ? n_tensors - i0 : ASD_OPT_MAX_TENSORS;
        AdamTable tab;
        for (int i = 0; i < n; ++i) ASD_CHECK_ARG(tensors[i0 + i].p && tensors[i0 + i].g && tensors[i0 + i].m && tensors[i0 + i].v && tensors[i0 + i].n > 0, "null tensor");
        const int chunks = fill(tab, tensors + i0, n);
        hipLaunchKernelGGL(adamw_kernel, dim3(chunks < 4096 ? chunks : 4096), dim3(256), 0, (hipStream_t)stream, tab, beta1, beta2, eps, adam_l2);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

int asd_adan_f32(const asd_opt_tensor* tensors, int32_t n_tensors, float beta1, float beta2, float beta3, float eps, float clip, int32_t no_prox,
                 void* stream) {
    ASD_CHECK_ARG(tensors && n_tensors > 0, "null argument");
    for (int i0 = 0; i0 < n_tensors; i0 += ASD_OPT_MAX_TENSORS) {
        const int n = n_tensors - i0 < ASD_OPT_MAX_TENSORS ? n_tensors - i0 : ASD_OPT_MAX_TENSORS;
        AdamTable tab;
        for (int i = 0; i < n; ++i) {
            const asd_opt_tensor& t = tensors[i0 + i];
            ASD_CHECK_ARG(t.p && t.g && t.m && t.v && t.n2 && t.prev && t.n > 0, "null tensor");
        }
        const int chunks = fill(tab, tensors + i0, n);
        hipLaunchKernelGGL(adan_kernel, dim3(chunks < 4096 ? chunks : 4096), dim3(256), 0, (hipStream_t)stream, tab, beta1, beta2, beta3, eps, clip, no_prox);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
