// trifield.hip — the fused field of `Triplane-transformer-sdf` (custom/amortized/models/geometry/triplane_transformer.py:139-240): contract ->
// three bilinear plane lookups (sample_from_planes, geometry/utils.py:81-93) concatenated to 96 features -> two VanillaMLP heads
// 96 -> 64 -> 64 -> 1 | 3 (no biases, ReLU) -> sdf + bias -> finite-difference sdf_grad, as ONE kernel forward and one pass backward —
// instead of a sampler launch plus library GEMMs that keep ~1.2 KB of autograd state per evaluation (245 GB at 256 x 256 x 4 views: the
// reference-shaped path needs activation checkpointing there).
//
// One thread per sample.  The 96 features are never held: layer 1 is evaluated TRANSPOSED — the lookup produces four channels at a time and
// each quad is multiplied into all 64 hidden units (weights W1^T [96][64]: wave-uniform rows, scalar loads) — so a thread keeps 64 (forward,
// sdf head) to 128 (centre point: both heads at once) accumulators; layers 2 and 3 stream over the second hidden layer.  The backward pass
// re-gathers the features (nothing but the points is saved), leaves the rows of the four weight-gradient products (ENC, H1, DA1, DA2) in a
// chunk workspace, forms the feature gradient by the same transposed walk and hands it to a scatter that accumulates in LDS: a tri-plane
// is 3 x 64 x 64 x 32 floats — 1.5 MB that EVERY sample of the step updates; a block owns (plane, 8 channels) as a 128 KB LDS image, walks a
// slice of the rows and adds its image to global memory once.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "trifield_common.h"
#include "trifield_mfma.h"

__device__ __forceinline__ floatx4 tf_quad(const float* __restrict__ planes, const tf_tap& t, int q) {     // channels 4 q .. 4 q + 3 of the plane
    floatx4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int corner = 0; corner < 4; ++corner) {
        if (t.off[corner] < 0) continue;
        const floatx4 s = *(const floatx4*)(planes + t.off[corner] + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = fmaf(t.w[corner], s[r], e[r]);
    }
    return e;
}

// Weight rows are wave-uniform.  As scalar loads they are latency-bound here (82 KB of weights cycle through a 16 KB scalar cache: 13 % VALU
// utilisation measured); instead a row of 64 weights is fetched by FOUR vector loads (lane l holds w[16 g + (l & 15)] — every 16-lane DPP row
// holds the same 16 values, one 64-byte request) and each multiply-add takes its weight from lane K of its own row through the DPP operand
// (row_newbcast:K on gfx90a+): one VALU issue per FMA, no scalar unit, no LDS.
// The multiply-adds are issued as blocks of 16 in ONE asm statement behind an `s_nop 1`: a DPP operand read needs two wait states behind a VALU write
// of that register (a register copy the compiler may place in front of the block) and nothing inside an asm statement is padded by the compiler.
// volatile: never sunk into a lane-divergent branch (the operand reads other lanes: the whole row must be executing).
__device__ __forceinline__ void tf_row(const float* __restrict__ row, float (&wv)[4]) {
    const int l16 = threadIdx.x & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) wv[g] = row[16 * g + l16];
}
// h[16 g + k] += w[16 g + k] * x, k = 0..15, for one 16-lane group of weights
__device__ __forceinline__ void tf_axpy16(float wv, float x, float* h) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]), "+v"(h[9]), "+v"(h[10]), "+v"(h[11]),
                   "+v"(h[12]), "+v"(h[13]), "+v"(h[14]), "+v"(h[15])
                 : "v"(wv), "v"(x));
}
__device__ __forceinline__ void tf_axpy(const float (&wv)[4], float x, float (&h)[TF_H]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) tf_axpy16(wv[g], x, &h[16 * g]);
}
// a[g] += sum_{kk < 4} w[16 g + K0 + kk] * h[16 g + K0 + kk] for the four groups g (four independent chains, interleaved)
template <int K0>
__device__ __forceinline__ void tf_dot4(float (&a)[4], const float (&wv)[4], const float (&h)[TF_H]) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %4, %8 row_newbcast:%c24 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %5, %12 row_newbcast:%c24 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %6, %16 row_newbcast:%c24 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %7, %20 row_newbcast:%c24 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %4, %9 row_newbcast:%c25 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %5, %13 row_newbcast:%c25 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %6, %17 row_newbcast:%c25 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %7, %21 row_newbcast:%c25 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %4, %10 row_newbcast:%c26 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %5, %14 row_newbcast:%c26 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %6, %18 row_newbcast:%c26 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %7, %22 row_newbcast:%c26 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %4, %11 row_newbcast:%c27 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %5, %15 row_newbcast:%c27 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %6, %19 row_newbcast:%c27 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %7, %23 row_newbcast:%c27 row_mask:0xf bank_mask:0xf\n\t"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
                 : "v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]),
                   "v"(h[K0]), "v"(h[K0 + 1]), "v"(h[K0 + 2]), "v"(h[K0 + 3]), "v"(h[16 + K0]), "v"(h[16 + K0 + 1]), "v"(h[16 + K0 + 2]), "v"(h[16 + K0 + 3]),
                   "v"(h[32 + K0]), "v"(h[32 + K0 + 1]), "v"(h[32 + K0 + 2]), "v"(h[32 + K0 + 3]), "v"(h[48 + K0]), "v"(h[48 + K0 + 1]), "v"(h[48 + K0 + 2]), "v"(h[48 + K0 + 3]),
                   "n"(K0), "n"(K0 + 1), "n"(K0 + 2), "n"(K0 + 3));
}
// sum_k w[k] * h[k] (four partial sums: four independent dependency chains)
__device__ __forceinline__ float tf_dot(const float (&wv)[4], const float (&h)[TF_H]) {
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    tf_dot4<0>(a4, wv, h); tf_dot4<4>(a4, wv, h); tf_dot4<8>(a4, wv, h); tf_dot4<12>(a4, wv, h);
    return (a4[0] + a4[1]) + (a4[2] + a4[3]);
}

// layer 1, transposed: h[k] += sum_r W1T[c0 + r][k] * e[r]
__device__ __forceinline__ void tf_l1_acc(const float* __restrict__ w1t, int c0, const floatx4& e, float (&h)[TF_H]) {
    float wv[4][4];                 // the four rows are requested up front: one exposed round trip per 256 multiply-adds
#pragma unroll
    for (int r = 0; r < 4; ++r) tf_row(w1t + (size_t)(c0 + r) * TF_H, wv[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) tf_axpy(wv[r], e[r], h);
}
// layers 2 + 3 on relu(h): out[o] = sum_j W3[o][j] relu(sum_k W2[j][k] relu(h[k]))
template <int O>
__device__ __forceinline__ void tf_l23(const float* __restrict__ w2, const float* __restrict__ w3, float (&h)[TF_H], float (&out)[O]) {
#pragma unroll
    for (int k = 0; k < TF_H; ++k) h[k] = fmaxf(h[k], 0.f);
#pragma unroll
    for (int o = 0; o < O; ++o) out[o] = 0.f;
    float wn[4];
    tf_row(w2, wn);
#pragma unroll 2
    for (int j = 0; j < TF_H; ++j) {
        float wv[4] = {wn[0], wn[1], wn[2], wn[3]};
        tf_row(w2 + (size_t)(j + 1 < TF_H ? j + 1 : j) * TF_H, wn);        // the next row's weights fly under this row's arithmetic
        const float a = fmaxf(tf_dot(wv, h), 0.f);
#pragma unroll
        for (int o = 0; o < O; ++o) out[o] = fmaf(w3[o * TF_H + j], a, out[o]);
    }
}

struct tf_weights {            // device pointers; w1t = W1^T [96][64], w2 [64][64], w3 [O][64]
    const float* s1t; const float* s2; const float* s3;      // sdf head (O = 1)
    const float* f1t; const float* f2; const float* f3;      // feature head (O = 3)
};

// sdf of one point (one head): the features are streamed, nothing is kept
__device__ __forceinline__ float tf_sdf(const tf_geom& g, const asd_field_cfg& c, const float* __restrict__ planes, const tf_weights& w, float px, float py,
                                        float pz) {
    float nx, ny, nz;
    tf_norm(c, px, py, pz, nx, ny, nz);
    float h[TF_H];
#pragma unroll
    for (int k = 0; k < TF_H; ++k) h[k] = 0.f;
#pragma unroll 1
    for (int plane = 0; plane < 3; ++plane) {
        tf_tap t;
        tf_setup(g, plane, nx, ny, nz, t);
#pragma unroll 1
        for (int q = 0; q < 8; ++q) tf_l1_acc(w.s1t, plane * 32 + 4 * q, tf_quad(planes, t, q), h);
    }
    float o[1];
    tf_l23<1>(w.s2, w.s3, h, o);
    return o[0] + tf_bias(c, px, py, pz);
}

__global__ __launch_bounds__(256) void trifield_fwd_kernel(const tf_geom g, const asd_field_cfg c, const float* __restrict__ planes, const tf_weights w,
                                                           const float* __restrict__ points, int n, float* __restrict__ sdf, float* __restrict__ features,
                                                           float* __restrict__ normal, float* __restrict__ fd_grad) {
    // every lane of a wave runs the arithmetic (the DPP operands read OTHER lanes of the row: no lane may be switched off); a lane past the end
    // works on the last sample and stores nothing
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        const bool live = base + (int)threadIdx.x < n;
        const int i = live ? base + (int)threadIdx.x : n - 1;
        const float px = points[3 * i], py = points[3 * i + 1], pz = points[3 * i + 2];
        float s;
        if (features) {      // centre point: both heads share the lookup
            float nx, ny, nz;
            tf_norm(c, px, py, pz, nx, ny, nz);
            float hs[TF_H], hf[TF_H];
#pragma unroll
            for (int k = 0; k < TF_H; ++k) hs[k] = hf[k] = 0.f;
#pragma unroll 1
            for (int plane = 0; plane < 3; ++plane) {
                tf_tap t;
                tf_setup(g, plane, nx, ny, nz, t);
#pragma unroll 1
                for (int q = 0; q < 8; ++q) {
                    const floatx4 e = tf_quad(planes, t, q);
                    tf_l1_acc(w.s1t, plane * 32 + 4 * q, e, hs);
                    tf_l1_acc(w.f1t, plane * 32 + 4 * q, e, hf);
                }
            }
            float o1[1], o3[3];
            tf_l23<1>(w.s2, w.s3, hs, o1);
            tf_l23<3>(w.f2, w.f3, hf, o3);
            s = o1[0] + tf_bias(c, px, py, pz);
            if (live) { features[3 * (size_t)i] = o3[0]; features[3 * (size_t)i + 1] = o3[1]; features[3 * (size_t)i + 2] = o3[2]; }
        } else {
            s = tf_sdf(g, c, planes, w, px, py, pz);
        }
        if (live) sdf[i] = s;
        if (normal || fd_grad) {
            float nr[3];
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const float qx = asd_clampf(px + (k == 0 ? c.fd_eps : 0.f), -c.radius, c.radius);
                const float qy = asd_clampf(py + (k == 1 ? c.fd_eps : 0.f), -c.radius, c.radius);
                const float qz = asd_clampf(pz + (k == 2 ? c.fd_eps : 0.f), -c.radius, c.radius);
                nr[k] = (tf_sdf(g, c, planes, w, qx, qy, qz) - s) / c.fd_eps;
            }
            if (fd_grad && live) { fd_grad[3 * (size_t)i] = nr[0]; fd_grad[3 * (size_t)i + 1] = nr[1]; fd_grad[3 * (size_t)i + 2] = nr[2]; }
            if (normal && live) {
                const float inv = 1.f / fmaxf(sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), 1e-12f);
                normal[3 * (size_t)i] = nr[0] * inv; normal[3 * (size_t)i + 1] = nr[1] * inv; normal[3 * (size_t)i + 2] = nr[2] * inv;
            }
        }
    }
}

// ---- backward, per chunk of samples ----------------------------------------------------------------------------------------------------
// rows of the chunk workspace (row = 4 * (i - i0) + pt for the sdf head with a normal gradient, i - i0 without; the feature head has one row
// per sample):  ENC [96]  H1 [64]  DA1 [64]  DA2 [64]  per head, DENC [96] + PTS [3] per sdf row (the feature head adds into the centre's DENC)
struct tf_rows {
    float* enc; float* h1s; float* da1s; float* da2s;        // [rows_s][...]
    float* h1f; float* da1f; float* da2f;                    // [rows_f][...]  (ENC of the feature head = the centre rows of enc)
    float* denc; float* pts;                                 // [rows_s][96], [rows_s][3]
};

// one head, one point: layer 1 from the ENC row already written at `enc_row`, then the backward of layers 3, 2 and the ReLU of layer 1.
// Leaves H1 / DA1 / DA2 rows and the block-level sums of dW3 (LDS atomics); returns nothing: DA1 is re-read for the feature-gradient walk.
template <int O>
__device__ __forceinline__ void tf_head_bwd(const float* __restrict__ w1t, const float* __restrict__ w2, const float* __restrict__ w3, const float* enc_row,
                                            const float (&dout)[O], float* __restrict__ h1_row, float* __restrict__ da1_row, float* __restrict__ da2_row,
                                            float* __restrict__ w3_acc /* LDS [O][64] */, bool active) {
    float h[TF_H];
#pragma unroll
    for (int k = 0; k < TF_H; ++k) h[k] = 0.f;
#pragma unroll 1
    for (int q = 0; q < TF_NIN / 4; ++q) {
        floatx4 e = {0.f, 0.f, 0.f, 0.f};
        if (active) e = *(const floatx4*)(enc_row + 4 * q);
        tf_l1_acc(w1t, 4 * q, e, h);
    }
#pragma unroll
    for (int k = 0; k < TF_H; ++k) h[k] = fmaxf(h[k], 0.f);
    if (active) {
#pragma unroll
        for (int k = 0; k < TF_H; k += 4) *(floatx4*)(h1_row + k) = floatx4{h[k], h[k + 1], h[k + 2], h[k + 3]};
    }
    float dh[TF_H];
#pragma unroll
    for (int k = 0; k < TF_H; ++k) dh[k] = 0.f;
    float wn[4];
    tf_row(w2, wn);
#pragma unroll 1
    for (int j0 = 0; j0 < TF_H; j0 += 4) {
        float g4[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;
            float wv[4] = {wn[0], wn[1], wn[2], wn[3]};
            tf_row(w2 + (size_t)(j + 1 < TF_H ? j + 1 : j) * TF_H, wn);
            const float a = tf_dot(wv, h);
            float g = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) {
                g = fmaf(dout[o], w3[o * TF_H + j], g);
                const float v = active ? dout[o] * fmaxf(a, 0.f) : 0.f;
                if (v != 0.f) atomicAdd(&w3_acc[o * TF_H + j], v);
            }
            g = (active && a > 0.f) ? g : 0.f;
            g4[jj] = g;
            tf_axpy(wv, g, dh);
        }
        if (active) *(floatx4*)(da2_row + j0) = floatx4{g4[0], g4[1], g4[2], g4[3]};
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < TF_H; k += 4)
            *(floatx4*)(da1_row + k) = floatx4{h[k] > 0.f ? dh[k] : 0.f, h[k + 1] > 0.f ? dh[k + 1] : 0.f, h[k + 2] > 0.f ? dh[k + 2] : 0.f, h[k + 3] > 0.f ? dh[k + 3] : 0.f};
    }
}

__global__ __launch_bounds__(256) void trifield_bwd_kernel(const tf_geom g, const asd_field_cfg c, const float* __restrict__ planes, const tf_weights w,
                                                           const float* __restrict__ points, const float* __restrict__ sdf, int i0, int n_chunk,
                                                           const float* __restrict__ d_sdf, const float* __restrict__ d_features, const float* __restrict__ d_normal,
                                                           const float* __restrict__ d_fd_grad, const tf_rows R, float* __restrict__ dw3s, float* __restrict__ dw3f) {
    // block sums of the two W3 gradients [sdf | feature x 3][64]: every lane adds its own term with an LDS atomic into copy (lane & 15)
    // (row stride 257: the 16 copies of one entry sit in 16 banks, the four lanes of a copy serialise), summed at the end
    constexpr int W3S = 4 * TF_H + 1;
    __shared__ float w3_all[16 * W3S];
    for (int q = threadIdx.x; q < 16 * W3S; q += 256) w3_all[q] = 0.f;
    __syncthreads();
    float* const w3_acc = w3_all + (threadIdx.x & 15) * W3S;
    const int li = blockIdx.x * 256 + threadIdx.x;       // sample inside the chunk
    const bool active = li < n_chunk;
    const int i = i0 + (active ? li : 0);
    const bool with_fd = d_normal || d_fd_grad;
    const int npt = with_fd ? 4 : 1;
    const float px = points[3 * (size_t)i], py = points[3 * (size_t)i + 1], pz = points[3 * (size_t)i + 2];
    const float s = sdf[i];
    float ds = (active && d_sdf) ? d_sdf[i] : 0.f;
    float dsk[3] = {0.f, 0.f, 0.f};
    // ---- pass A: re-gather the features of every point of the stencil into its ENC row; the probes' sdf for the normal's gradient
    float sk[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int pt = 0; pt < npt; ++pt) {
        float qx = px, qy = py, qz = pz;
        if (pt > 0) {
            qx = asd_clampf(px + (pt == 1 ? c.fd_eps : 0.f), -c.radius, c.radius);
            qy = asd_clampf(py + (pt == 2 ? c.fd_eps : 0.f), -c.radius, c.radius);
            qz = asd_clampf(pz + (pt == 3 ? c.fd_eps : 0.f), -c.radius, c.radius);
        }
        float nx, ny, nz;
        tf_norm(c, qx, qy, qz, nx, ny, nz);
        const size_t row = (size_t)npt * li + pt;
        float h[TF_H];
#pragma unroll
        for (int k = 0; k < TF_H; ++k) h[k] = 0.f;
#pragma unroll 1
        for (int plane = 0; plane < 3; ++plane) {
            tf_tap t;
            tf_setup(g, plane, nx, ny, nz, t);
#pragma unroll 1
            for (int q = 0; q < 8; ++q) {
                const floatx4 e = tf_quad(planes, t, q);
                if (active) *(floatx4*)(R.enc + row * TF_NIN + plane * 32 + 4 * q) = e;
                if (pt > 0 && d_normal) tf_l1_acc(w.s1t, plane * 32 + 4 * q, e, h);        // (the normalisation needs the probes' values)
            }
        }
        if (active) { R.pts[3 * row] = nx; R.pts[3 * row + 1] = ny; R.pts[3 * row + 2] = nz; }
        if (pt > 0 && d_normal) {
            float o[1];
            tf_l23<1>(w.s2, w.s3, h, o);
            sk[pt - 1] = o[0] + tf_bias(c, qx, qy, qz);
        }
    }
    if (with_fd && active) {
        float dnr[3] = {0.f, 0.f, 0.f};
        if (d_normal) {
            float nr[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) nr[k] = (sk[k] - s) / c.fd_eps;
            const float len = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
            const float g0 = d_normal[3 * (size_t)i], g1 = d_normal[3 * (size_t)i + 1], g2 = d_normal[3 * (size_t)i + 2];
            if (len > 1e-12f) {
                const float inv = 1.f / len;
                const float n0 = nr[0] * inv, n1 = nr[1] * inv, n2 = nr[2] * inv;
                const float dot = n0 * g0 + n1 * g1 + n2 * g2;
                dnr[0] = (g0 - n0 * dot) * inv; dnr[1] = (g1 - n1 * dot) * inv; dnr[2] = (g2 - n2 * dot) * inv;
            } else {
                dnr[0] = g0 * 1e12f; dnr[1] = g1 * 1e12f; dnr[2] = g2 * 1e12f;
            }
        }
        if (d_fd_grad) { dnr[0] += d_fd_grad[3 * (size_t)i]; dnr[1] += d_fd_grad[3 * (size_t)i + 1]; dnr[2] += d_fd_grad[3 * (size_t)i + 2]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { dsk[k] = dnr[k] / c.fd_eps; ds -= dnr[k] / c.fd_eps; }
    }
    // ---- pass B: the heads' backward, point by point (rows written above are read back by the same thread)
#pragma unroll 1
    for (int pt = 0; pt < npt; ++pt) {
        const size_t row = (size_t)npt * li + pt;
        const float dout[1] = {pt == 0 ? ds : dsk[pt - 1]};
        tf_head_bwd<1>(w.s1t, w.s2, w.s3, R.enc + row * TF_NIN, dout, R.h1s + row * TF_H, R.da1s + row * TF_H, R.da2s + row * TF_H, w3_acc, active);
    }
    const bool feat = d_features != nullptr;
    if (feat) {
        float df[3] = {0.f, 0.f, 0.f};
        if (active) { df[0] = d_features[3 * (size_t)i]; df[1] = d_features[3 * (size_t)i + 1]; df[2] = d_features[3 * (size_t)i + 2]; }
        tf_head_bwd<3>(w.f1t, w.f2, w.f3, R.enc + (size_t)npt * li * TF_NIN, df, R.h1f + (size_t)li * TF_H, R.da1f + (size_t)li * TF_H, R.da2f + (size_t)li * TF_H,
                       w3_acc + TF_H, active);
    }
    // ---- pass C: feature gradient of every point: denc[c] = sum_k DA1[k] W1[k][c]  (= W1T[c][k]), the centre point through both heads
#pragma unroll 1
    for (int pt = 0; pt < npt; ++pt) {
        const size_t row = (size_t)npt * li + pt;
        float da[TF_H], db[TF_H];
#pragma unroll
        for (int k = 0; k < TF_H; ++k) { da[k] = 0.f; db[k] = 0.f; }
        if (active) {
#pragma unroll
            for (int k = 0; k < TF_H; k += 4) {
                const floatx4 v = *(const floatx4*)(R.da1s + row * TF_H + k);
                da[k] = v[0]; da[k + 1] = v[1]; da[k + 2] = v[2]; da[k + 3] = v[3];
            }
            if (feat && pt == 0) {
#pragma unroll
                for (int k = 0; k < TF_H; k += 4) {
                    const floatx4 v = *(const floatx4*)(R.da1f + (size_t)li * TF_H + k);
                    db[k] = v[0]; db[k + 1] = v[1]; db[k + 2] = v[2]; db[k + 3] = v[3];
                }
            }
        }
        const bool two = feat && pt == 0;
#pragma unroll 1
        for (int q = 0; q < TF_NIN / 4; ++q) {
            floatx4 d4 = {0.f, 0.f, 0.f, 0.f};
            float ws_[4][4], wf_[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tf_row(w.s1t + (size_t)(4 * q + r) * TF_H, ws_[r]);
                if (two) tf_row(w.f1t + (size_t)(4 * q + r) * TF_H, wf_[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = tf_dot(ws_[r], da);
                if (two) a += tf_dot(wf_[r], db);
                d4[r] = a;
            }
            if (active) *(floatx4*)(R.denc + row * TF_NIN + 4 * q) = d4;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 4 * TF_H; q += 256) {
        float a = 0.f;
#pragma unroll
        for (int cp = 0; cp < 16; ++cp) a += w3_all[cp * W3S + q];
        if (q < TF_H) atomicAdd(&dw3s[q], a);
        else if (feat) atomicAdd(&dw3f[q - TF_H], a);
    }
}

// out[64][NB] += sum_r A[r][0:64] (x) B[r][0:NB]   (rows r = 0 .. rows - 1; B rows `ldb` floats apart): the four weight-gradient products.
// 256 threads = 16 x 16 tiles of 4 x (NB / 16) outputs, 64-row LDS stages, a block walks every gridDim.x-th stage and adds its total once.
template <int NB>
__global__ __launch_bounds__(256) void tf_outer_kernel(const float* __restrict__ A, const float* __restrict__ B, size_t ldb, size_t rows, float* __restrict__ out) {
    constexpr int JB = NB / 16;
    __shared__ __attribute__((aligned(16))) float a_s[64 * TF_H];
    __shared__ __attribute__((aligned(16))) float b_s[64 * NB];
    const int tid = threadIdx.x, h0 = (tid >> 4) * 4, k0 = (tid & 15) * JB;
    float acc[4][JB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = 0.f;
    for (size_t r0 = (size_t)blockIdx.x * 64; r0 < rows; r0 += (size_t)gridDim.x * 64) {
        __syncthreads();
        for (int q = tid; q < 64 * TF_H / 4; q += 256) {
            const size_t r = r0 + q / (TF_H / 4);
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < rows) v = *(const floatx4*)(A + r * TF_H + (q % (TF_H / 4)) * 4);
            *(floatx4*)(a_s + 4 * q) = v;
        }
        for (int q = tid; q < 64 * NB / 4; q += 256) {
            const size_t r = r0 + q / (NB / 4);
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < rows) v = *(const floatx4*)(B + r * ldb + (q % (NB / 4)) * 4);
            *(floatx4*)(b_s + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < 64; ++t) {
            const floatx4 a = *(const floatx4*)(a_s + t * TF_H + h0);
            float bv[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) bv[j] = b_s[t * NB + k0 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[i][j] = fmaf(a[i], bv[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) atomicAdd(out + (size_t)(h0 + i) * NB + k0 + j, acc[i][j]);
}

// tri-plane scatter through an LDS image: block (slice, plane, 8-channel group) accumulates its rows' contributions to [H * W][8] floats in LDS
// (H = W = 64: 128 KB) and adds the image to d_planes once
__global__ __launch_bounds__(512) void tf_scatter_kernel(const float* __restrict__ denc, const float* __restrict__ pts, size_t rows, int H, int W, int slices,
                                                         float* __restrict__ d_planes) {
    extern __shared__ float img[];          // [H * W][8]
    const int slice = blockIdx.x, plane = blockIdx.y, grp = blockIdx.z;
    const int cells = H * W;
    for (int q = threadIdx.x; q < cells * 8; q += 512) img[q] = 0.f;
    __syncthreads();
    const size_t per = (rows + slices - 1) / slices, r0 = (size_t)slice * per, r1 = min(rows, r0 + per);
    for (size_t r = r0 + threadIdx.x; r < r1; r += 512) {
        float u, v, fx, fy;
        int x0, y0;
        tf_plane_uv(pts[3 * r], pts[3 * r + 1], pts[3 * r + 2], plane, u, v);
        tf_axis(u, W, x0, fx); tf_axis(v, H, y0, fy);
        const floatx4 g0 = *(const floatx4*)(denc + r * TF_NIN + plane * 32 + grp * 8), g1 = *(const floatx4*)(denc + r * TF_NIN + plane * 32 + grp * 8 + 4);
#pragma unroll
        for (int corner = 0; corner < 4; ++corner) {
            const int dx = corner & 1, dy = corner >> 1;
            const int x = x0 + dx, y = y0 + dy;
            if (x < 0 || x >= W || y < 0 || y >= H) continue;
            const float wt = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
            float* dst = img + (y * W + x) * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) { atomicAdd(dst + k, wt * g0[k]); atomicAdd(dst + 4 + k, wt * g1[k]); }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < cells * 8; q += 512) {
        const float v = img[q];
        if (v != 0.f) atomicAdd(d_planes + ((size_t)plane * cells + (q >> 3)) * 32 + grp * 8 + (q & 7), v);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// ASD_TRI_MFMA=0 selects the one-thread-per-sample kernels of this file (the first form of the field, kept as the A/B partner); the default is
// the matrix-pipe chain of trifield_mfma.hip
static bool tf_use_mfma() {
    static const bool on = !(getenv("ASD_TRI_MFMA") && getenv("ASD_TRI_MFMA")[0] == '0');
    return on;
}

// samples per backward chunk: the vector-pipe pass keeps 1.5 KB of rows per sample point (x 4 with a normal gradient: 1.5 GB per 256 k samples);
// the matrix-pipe pass keeps the 400-byte feature-gradient row only and takes 1 M samples at a time (1.7 GB), so that its weight-gradient kernels
// amortise their prologue (56-80 KB of weight images per block) and epilogue (11 k atomics per wave) over 16-64 tiles per wave
#define TF_CHUNK_VALU (256 * 1024)
#define TF_CHUNK_MFMA (1024 * 1024)
static int64_t tf_chunk() {                    // ASD_TRI_CHUNK (samples, read per call): the tests walk several chunks at small sizes
    const char* e = getenv("ASD_TRI_CHUNK");
    if (e && atoll(e) >= 64) return atoll(e);
    return tf_use_mfma() ? TF_CHUNK_MFMA : TF_CHUNK_VALU;
}

static int tf_check(const asd_field_cfg* c, int H, int W, int C) {
    ASD_CHECK_ARG(c && H > 0 && W > 0, "bad argument");
    if (C != 32 || c->n_hidden != 64 || c->n_feature_dims != 3 || c->field_mode != ASD_FIELD_SDF || !(c->bias_mode == ASD_BIAS_SPHERE || c->bias_mode == ASD_BIAS_CONST)) {
        asd_set_error("tri-plane field kernels: 3 x 32 channels, two hidden layers of 64, 3 feature dims, sdf mode, sphere / constant bias");
        return ASD_ERR_UNSUPPORTED;
    }
    return ASD_OK;
}
static tf_weights tf_w(const float* const* w6) { return tf_weights{w6[0], w6[1], w6[2], w6[3], w6[4], w6[5]}; }

extern "C" {

int asd_trifield_fwd_workspace(int32_t H, int32_t W, int64_t* n_floats) {
    ASD_CHECK_ARG(n_floats && H > 0 && W > 0, "bad argument");
    *n_floats = tfm_prep_floats(H, W);
    return ASD_OK;
}

int asd_trifield_fwd(const float* planes_cl, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* const* weights /* [host] 6 device pointers:
                     sdf W1^T [96][64], W2 [64][64], W3 [1][64], feature W1^T, W2, W3 [3][64] */, const float* points, int32_t n, float* sdf, float* features,
                     float* normal, float* fd_grad, float* workspace, void* stream) {
    if (n == 0) return ASD_OK;
    ASD_CHECK_ARG(planes_cl && weights && points && sdf && workspace && n > 0, "null argument");
    const int rc = tf_check(cfg, H, W, C);
    if (rc != ASD_OK) return rc;
    if (tf_use_mfma()) {
        const int rp = tfm_prepare(planes_cl, H, W, weights, workspace, (hipStream_t)stream);
        if (rp != ASD_OK) return rp;
        tfm_forward(tf_geom{H, W}, cfg, planes_cl, weights, workspace, points, n, sdf, features, normal, fd_grad, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(trifield_fwd_kernel, dim3(asd_grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, tf_geom{H, W}, *cfg, planes_cl, tf_w(weights), points,
                           n, sdf, features, normal, fd_grad);
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

// ASD_TRI_SORT=0: the run-aggregated atomic scatter (asd_triplane_sample_bwd_rows) instead of the sorted one
static bool tf_use_sort(int H, int W) {
    static const bool on = !(getenv("ASD_TRI_SORT") && getenv("ASD_TRI_SORT")[0] == '0');
    return on && tfs_supported(H, W);
}

int asd_trifield_bwd_workspace(int32_t H, int32_t W, int32_t n, int32_t with_normal, int64_t* n_floats) {
    ASD_CHECK_ARG(n_floats && n >= 0 && H > 0 && W > 0, "bad argument");
    const int64_t ch = n < tf_chunk() ? n : tf_chunk(), rs = ch * (with_normal ? 4 : 1);
    // matrix-pipe pass: feature-gradient rows + points for the scatter, scales / weight images, the sort's bins and row list; the vector-pipe
    // pass adds the rows of its four weight-gradient products
    *n_floats = rs * (TF_NIN + 4) + 1024 + tfm_prep_floats(H, W) + tfs_work_ints((int)rs, H, W)
                + (tf_use_mfma() ? 0 : rs * (TF_NIN + 3 * TF_H) + ch * 3 * TF_H);
    return ASD_OK;
}

int asd_trifield_bwd(const float* planes_cl, int32_t H, int32_t W, int32_t C, const asd_field_cfg* cfg, const float* const* weights, const float* points,
                     const float* sdf, int32_t n, const float* d_sdf, const float* d_features, const float* d_normal, const float* d_fd_grad, float* d_planes_cl,
                     float* const* d_weights /* [host] 6 device pointers, same shapes (W1 gradients as W1^T), accumulated (+=) */, float* workspace, void* stream) {
    if (n == 0) return ASD_OK;
    ASD_CHECK_ARG(planes_cl && weights && d_weights && points && sdf && d_planes_cl && workspace && n > 0, "null argument");
    const int rc = tf_check(cfg, H, W, C);
    if (rc != ASD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int with_fd = d_normal != nullptr || d_fd_grad != nullptr, npt = with_fd ? 4 : 1;
    const int64_t ch = n < tf_chunk() ? n : tf_chunk(), rs_max = ch * npt;
    tf_rows R;
    float* p = workspace;
    R.denc = p; p += rs_max * TF_NIN;
    R.pts = p; p += rs_max * 4;
    p += 1024;
    float* const prep = p; p += tfm_prep_floats(H, W);
    int* const sort_work = (int*)p; p += tfs_work_ints((int)rs_max, H, W);
    const bool mfma = tf_use_mfma(), sorted = tf_use_sort(H, W);
    R.enc = p; p += rs_max * TF_NIN;           // (vector-pipe pass only: not part of the workspace otherwise, and never touched)
    R.h1s = p; p += rs_max * TF_H;
    R.da1s = p; p += rs_max * TF_H;
    R.da2s = p; p += rs_max * TF_H;
    R.h1f = p; p += ch * TF_H;
    R.da1f = p; p += ch * TF_H;
    R.da2f = p; p += ch * TF_H;
    const tf_weights w = tf_w(weights);
    if (mfma) {
        const int rp = tfm_prepare(planes_cl, H, W, weights, prep, s);
        if (rp != ASD_OK) return rp;
    }
    // the LDS-image scatter (tf_scatter_kernel) measured 2.0 ms per 1 M rows against 0.68 ms for the run-aggregated global atomics of
    // asd_triplane_sample_bwd (LDS float atomics retire ~0.4 lane-ops per clock and CU): kept for reference, off
    static const bool want_lds = getenv("ASD_TRI_LDS_SCATTER") && getenv("ASD_TRI_LDS_SCATTER")[0] == '1';
    const bool lds_scatter = want_lds && (size_t)H * W * 8 * 4 <= 150 * 1024;
    if (lds_scatter) {
        static std::atomic<unsigned long long> attr_devmask{0}; bool attr = !asd_attr_needed(attr_devmask);
        if (!attr) { (void)hipFuncSetAttribute((const void*)tf_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    }
    for (int64_t i0 = 0; i0 < n; i0 += ch) {
        const int nc = (int)((n - i0) < ch ? (n - i0) : ch);
        const size_t rows_s = (size_t)nc * npt;
        if (mfma) {
            tfm_backward_chunk(tf_geom{H, W}, cfg, planes_cl, weights, prep, points, sdf, (int)i0, nc, npt, d_sdf, d_features, d_normal, d_fd_grad, R.denc, R.pts,
                               d_weights, s);
        } else {
            hipLaunchKernelGGL(trifield_bwd_kernel, dim3(asd_div_up(nc, 256)), dim3(256), 0, s, tf_geom{H, W}, *cfg, planes_cl, w, points, sdf, (int)i0, nc, d_sdf,
                               d_features, d_normal, d_fd_grad, R, d_weights[2], d_weights[5]);
        }
        // feature gradient -> planes
        if (lds_scatter) {
            const int slices = 20;
            hipLaunchKernelGGL(tf_scatter_kernel, dim3(slices, 3, 4), dim3(512), (size_t)H * W * 8 * 4, s, R.denc, R.pts, rows_s, H, W, slices, d_planes_cl);
        } else if (sorted) {
            tfs_scatter(R.denc, R.pts, (int)rows_s, H, W, d_planes_cl, sort_work, s);
        } else {
            static const int run = getenv("ASD_TRI_RUN") ? atoi(getenv("ASD_TRI_RUN")) : 128;     // rows in ray order: C5 step 100.9 ms at 8, 93.6 at 128
            const int rc2 = asd_triplane_sample_bwd_rows(R.denc, H, W, 32, R.pts, (int32_t)rows_s, d_planes_cl, run, stream);
            if (rc2 != ASD_OK) return rc2;
        }
        // weight gradients: dW1^T [96][64] = ENC^T DA1 is accumulated as out[64][96]^T — the kernel's `out` is [64][NB] = DA1^T ENC; the W1 gradient is
        // handed back in that ([64][96]) layout for the first layer, see the header
        if (mfma) continue;                     // (the matrix-pipe pass has accumulated the weight gradients itself)
        const int gb = 1024;
        hipLaunchKernelGGL((tf_outer_kernel<TF_NIN>), dim3(gb), dim3(256), 0, s, R.da1s, R.enc, (size_t)TF_NIN, rows_s, d_weights[0]);
        hipLaunchKernelGGL((tf_outer_kernel<TF_H>), dim3(gb), dim3(256), 0, s, R.da2s, R.h1s, (size_t)TF_H, rows_s, d_weights[1]);
        if (d_features) {
            hipLaunchKernelGGL((tf_outer_kernel<TF_NIN>), dim3(gb), dim3(256), 0, s, R.da1f, R.enc, (size_t)TF_NIN * npt, (size_t)nc, d_weights[3]);
            hipLaunchKernelGGL((tf_outer_kernel<TF_H>), dim3(gb), dim3(256), 0, s, R.da2f, R.h1f, (size_t)TF_H, (size_t)nc, d_weights[4]);
        }
    }
    ASD_LAUNCH_CHECK();
    return ASD_OK;
}

}  // extern "C"
