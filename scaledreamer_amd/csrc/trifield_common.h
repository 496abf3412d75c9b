// trifield_common.h — geometry helpers shared by the two evaluations of the tri-plane field (trifield.hip: one thread per sample on the vector
// pipe; trifield_mfma.hip: tiles of rows on the matrix pipe): contraction to grid_sample coordinates, plane projection, bilinear taps.
#pragma once
#include "asd_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define TF_H 64
#define TF_NIN 96

struct tf_geom { int H, W; };

__device__ __forceinline__ void tf_axis(float x, int size, int& i0, float& w1) {
    const float ix = ((x + 1.f) * (float)size - 1.f) * 0.5f;   // grid_sample, align_corners = False
    const float f = floorf(ix);
    i0 = (int)f;
    w1 = ix - f;
}
__device__ __forceinline__ void tf_plane_uv(float x, float y, float z, int plane, float& u, float& v) {    // (x,y), (x,z), (z,y); first -> W
    if (plane == 0) { u = x; v = y; } else if (plane == 1) { u = x; v = z; } else { u = z; v = y; }
}
__device__ __forceinline__ float tf_bias(const asd_field_cfg& c, float px, float py, float pz) {
    if (c.bias_mode == ASD_BIAS_SPHERE) return sqrtf(px * px + py * py + pz * pz) - c.bias_value;
    return c.bias_value;
}
// the bilinear setup of one plane for a point in grid_sample coordinates
struct tf_tap { int off[4]; float w[4]; };
__device__ __forceinline__ void tf_setup(const tf_geom& g, int plane, float nx, float ny, float nz, tf_tap& t) {
    float u, v, fx, fy;
    int x0, y0;
    tf_plane_uv(nx, ny, nz, plane, u, v);
    tf_axis(u, g.W, x0, fx); tf_axis(v, g.H, y0, fy);
#pragma unroll
    for (int corner = 0; corner < 4; ++corner) {
        const int dx = corner & 1, dy = corner >> 1;
        const int x = x0 + dx, y = y0 + dy;
        const bool ok = x >= 0 && x < g.W && y >= 0 && y < g.H;
        t.off[corner] = ok ? ((plane * g.H + y) * g.W + x) * 32 : -1;
        t.w[corner] = ok ? (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) : 0.f;
    }
}

// a / d, as a multiplication when d is a power of two (the usual box: +-radius) — both are exact, so the result is the same bit for bit; d is
// wave-uniform, the branch is scalar (an fp32 division is ten instructions, three of them per point)
__device__ __forceinline__ float tf_div(float a, float d) {
    const unsigned b = __float_as_uint(d), e = (b >> 23) & 0xffu;
    if ((b & 0x007fffffu) == 0u && e >= 2u && e <= 252u) return a * __uint_as_float((b & 0x80000000u) | ((254u - e) << 23));
    return a / d;
}
__device__ __forceinline__ void tf_norm(const asd_field_cfg& c, float px, float py, float pz, float& nx, float& ny, float& nz) {
    nx = 2.f * tf_div(px - c.bbox_min[0], c.bbox_max[0] - c.bbox_min[0]) - 1.f;
    ny = 2.f * tf_div(py - c.bbox_min[1], c.bbox_max[1] - c.bbox_min[1]) - 1.f;
    nz = 2.f * tf_div(pz - c.bbox_min[2], c.bbox_max[2] - c.bbox_min[2]) - 1.f;
}
