// Probe: how fast can 256 CUs stream a COLD 29.5 MB weight matrix [N = 1280][K = 11520] fp16 out of HBM in the access pattern of the GEMM
// kernels (a block reads 64 rows x 128 B per k-step: 64 separate 128-B pieces 23 KB apart) against a tile-major layout (the same 8 KB of a
// k-step contiguous), for different numbers of blocks and k-steps in flight?  No LDS, no MFMA: the loads are xor-reduced.
//   build: hipcc --offload-arch=gfx950 -O3 tools/wstream_probe.hip -o /tmp/wstream_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int N = 1280, K = 11520, KS = K / 64;   // 180 k-steps

// TILED = false: W[n][k] row-major; true: W[(n/64)][ks][64 rows][64 k]
template <bool TILED, int DEPTH>
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ W, int split, uint4* __restrict__ sink) {
    const int tn = blockIdx.x % (N / 64), kz = blockIdx.x / (N / 64);
    const int per = (KS + split - 1) / split, ks0 = kz * per, ks1 = min(KS, ks0 + per);
    const int tid = threadIdx.x, r = tid >> 3, c = tid & 7;
    uint4 acc = {0, 0, 0, 0};
    for (int ks = ks0; ks < ks1; ks += DEPTH) {
        uint4 v[DEPTH][2];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int k = min(ks + d, ks1 - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                size_t idx;   // in 16-byte units
                if (TILED) idx = ((size_t)(tn * KS + k) * 64 + (r + 32 * h)) * 8 + c;
                else idx = ((size_t)(tn * 64 + r + 32 * h) * K + (size_t)k * 64) / 8 + c;
                v[d][h] = W[idx];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int h = 0; h < 2; ++h) { acc.x ^= v[d][h].x; acc.y ^= v[d][h].y; acc.z ^= v[d][h].z; acc.w ^= v[d][h].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x * 256 + tid] = acc;
}

template <bool TILED, int DEPTH>
static float run(const std::vector<uint4*>& pool, int split, uint4* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 28;
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_stream<TILED, DEPTH>), dim3(N / 64 * split), dim3(256), 0, 0, pool[i % pool.size()], split, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<TILED, DEPTH>), dim3(N / 64 * split), dim3(256), 0, 0, pool[(i + 4) % pool.size()], split, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const size_t bytes = (size_t)N * K * 2;
    std::vector<uint4*> pool(14);
    for (auto& p : pool) { hipMalloc(&p, bytes); hipMemset(p, 1, bytes); }
    uint4* sink; hipMalloc(&sink, 1 << 24);
    printf("weights %.1f MB, pool %zu (%.0f MB)\n", bytes / 1e6, pool.size(), pool.size() * bytes / 1e6);
    for (int split : {5, 10, 20, 40, 90}) {
        const float a1 = run<false, 1>(pool, split, sink), a2 = run<false, 2>(pool, split, sink), a4 = run<false, 4>(pool, split, sink), a8 = run<false, 8>(pool, split, sink);
        const float b1 = run<true, 1>(pool, split, sink), b2 = run<true, 2>(pool, split, sink), b4 = run<true, 4>(pool, split, sink), b8 = run<true, 8>(pool, split, sink);
        printf("blocks %4d: row-major depth 1/2/4/8: %6.1f %6.1f %6.1f %6.1f us | tile-major: %6.1f %6.1f %6.1f %6.1f us   (%.2f TB/s best)\n", 20 * split, a1, a2, a4, a8,
               b1, b2, b4, b8, bytes / 1e6 / fminf(fminf(fminf(a1, a2), fminf(a4, a8)), fminf(fminf(b1, b2), fminf(b4, b8))));
    }
    return 0;
}
