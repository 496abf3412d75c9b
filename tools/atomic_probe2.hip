// Probe 2: does gfx950 coalesce fp32 atomics of one wave instruction that fall into the same 16/32/64/128-B block?
// (tools/atomic_probe.hip found 21 G atomics/s for one random address per lane, independent of scope and table size.)
// The hash-grid gradient scatter adds TWO adjacent floats per table entry (and on dense levels / even cx the x-neighbour
// entry is adjacent too), so if requests — not dwords — are what is rate limited, laying the lanes of one instruction out as
// [entry k: f0 | f1] halves the cost.   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe2.hip -o /tmp/atomic_probe2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// GROUP = number of adjacent dwords that consecutive lanes of one instruction update (1 = one random dword per lane)
template <int GROUP>
__global__ void k_group(float* tab, uint32_t entry_mask, int per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = t / GROUP, sub = t % GROUP;
    uint32_t x = g * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t e = (x >> 8) & entry_mask;      // random block of GROUP dwords
        atomicAdd(tab + (size_t)e * GROUP + sub, 1.0f);
    }
}

// today's scatter: one random 8-byte entry per lane, two instructions (f0 then f1)
__global__ void k_two_instr(float* tab, uint32_t entry_mask, int per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t e = (x >> 8) & entry_mask;
        atomicAdd(tab + 2 * (size_t)e, 1.0f);
        atomicAdd(tab + 2 * (size_t)e + 1, 1.0f);
    }
}

// the same work transposed through the wave: instruction j covers the entries of source lanes 32j..32j+31, lane pair (2k, 2k+1)
// updates (f0, f1) of one entry
__global__ void k_paired(float* tab, uint32_t entry_mask, int per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t x = t * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t e = (x >> 8) & entry_mask;
        const float v0 = 1.0f, v1 = 1.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int src = 32 * j + (lane >> 1);
            const uint32_t es = __shfl(e, src, 64);
            const float a = __shfl(v0, src, 64), b = __shfl(v1, src, 64);
            atomicAdd(tab + 2 * (size_t)es + (lane & 1), (lane & 1) ? b : a);
        }
    }
}

__global__ void k_f64(double* tab, uint32_t entry_mask, int per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        atomicAdd(tab + ((x >> 8) & entry_mask), 1.0);
    }
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    const size_t n = 1u << 24;  // 64 MB of floats
    float* tab;
    hipMalloc(&tab, n * 4);
    hipMemset(tab, 0, n * 4);
    const int blocks = 2048, threads = 256, per = 64;
    const double lanes = (double)blocks * threads * per;
#define RUN_GROUP(G)                                                                                                   \
    {                                                                                                                  \
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_group<G>, dim3(blocks), dim3(threads), 0, 0, tab,          \
                                                          (uint32_t)(n / G - 1), per); });                             \
        printf("group of %2d adjacent dwords per request : %7.3f ms  %7.2f G dword-atomics/s  %7.2f G blocks/s\n", G,  \
               ms, lanes / ms / 1e6, lanes / G / ms / 1e6);                                                            \
    }
    RUN_GROUP(1) RUN_GROUP(2) RUN_GROUP(4) RUN_GROUP(8) RUN_GROUP(16) RUN_GROUP(32)
    float ms = time_ms([&] { hipLaunchKernelGGL(k_two_instr, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n / 2 - 1), per); });
    printf("entry per lane, two instructions        : %7.3f ms  %7.2f G dword-atomics/s\n", ms, 2 * lanes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_paired, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n / 2 - 1), per); });
    printf("same work, lane pairs share an entry     : %7.3f ms  %7.2f G dword-atomics/s\n", ms, 2 * lanes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_f64, dim3(blocks), dim3(threads), 0, 0, (double*)tab, (uint32_t)(n / 2 - 1), per); });
    printf("f64 atomic per lane                      : %7.3f ms  %7.2f G atomics/s\n", ms, lanes / ms / 1e6);
    // correctness of the paired form
    hipMemset(tab, 0, n * 4);
    hipLaunchKernelGGL(k_paired, dim3(64), dim3(256), 0, 0, tab, 1023u, 16);
    hipDeviceSynchronize();
    static float h[2048];
    hipMemcpy(h, tab, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (float v : h) s += v;
    printf("paired sum = %.0f (expected %.0f)\n", s, 2.0 * 64 * 256 * 16);
    return 0;
}
