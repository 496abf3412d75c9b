// Probe: throughput of fp32 global atomics on gfx950 by scope and address pattern (decides how the hash-grid
// gradient scatter is organised).  build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE>
__global__ void k_atomic(float* tab, uint32_t mask, int per_thread, int pattern) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        uint32_t idx;
        if (pattern == 0) { x = x * 1664525u + 1013904223u; idx = (x >> 8) & mask; }        // uniform random
        else if (pattern == 1) { idx = ((t >> 4) * 8 + (i & 7)) & mask; }                       // runs of 16 lanes share an address
        else { x = x * 1664525u + 1013904223u; idx = (x >> 8) & 1023u; }                        // 1024 hot addresses
        if (SCOPE == 0) atomicAdd(tab + idx, 1.0f);
        else if (SCOPE == 1) __hip_atomic_fetch_add(tab + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(tab + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_pairs(float* tab, uint32_t mask, int per_thread) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t * 2654435761u + 12345u;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 8) & mask;
        atomicAdd(tab + 2 * idx, 1.0f);
        atomicAdd(tab + 2 * idx + 1, 1.0f);
    }
}

int main() {
    const size_t n = 1u << 24;  // 64 MB table
    float* tab;
    hipMalloc(&tab, n * 4);
    hipMemset(tab, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, threads = 256, per = 64;
    const double total = (double)blocks * threads * per;
    const char* pn[3] = {"random 64MB", "runs of 16 lanes", "1024 hot addrs"};
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int scope = 0; scope < 3; ++scope) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (scope == 0) hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n - 1), per, pattern);
                if (scope == 1) hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n - 1), per, pattern);
                if (scope == 2) hipLaunchKernelGGL(k_atomic<2>, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n - 1), per, pattern);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%-18s scope=%s : %8.3f ms  %7.2f G atomics/s\n", pn[pattern], scope == 0 ? "atomicAdd " : scope == 1 ? "workgroup " : "agent     ", ms,
                   total / ms / 1e6);
        }
    // rate vs table size (does a level-sized table that fits the L2s scatter faster?), pairs of adjacent floats as in the hash grid
    for (int lg = 18; lg <= 24; ++lg) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_pairs, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)((1u << (lg - 1)) - 1), per);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("random pairs, table %6.2f MB : %8.3f ms  %7.2f G atomics/s\n", (double)(1u << lg) * 4 / 1048576.0, ms, 2 * total / ms / 1e6);
    }
    // correctness of workgroup-scope atomics across CUs/XCDs on a hot address set
    hipMemset(tab, 0, n * 4);
    hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n - 1), per, 2);
    hipDeviceSynchronize();
    std::vector<float> h(1024);
    hipMemcpy(h.data(), tab, 4096, hipMemcpyDeviceToHost);
    double s = 0; for (float v : h) s += v;
    printf("workgroup-scope hot-set sum = %.0f (expected %.0f)\n", s, total);
    return 0;
}
