// Probe: how fast can a CU pull L2-resident (or HBM-resident) data — LDS-DMA (global_load_lds_dwordx4) vs plain
// global_load_dwordx4 into registers — as a function of the number of active CUs.  Decides whether the ~8 TB/s aggregate
// tile-load rate of the GEMM kernels is a property of the LDS-DMA path, of the L2, or of the fabric.
// build: hipcc --offload-arch=gfx950 -O3 tools/load_probe.hip -o /tmp/load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// every wave streams `iters` x 1 KiB; the block's waves walk a private window of `span` bytes (L2-resident when small)
template <int MODE>
__global__ __launch_bounds__(512) void k_load(const char* src, size_t span_per_block, int iters, float* sink, int share = 1, int same_xcd = 1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // `share` blocks stream the SAME window concurrently: on one XCD (block ids congruent mod 8) or spread over XCDs
    const int b = blockIdx.x;
    const int win = share == 1 ? b : (same_xcd ? ((b >> 3) / share) * 8 + (b & 7) : b / share);
    const char* base = src + (size_t)win * span_per_block;
    const unsigned span = (unsigned)span_per_block;
    unsigned off = (unsigned)(wave * 1024 + lane * 16);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + off), (LDS_AS void*)(smem + wave * 8192 + u * 1024), 16, 0, 0);
            } else {
                const float4 v = *(const float4*)(base + off);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            off += 8192;
            if (off >= span) off -= span;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) { __syncthreads(); acc.x = ((float*)smem)[threadIdx.x]; }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf; float* sink;
    hipMalloc(&buf, total); hipMalloc(&sink, 4);
    hipMemset(buf, 1, total);
    hipFuncSetAttribute((const void*)k_load<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048;   // 2 MiB per wave
    for (int mode = 0; mode < 2; ++mode)
        for (size_t span : {(size_t)65536, (size_t)1 << 20, (size_t)4 << 20}) {   // per-block window: 64 KiB (L2 hit), 1 MiB, 4 MiB (misses once blocks * span > L2)
            for (int blocks : {32, 64, 128, 256, 512}) {
                if ((size_t)blocks * span > total) continue;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(512), 65536, 0, buf, span, iters, sink);
                    else hipLaunchKernelGGL(k_load<1>, dim3(blocks), dim3(512), 0, 0, buf, span, iters, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                }
                const double bytes = (double)blocks * 8 * iters * 1024;
                printf("%s window %5zu KiB/block  %3d blocks: %7.3f ms  %6.2f TB/s  %6.1f GB/s per block\n", mode == 0 ? "lds-dma " : "vgpr    ",
                       span >> 10, blocks, ms, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
            }
        }
    for (int same : {1, 0})
        for (int share : {1, 5}) {
            float ms = 0;
            const int blocks = 480;   // one resident round (2 blocks per CU fit: 64 KB LDS each)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_load<0>, dim3(blocks), dim3(512), 65536, 0, buf, (size_t)1 << 20, iters, sink, share, same);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)blocks * 8 * iters * 1024;
            printf("lds-dma 1 MiB windows, %d blocks, %d blocks per window (%s): %7.3f ms  %6.2f TB/s\n", blocks, share, same ? "same XCD" : "consecutive ids", ms,
                   bytes / ms / 1e9);
        }
    return 0;
}
