// Probe: does global_load_lds (16 B per lane, LDS-DMA) accept source addresses that are only 2-byte aligned, and what does it cost?
// The weight-gradient GEMM of the 3-D convolution (csrc/conv3d.hip) contracts over voxels; tap (kd, ky, kx) of the input is the SAME
// channel-major plane shifted by kd * Hp * Wp + ky * Wp + kx elements, i.e. kx = +-1 moves the 16-byte chunks of a row by 2 bytes.
//   build: hipcc --offload-arch=gfx950 -O3 tools/lds_dma_align_probe.hip -o /tmp/lds_dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// every wave copies `iters` KiB from src + shift (bytes) to LDS and writes them back out
__global__ __launch_bounds__(256) void k_copy(const char* src, char* dst, int shift, int iters, size_t stride) {
    __shared__ __attribute__((aligned(16))) char lds[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t g = (size_t)blockIdx.x * 4 + wave;
    const char* s = src + g * stride + shift + lane * 16;
    char* d = dst + g * stride + lane * 16;
    for (int i = 0; i < iters; ++i) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(s + (size_t)i * 1024), (LDS_AS void*)lds[wave], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 v = *(const uint4*)(lds[wave] + lane * 16);
        *(uint4*)(d + (size_t)i * 1024) = v;
    }
}

int main() {
    const int blocks = 2048, iters = 64;
    const size_t stride = (size_t)iters * 1024 + 64, total = (size_t)blocks * 4 * stride + 4096;
    std::vector<uint8_t> h(total);
    for (size_t i = 0; i < total; ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
    char *src, *dst;
    hipMalloc(&src, total); hipMalloc(&dst, total);
    hipMemcpy(src, h.data(), total, hipMemcpyHostToDevice);
    std::vector<uint8_t> out(total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shift : {0, 2, 4, 6, 8, 14, 16, 18, 62}) {
        hipMemset(dst, 0, total);
        k_copy<<<blocks, 256>>>(src, dst, shift, iters, stride);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_copy<<<blocks, 256>>>(src, dst, shift, iters, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(out.data(), dst, total, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t g = 0; g < (size_t)blocks * 4; ++g)
            for (size_t b = 0; b < (size_t)iters * 1024; ++b)
                if (out[g * stride + b] != h[g * stride + shift + b]) ++bad;
        printf("shift %2d B: %zu wrong bytes of %zu, %.1f us per launch, %.2f TB/s\n", shift, bad, (size_t)blocks * 4 * iters * 1024, ms / 5 * 1e3,
               (double)blocks * 4 * iters * 1024 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
