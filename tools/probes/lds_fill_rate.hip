// How fast can one CU pull L2-resident data into its LDS on gfx950 — through LDS-DMA (global_load_lds_dwordx4), through registers
// (global_load_dwordx4 + ds_write_b128), or both at once?  One workgroup of 8 waves per CU (256 workgroups), every wave streams
// 16 B per lane per copy out of a 2 MiB window (L2 hits after the first pass), DEPTH copies in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 lds_fill_rate.hip -o lds_fill_rate && ./lds_fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// MODE 3 / 4: LDS-DMA / registers with the ROW-STRIDED pattern of a GEMM operand tile: a copy = 8 rows x 128 B, rows PITCH bytes apart
#define PITCH 7168
template <int MODE, int DEPTH>   // MODE 0: DMA   1: registers + ds_write   2: alternate copies between the two paths
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, int iters, int window, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* mine = lds + wave * (DEPTH * 1024);
    uint32_t off = (uint32_t)(((blockIdx.x * 8 + wave) * 4096 + lane * 16) % window);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint4 r[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* p = src + ((off + d * 1024) & (uint32_t)(window - 1));
            if (MODE >= 3) p = src + ((((off >> 10) + d * 8) * PITCH) & (uint32_t)(window - 1) & ~127u) + (lane >> 3) * PITCH + (lane & 7) * 16;
            const bool dma = MODE == 0 || MODE == 3 || (MODE == 2 && (d & 1) == 0);
            if (dma) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(mine + d * 1024), 16, 0, 0);
            else r[d] = *reinterpret_cast<const uint4*>(p);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const bool dma = MODE == 0 || MODE == 3 || (MODE == 2 && (d & 1) == 0);
            if (!dma) *reinterpret_cast<uint4*>(mine + d * 1024 + lane * 16) = r[d];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        off = (off + DEPTH * 1024 * 97) & (uint32_t)(window - 1);
        if (it == iters - 1) acc += *reinterpret_cast<const float*>(mine + lane * 4);
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* src, float* sink, int window, const char* name) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)fill<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, src, 200, window, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, src, iters, window, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 256.0 * 8 * DEPTH * 1024.0 * iters;
    printf("  %-28s depth %2d (%3d KiB in flight per CU): %7.2f TB/s total = %6.1f GB/s per CU\n", name, DEPTH, 8 * DEPTH, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const int window = 2 << 20;
    char* src; float* sink;
    hipMalloc(&src, window + 16 * PITCH); hipMemset(src, 1, window + 16 * PITCH); hipMalloc(&sink, 4);
    printf("L2-resident source (2 MiB window), 256 workgroups x 8 waves:\n");
    run<0, 4>(src, sink, window, "LDS-DMA");        run<0, 8>(src, sink, window, "LDS-DMA");        run<0, 16>(src, sink, window, "LDS-DMA");
    run<1, 4>(src, sink, window, "registers + ds_write_b128"); run<1, 8>(src, sink, window, "registers + ds_write_b128"); run<1, 16>(src, sink, window, "registers + ds_write_b128");
    run<2, 4>(src, sink, window, "alternating");    run<2, 8>(src, sink, window, "alternating");    run<2, 16>(src, sink, window, "alternating");
    run<3, 4>(src, sink, window, "LDS-DMA, row-strided");  run<3, 8>(src, sink, window, "LDS-DMA, row-strided");  run<3, 16>(src, sink, window, "LDS-DMA, row-strided");
    run<4, 8>(src, sink, window, "registers, row-strided");
    const int big = 1 << 30;
    char* src2; hipMalloc(&src2, (size_t)big + 16 * PITCH); hipMemset(src2, 1, (size_t)big + 16 * PITCH);
    printf("HBM source (1 GiB window):\n");
    run<0, 8>(src2, sink, big, "LDS-DMA");  run<0, 16>(src2, sink, big, "LDS-DMA");
    run<1, 8>(src2, sink, big, "registers + ds_write_b128");  run<1, 16>(src2, sink, big, "registers + ds_write_b128");
    run<2, 16>(src2, sink, big, "alternating");
    return 0;
}
