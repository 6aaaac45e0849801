// Does the ROW-STRIDED access pattern of a weight-streaming GEMM tile (160 weight rows x 128 B per K-tile, rows 7168 B apart) cost
// HBM bandwidth against the same bytes laid out tile-contiguously?  256 workgroups x 8 waves; workgroup b streams its 160 rows of
// an [N][3584] bf16 matrix K-tile by K-tile through LDS-DMA (16 B per lane, 1 KiB per copy = 8 rows x 128 B), RING K-tiles in
// flight per wave; COLD data (a different matrix of a 2.3 GiB pool per launch).
//   hipcc --offload-arch=gfx950 -O3 weight_stream.hip -o weight_stream && ./weight_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ROWS 160
#define KB 3584
#define NKT (KB / 64)
#define COPIES (ROWS / 8)

template <bool CONTIG, int RING>
__global__ __launch_bounds__(512) void stream(const char* __restrict__ w, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + (int64_t)blockIdx.x * ROWS * KB * 2;
    // strided: copy c of K-tile kt = rows 8c..8c+7, bytes [128 kt, 128 kt + 128); lane -> row 8c + lane/8, 16-byte chunk lane%8
    const uint32_t lane_off = CONTIG ? lane * 16 : (uint32_t)((lane >> 3) * KB * 2 + (lane & 7) * 16);
    for (int kt = 0; kt < NKT; ++kt) {
        char* slot = lds + (kt % RING) * (COPIES * 1024);
        for (int c = wave; c < COPIES; c += 8) {
            const char* p = CONTIG ? base + ((int64_t)kt * COPIES + c) * 1024 + lane_off : base + (int64_t)c * 8 * KB * 2 + kt * 128 + lane_off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(slot + c * 1024), 16, 0, 0);
        }
        // keep RING - 1 K-tiles of this wave's copies in flight (each wave issues 2 or 3 copies per K-tile: wait on the older tile)
        if (RING == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (RING == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (*reinterpret_cast<const float*>(lds + threadIdx.x * 4) == 123.456f) sink[0] = 1.f;
}

template <bool CONTIG, int RING>
static void run(const char* pool, int n_mat, float* sink, const char* name) {
    const int64_t mat = (int64_t)256 * ROWS * KB * 2;
    hipFuncSetAttribute((const void*)stream<CONTIG, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, RING * COPIES * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream<CONTIG, RING>), dim3(256), dim3(512), RING * COPIES * 1024, 0, pool, sink);
    hipEventRecord(a, 0);
    for (int i = 1; i < n_mat; ++i) hipLaunchKernelGGL((stream<CONTIG, RING>), dim3(256), dim3(512), RING * COPIES * 1024, 0, pool + i * mat, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  %-16s ring %d (%3d KiB in flight per CU): %6.1f us per 294 MB matrix = %5.2f TB/s\n", name, RING, (RING - 1) * COPIES, ms * 1e3 / (n_mat - 1), (double)mat * (n_mat - 1) / ms / 1e9);
}

int main() {
    const int n_mat = 9;
    const int64_t mat = (int64_t)256 * ROWS * KB * 2;
    char* pool; float* sink;
    hipMalloc(&pool, mat * n_mat); hipMemset(pool, 1, mat * n_mat); hipMalloc(&sink, 4);
    printf("cold weight stream, 256 workgroups x (160 rows x 3584) bf16:\n");
    run<false, 2>(pool, n_mat, sink, "row-strided"); run<false, 3>(pool, n_mat, sink, "row-strided"); run<false, 5>(pool, n_mat, sink, "row-strided");
    run<true, 2>(pool, n_mat, sink, "tile-contiguous"); run<true, 3>(pool, n_mat, sink, "tile-contiguous"); run<true, 5>(pool, n_mat, sink, "tile-contiguous");
    return 0;
}
