// Do MFMA and VALU work overlap on a gfx950 SIMD — inside one wave, and between the two waves that share a SIMD?
// Each wave runs ITER rounds of NM v_mfma_f32_32x32x16_bf16 (4 independent accumulators) and NV VALU ops (v_exp_f32 / v_fma_f32 on
// 16 independent chains):
//   mode 0: MFMAs only          mode 1: VALU only
//   mode 2: both, in two phases (sched_barrier between them)      mode 3: both, interleaved 1 MFMA : NV/NM VALU
// launched with 1 wave per SIMD (256 workgroups of 256 threads, 96 KiB LDS each) and 2 waves per SIMD (512 workgroups, 64 KiB LDS).
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define NM 16
#define NV 128

template <int I, int N, int PER>
__device__ __forceinline__ void pin_mix() {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
    if constexpr (I + 1 < N) pin_mix<I + 1, N, PER>();
}

template <int MODE, int LDS_KB, bool USE_EXP>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ char pad[LDS_KB * 1024];
    if (iters < 0) out[threadIdx.x] = pad[threadIdx.x];
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (short)(0x3f80 + threadIdx.x); y[i] = (short)(0x3c00 + i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE != 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
        }
        if constexpr (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE != 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (USE_EXP) v[j & 15] = __builtin_amdgcn_exp2f(v[j & 15]);
                else v[j & 15] = __builtin_fmaf(v[j & 15], 1.0001f, 0.5f);
            }
        }
        if constexpr (MODE == 3) pin_mix<0, NM, NV / NM>();
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int LDS_KB, bool USE_EXP>
static float run(float* d, int grid, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, LDS_KB, USE_EXP>), dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((probe<MODE, LDS_KB, USE_EXP>), dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f;
}

template <bool USE_EXP>
static void suite(float* d, int iters) {
    printf("VALU op = %s, per round: %d MFMA 32x32x16 (%d matrix-pipe cycles) + %d VALU ops\n", USE_EXP ? "v_exp_f32 (quarter rate)" : "v_fma_f32", NM, NM * 32, NV);
    const float m1 = run<0, 96, USE_EXP>(d, 256, iters), v1 = run<1, 96, USE_EXP>(d, 256, iters);
    const float p1 = run<2, 96, USE_EXP>(d, 256, iters), i1 = run<3, 96, USE_EXP>(d, 256, iters);
    const float m2 = run<0, 64, USE_EXP>(d, 512, iters), v2 = run<1, 64, USE_EXP>(d, 512, iters);
    const float p2 = run<2, 64, USE_EXP>(d, 512, iters), i2 = run<3, 64, USE_EXP>(d, 512, iters);
    printf("  1 wave / SIMD: mfma %8.1f us  valu %8.1f us  phased %8.1f us  interleaved %8.1f us\n", m1, v1, p1, i1);
    printf("  2 waves/ SIMD: mfma %8.1f us  valu %8.1f us  phased %8.1f us  interleaved %8.1f us   (twice the work)\n", m2, v2, p2, i2);
}

int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    const int iters = 2000;
    suite<false>(d, iters);
    suite<true>(d, iters);
    return 0;
}
