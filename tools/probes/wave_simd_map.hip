// Which SIMD does wave w of a 512-thread workgroup run on (gfx950)?  hipcc --offload-arch=gfx950 wave_simd_map.hip -o wave_simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4); unsigned h[64 * 8];
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 65536, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d: simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
