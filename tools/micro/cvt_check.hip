// v_cvt_pk_bf16_f32 against the integer round-to-nearest-even formulation, over all 2^32 fp32 bit patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ uint16_t soft(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__global__ void k(unsigned long long* out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, n = gridDim.x * (uint64_t)blockDim.x;
    unsigned long long diff = 0, diff_nan = 0, diff_den = 0, first = ~0ull;
    for (uint64_t u = tid; u < (1ull << 32); u += n) {
        const float f = __uint_as_float((uint32_t)u);
        const uint16_t a = soft(f), b = __builtin_bit_cast(uint16_t, (__bf16)f);
        if (a != b) {
            const uint32_t mag = (uint32_t)u & 0x7fffffffu;
            if (mag > 0x7f800000u) ++diff_nan; else if (mag < 0x00800000u) ++diff_den; else { ++diff; if (u < first) first = u; }
        }
    }
    atomicAdd(&out[0], diff); atomicAdd(&out[1], diff_nan); atomicAdd(&out[2], diff_den); atomicMin(&out[3], first);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 32); unsigned long long h[4] = {0, 0, 0, ~0ull};
    hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    k<<<4096, 256>>>(d); hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("normal/inf patterns that differ: %llu (first 0x%llx); NaN patterns that differ: %llu; fp32-denormal patterns that differ: %llu\n", h[0], h[3], h[1], h[2]);
    return 0;
}
