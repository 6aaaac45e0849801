// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = element index; prints per-lane results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(int mode, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t addr;
    if (mode == 0) addr = l * 8;                                   // each lane: its own 4 consecutive elements
    else if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 8;       // row = l&15 (row stride 32 elems), col group = l>>4
    else if (mode == 2) addr = (l >> 2) * 64 + (l & 3) * 8;        // 4 lanes per row
    else addr = 0;
    uint32_t base = (uint32_t)(uintptr_t)lds;                      // LDS byte address of the array
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base + addr) : "memory");
    out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
