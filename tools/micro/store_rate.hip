// Per-CU global store throughput on gfx950: one 256-thread workgroup per CU writes 128-KiB "tiles" from registers with different
// instruction widths / address patterns / cache policies.  Prints bytes per clock per CU (wall time x 2.1 GHz nominal) and GB/s per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o gpurun_out/store_rate && gpurun_out/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(uint8_t* base, int64_t region, int tiles) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint8_t* wg = base + (int64_t)blockIdx.x * region;
    uint4 v = make_uint4(t, blockIdx.x, 3, 4);
    for (int it = 0; it < tiles; ++it) {
        uint8_t* tile = wg + (int64_t)(it & 15) * 131072;              // 16 tiles = 2 MiB per workgroup, revisited
        if constexpr (MODE == 0) {                                       // the GEMM epilogue's pattern: 16 lanes x 16 B in two 128-B halves 256 B apart, 4 rows (512-B pitch) per instruction
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = i * 16 + (t >> 4), c8 = (t & 15) * 8, col = (c8 >> 6) * 128 + p * 64 + (c8 & 63);
                    *reinterpret_cast<uint4*>(tile + row * 512 + col * 2) = v;
                }
        } else if constexpr (MODE == 1) {                                // fully contiguous: a wave writes 1 KiB, the workgroup 4 KiB per instruction
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(tile + i * 4096 + t * 16) = v;
        } else if constexpr (MODE == 2) {                                // dwordx2, contiguous
#pragma unroll
            for (int i = 0; i < 64; ++i) *reinterpret_cast<uint2*>(tile + i * 2048 + t * 8) = make_uint2(v.x, v.y);
        } else if constexpr (MODE == 3) {                                // dword, contiguous
#pragma unroll 32
            for (int i = 0; i < 128; ++i) *reinterpret_cast<uint32_t*>(tile + i * 1024 + t * 4) = v.x;
        } else if constexpr (MODE == 4) {                                // contiguous dwordx4, nontemporal
#pragma unroll
            for (int i = 0; i < 32; ++i) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u4*>(tile + i * 4096 + t * 16)); }
        } else if constexpr (MODE == 5) {                                // each wave owns a contiguous 32-KiB quarter: 1 KiB per instruction per wave
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(tile + wave * 32768 + i * 1024 + lane * 16) = v;
        } else if constexpr (MODE == 6) {                                // same tile every time (write hits in L2)
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(wg + i * 4096 + t * 16) = v;
        } else if constexpr (MODE == 7) {                                // only wave 0 stores (one wave's issue rate)
            if (wave == 0) {
#pragma unroll
                for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(tile + i * 1024 + lane * 16) = v;
            }
        }
        v.x += 1;
    }
}

template <int MODE>
static void run(const char* tag, uint8_t* buf, int64_t region, int grid, int tiles, double bytes_per_tile) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    store_kernel<MODE><<<grid, 256>>>(buf, region, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    store_kernel<MODE><<<grid, 256>>>(buf, region, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_tile_us = ms * 1e3 / tiles, gbs = bytes_per_tile / (per_tile_us * 1e-6) / 1e9;
    printf("grid %3d  %-42s %7.2f us per 128-KiB tile  %6.1f GB/s per CU  %5.1f B/clk @2.1GHz  (%.2f TB/s chip)\n", grid, tag, per_tile_us * (131072.0 / bytes_per_tile), gbs, gbs / 2.1, gbs * grid / 1e3);
}

int main() {
    const int64_t region = 2 << 20;
    uint8_t* buf; hipMalloc(&buf, region * 256);
    for (int grid : {1, 8, 64, 256}) {
        const int tiles = 400;
        run<0>("dwordx4, epilogue pattern (8 x 128 B)", buf, region, grid, tiles, 131072);
        run<1>("dwordx4, contiguous 4 KiB / WG instr", buf, region, grid, tiles, 131072);
        run<5>("dwordx4, wave-contiguous quarters", buf, region, grid, tiles, 131072);
        run<2>("dwordx2, contiguous", buf, region, grid, tiles, 131072);
        run<3>("dword, contiguous", buf, region, grid, tiles, 131072);
        run<4>("dwordx4 nontemporal, contiguous", buf, region, grid, tiles, 131072);
        run<6>("dwordx4, same 128 KiB every time", buf, region, grid, tiles, 131072);
        run<7>("dwordx4, one wave only (32 KiB)", buf, region, grid, tiles, 32768);
    }
    return 0;
}
