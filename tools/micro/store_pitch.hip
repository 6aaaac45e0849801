// Store throughput into a 256 x 256 bf16 tile of a LARGE row-major matrix (row pitch 75,776 B: every tile row lies in a different
// 64-KiB page fragment), the GEMM epilogue's situation, against the same bytes written contiguously.  One 256-thread workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pitch.hip -o /tmp/store_pitch && /tmp/store_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int64_t T = 28672, N = 37888, PITCH = N * 2;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t* base, int tiles, int stride) {
    const int t = threadIdx.x;
    uint4 v = make_uint4(t, blockIdx.x, 3, 4);
    for (int it = 0; it < tiles; ++it) {
        const int id = blockIdx.x + it * stride;                         // fresh tile every time
        const int tm = id % 112, tn = (id / 112) % 148;
        uint8_t* tile = base + (int64_t)tm * 256 * PITCH + tn * 512;
        if constexpr (MODE == 0) {                                       // epilogue as built: two passes, 4 rows x (2 x 128 B) per wave instruction
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = i * 16 + (t >> 4), c8 = (t & 15) * 8, col = (c8 >> 6) * 128 + p * 64 + (c8 & 63);
                    *reinterpret_cast<uint4*>(tile + row * PITCH + col * 2) = v;
                }
        } else if constexpr (MODE == 1) {                                // one pass, 2 full rows (512 B each) per wave instruction
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = i * 8 + (t >> 5);
                *reinterpret_cast<uint4*>(tile + row * PITCH + (t & 31) * 16) = v;
            }
        } else if constexpr (MODE == 2) {                                // one pass, each wave walks its own 64 consecutive rows
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = (t >> 6) * 64 + i * 2 + ((t >> 5) & 1);
                *reinterpret_cast<uint4*>(tile + row * PITCH + (t & 31) * 16) = v;
            }
        } else if constexpr (MODE == 3) {                                // same bytes, contiguous 128 KiB (fresh every time)
            uint8_t* lin = base + (int64_t)(id % 16384) * 131072;
#pragma unroll
            for (int i = 0; i < 32; ++i) *reinterpret_cast<uint4*>(lin + i * 4096 + t * 16) = v;
        } else if constexpr (MODE == 4) {                                // pitch 4096 (one page per row... rows 4 KiB apart): 256 x 512 B
            uint8_t* lin = base + (int64_t)(id % 2048) * (256 * 4096);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = i * 8 + (t >> 5);
                *reinterpret_cast<uint4*>(lin + row * 4096 + (t & 31) * 16) = v;
            }
        }
        v.x += 1;
    }
}

template <int MODE>
static void run(const char* tag, uint8_t* buf, int grid, int tiles) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(buf, 4, grid);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(buf, tiles, grid);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / tiles;
    printf("grid %3d  %-58s %7.2f us per tile  %6.1f GB/s per CU  (%.2f TB/s chip)\n", grid, tag, us, 131072.0 / us / 1e3, 131072.0 / us / 1e3 * grid / 1e3);
}

int main() {
    uint8_t* buf; if (hipMalloc(&buf, T * PITCH) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, T * PITCH);
    for (int grid : {1, 8, 256}) {
        const int tiles = grid == 256 ? 64 : 400;
        run<0>("matrix tile, two passes of 4 rows x 2 x 128 B per instr", buf, grid, tiles);
        run<1>("matrix tile, one pass, 2 full rows per instr", buf, grid, tiles);
        run<2>("matrix tile, one pass, wave-private row ranges", buf, grid, tiles);
        run<3>("contiguous 128 KiB, fresh", buf, grid, tiles);
        run<4>("256 rows x 512 B at 4-KiB pitch", buf, grid, tiles);
    }
    return 0;
}
