// gemm_tiles_train.hip — training-shape instantiations of the tile family (st_gemm_nt / st_gemm_nt_variant).
#include "gemm_tile_kernel.h"

int st_gemm_asm4_debug(int dbg, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, int64_t ldc, int M, int N, int K, hipStream_t s);
int st_gemm_asm4_dispatch(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res, int64_t ldr,
                          uint16_t* Cb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s);

// variant ids: 0 = 128x128 2x2 waves 2 stages, 1 = 128x128 3 stages, 2 = 256x128 4x2 2 stages, 3 = 256x128 4x2 3 stages,
//              4 = 256x256 4x2 2 stages, 5 = 128x256 2x4 3 stages, 6 / 7 = 256x256 / 128x128 with the mid-tile barrier schedule
int st_gemm_tile_dispatch(int variant, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                          const uint16_t* res, int64_t ldr, uint16_t* Cb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K,
                          hipStream_t s) {
#define TILE_GO(BM, BN, WM, WN, ST, MB) TILE_GO_PP(BM, BN, WM, WN, ST, MB, false, false)
#define TILE_GO_LE(BM, BN, WM, WN, ST, MB, LE) TILE_GO_PP(BM, BN, WM, WN, ST, MB, LE, false)
#define TILE_GO_PP(BM, BN, WM, WN, ST, MB, LE, PPV)                                                                                   \
    do {                                                                                                                         \
        if (Cb) {                                                                                                                \
            if (bias && res) return launch_tile<BM, BN, WM, WN, ST, true, true, true, false, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);   \
            if (bias) return launch_tile<BM, BN, WM, WN, ST, true, false, true, false, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);         \
            if (res) return launch_tile<BM, BN, WM, WN, ST, false, true, true, false, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);          \
            return launch_tile<BM, BN, WM, WN, ST, false, false, true, false, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);                  \
        }                                                                                                                        \
        if (accumulate) return launch_tile<BM, BN, WM, WN, ST, false, false, false, true, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);      \
        return launch_tile<BM, BN, WM, WN, ST, false, false, false, false, MB, LE, PPV>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, s);                     \
    } while (0)
    switch (variant) {
        case 41: case 42: case 43: case 44: case 45: case 46: case 47: return st_gemm_asm4_debug(variant - 40, A, lda, B, ldb, Cb, ldc, M, N, K, s);   // timing experiments (wrong results)
        case 40: return st_gemm_asm4_dispatch(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, accumulate, M, N, K, s);   // 4 waves x 128x128, hand-scheduled K loop (gemm_asm4.hip)
        case 0: TILE_GO(128, 128, 2, 2, 2, false);
        case 1: TILE_GO(128, 128, 2, 2, 3, false);
        case 2: TILE_GO(256, 128, 4, 2, 2, false);
        case 3: TILE_GO(256, 128, 4, 2, 3, false);
        case 4: TILE_GO(256, 256, 4, 2, 2, false);
        case 5: TILE_GO(128, 256, 2, 4, 3, false);
        case 6: TILE_GO(256, 256, 4, 2, 2, true);
        case 7: TILE_GO(128, 128, 2, 2, 2, true);
        case 8: TILE_GO(256, 256, 2, 2, 2, true);          // 4 waves x (128 x 128): one wave per SIMD, accumulators fill the AGPRs
        case 9: TILE_GO(256, 256, 2, 2, 2, false);
        case 23: TILE_GO_LE(256, 256, 4, 2, 2, true, true);   // variant 6 with the LDS-staged epilogue
        case 31: TILE_GO_PP(256, 256, 4, 2, 2, true, true, true);   // variant 23 on the ping-pong schedule
        default: return ST_EINVAL;
    }
#undef TILE_GO
#undef TILE_GO_LE
#undef TILE_GO_PP
}

