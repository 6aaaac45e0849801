// gemm_tiles_decode.hip — decode-shaped (M <= 512, weight-streaming) instantiations of the tile family.
#include "gemm_tile_kernel.h"

// Decode-shaped launches (M <= 256 rows, weight streaming): small-M tiles with a 3-slot ring and optional split-K into fp32
// slabs [split][M][N] (ldc = N) that gemm_skinny_finish sums in a fixed order.
//   10 = 64x64 1x4   11 = 64x128 1x4   12 = 64x256 1x4   13 = 128x64 2x2   14 = 128x128 2x2   15 = 256x64 4x1
//   16 = 256x128 4x2 (3 slots)   17 = 256x128 4x2 (2 slots)   18 = 256x256 4x2 (2 slots)   19 = 128x128 2x2 (2 slots)
//   20 = 128x256 2x4 (2 slots)   21 = 64x128 1x4 (2 slots)   22 = 256x192 4x2 (2 slots)
static inline int BM_OF(int variant) { return (variant >= 10 && variant <= 12) || variant == 21 ? 64 : ((variant == 13 || variant == 14 || variant == 19 || variant == 20) ? 128 : 256); }

int st_gemm_tile_decode(int variant, int splits, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                        const uint16_t* res, int64_t ldr, uint16_t* Cb, float* slabs, int M, int N, int K, int64_t ldc, hipStream_t s) {
#define DEC_GO_NT(BM, BN, WM, WN, ST, NT)                                                                                        \
    do {                                                                                                                         \
        if (splits > 1 || splits < 0) return launch_tile<BM, BN, WM, WN, ST, false, false, false, false, false, false, false, NT>(A, lda, B, ldb, nullptr, nullptr, 0, nullptr, slabs, N, M, N, K, s, splits < 0 ? 1 : splits, (int64_t)M * N); \
        if (bias && res) return launch_tile<BM, BN, WM, WN, ST, true, true, true, false, false, false, false, NT>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);   \
        if (bias) return launch_tile<BM, BN, WM, WN, ST, true, false, true, false, false, false, false, NT>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);         \
        if (res) return launch_tile<BM, BN, WM, WN, ST, false, true, true, false, false, false, false, NT>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);          \
        return launch_tile<BM, BN, WM, WN, ST, false, false, true, false, false, false, false, NT>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);                  \
    } while (0)
    // weights read by ONE row tile stream with the non-temporal policy (nothing re-reads them before the next decode iteration);
    // with two row tiles (257..512 rows) the second tile's read is an L2 / MALL hit that nt would throw away
    // (ST_DECODE_NT=2: nt with two row tiles as well — an A/B switch, profiles/r05_notes.md)
    const bool nt = g_decode_nt >= 2 || (g_decode_nt && M <= BM_OF(variant));
#define DEC_GO(BM, BN, WM, WN, ST) do { if (nt) DEC_GO_NT(BM, BN, WM, WN, ST, true); else DEC_GO_NT(BM, BN, WM, WN, ST, false); } while (0)
    switch (variant) {
        case 10: DEC_GO(64, 64, 1, 4, 3);
        case 11: DEC_GO(64, 128, 1, 4, 3);
        case 12: DEC_GO_NT(64, 256, 1, 4, 3, false);          // tuning-only tile: default cache policy
        case 13: DEC_GO(128, 64, 2, 2, 3);
        case 14: DEC_GO(128, 128, 2, 2, 3);
        case 15: DEC_GO_NT(256, 64, 4, 1, 3, false);          // tuning-only tile: default cache policy
        case 16: DEC_GO(256, 128, 4, 2, 3);
        case 17: DEC_GO_NT(256, 128, 4, 2, 2, false);          // tuning-only tile: default cache policy
        case 18: DEC_GO(256, 256, 4, 2, 2);
        case 19: DEC_GO_NT(128, 128, 2, 2, 2, false);          // tuning-only tile: default cache policy
        case 20: DEC_GO_NT(128, 256, 2, 4, 2, false);          // tuning-only tile: default cache policy
        case 21: DEC_GO_NT(64, 128, 1, 4, 2, false);          // tuning-only tile: default cache policy
        case 22: DEC_GO_NT(256, 192, 4, 2, 2, false);          // tuning-only tile: default cache policy
        // 28 = the training tile (256x256, mid-tile barrier schedule; bf16 outputs through the LDS-staged epilogue) with split-K slabs:
        // 257..512-row decode batches have 2 row tiles, and few column tiles x many K-slices of this tile beat the 256x128 ring
        case 28:
            if (splits > 1 || splits < 0) return launch_tile<256, 256, 4, 2, 2, false, false, false, false, true>(A, lda, B, ldb, nullptr, nullptr, 0, nullptr, slabs, N, M, N, K, s, splits < 0 ? 1 : splits, (int64_t)M * N);
            if (bias && res) return launch_tile<256, 256, 4, 2, 2, true, true, true, false, true, true>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);
            if (bias) return launch_tile<256, 256, 4, 2, 2, true, false, true, false, true, true>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);
            if (res) return launch_tile<256, 256, 4, 2, 2, false, true, true, false, true, true>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);
            return launch_tile<256, 256, 4, 2, 2, false, false, true, false, true, true>(A, lda, B, ldb, bias, res, ldr, Cb, nullptr, ldc, M, N, K, s);
        default: return ST_EINVAL;
    }
#undef DEC_GO
}

