// gemm_tiles_swiglu_small.hip — decode gate/up + SwiGLU tiles for <= 128 rows with ONE workgroup per CU (round 5).
//
// The 64x128 / 128x128 tiles of rounds 2-4 cut the 7B gate/up (37888 weight rows) into 296 column tiles: on 256 CUs the 40 CUs that run two of
// them set the launch time (a decode tile runs at the rate its CU can stage bytes into the LDS, profiles/r05_notes.md §1b).  The 8-column
// SwiGLU interleave (SW8) allows any tile width that is a multiple of 16: 160 weight rows = 80 output columns give 237 tiles, one per CU.
#include "gemm_tile_kernel.h"

// variant 8 = 64 x 160 (2 x 2 waves, 3 slots), 9 = 128 x 160 (2 x 2 waves, 3 slots); weights streamed with the nt policy (one row tile)
int st_gemm_swiglu_small(int variant, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* out, int64_t ldo, int M, int I, int K,
                         hipStream_t s) {
    if (variant == 8) {
        if (g_decode_nt) return launch_tile_swiglu<64, 160, 2, 2, 3, false, true, true>(A, lda, W, ldw, out, ldo, M, I, K, s);
        return launch_tile_swiglu<64, 160, 2, 2, 3, false, true, false>(A, lda, W, ldw, out, ldo, M, I, K, s);
    }
    if (variant == 9) {
        if (g_decode_nt) return launch_tile_swiglu<128, 160, 2, 2, 3, false, true, true>(A, lda, W, ldw, out, ldo, M, I, K, s);
        return launch_tile_swiglu<128, 160, 2, 2, 3, false, true, false>(A, lda, W, ldw, out, ldo, M, I, K, s);
    }
    return ST_EINVAL;
}
