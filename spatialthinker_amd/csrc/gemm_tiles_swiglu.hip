// gemm_tiles_swiglu.hip — gate/up projection with the SwiGLU epilogue (training and decode).
#include "gemm_tile_kernel.h"
#include <stdlib.h>

extern int g_train_variant;
int st_gemm_asm4_swiglu(const uint16_t* A, int64_t lda, const uint16_t* gate_up_w, int64_t ldb, uint16_t* gu_out, int64_t ldgu, uint16_t* m_out,
                        int64_t ldm, int M, int I, int K, hipStream_t s, int split_tail = 0);
// gemm_swiglu512.hip: all (<= 512) rows x 80 output columns per workgroup, ONE pass over the weights (round 5)
int st_gemm_swiglu512_launch(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* out, int64_t ldo, int M, int I, int K,
                             hipStream_t s);
// gemm_tiles_swiglu_small.hip: <= 128 rows, 160 weight rows per tile = 237 tiles, one per CU (variants 8 = 64 rows, 9 = 128 rows)
int st_gemm_swiglu_small(int variant, const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* out, int64_t ldo, int M, int I, int K,
                         hipStream_t s);
// 257..512 rows: plan 512 = the one-pass tile (ST_DECODE_GU512=0 falls back to the two-round 256x160 tile)
static bool swiglu_decode_on_512(int M) {
    static const bool on = [] { const char* e = getenv("ST_DECODE_GU512"); return !e || e[0] != '0'; }();
    return on && M > 256 && M <= 512;
}

/* gate/up projection with the SwiGLU in the epilogue for any M (training / prefill): m_out[M, I] = silu(A gate_w^T) * (A up_w^T);
 * gu_out (optional, [M, 2I]) additionally receives the bf16 gate|up values the backward needs. */
extern "C" int st_gemm_swiglu(const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* gu_out, int64_t ldgu,
                              st_bf16* m_out, int64_t ldm, int M, int I, int K, st_stream_t stream) {
    if (!A || !gate_up_w || !m_out || M <= 0 || I <= 0 || K <= 0 || (K % 64) || (lda & 7) || (ldb & 7) || lda < K || ldb < K || ldm < I ||
        (gu_out && ldgu < 2 * (int64_t)I) || (((uintptr_t)A) & 15) || (((uintptr_t)gate_up_w) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM, s, 2.0 * (double)M * (double)(2 * I) * (double)K, st_prof_tag(3, gu_out ? 1 : 0, M, 2 * I, K));
    if (g_train_variant == 40 && (int64_t)st_cdiv(M, 256) * st_cdiv(I, 128) >= 128 && lda < (1 << 22) && ldb < (1 << 22))
        return st_gemm_asm4_swiglu(A, lda, gate_up_w, ldb, gu_out, ldgu, m_out, ldm, M, I, K, s);
    return launch_tile_swiglu<256, 256, 4, 2, 2, true>(A, lda, gate_up_w, ldb, m_out, ldm, M, I, K, s, gu_out, ldgu);
}

/* tuning entry: the decode gate/up + SwiGLU GEMM on an explicit tile (tools/decode_swiglu_tune.py):
 * 1 = 256x160 3 slots (8-column interleave)   2 = 256x192 2 slots   3 = 256x256 2 slots   4 = 256x256 mid-tile barrier (training tile)
 * 5 = 256x192 mid-tile barrier   6 = 128x128 3 slots   7 = 64x128 3 slots   40 = 4-wave training tile + K-split SwiGLU tail */
extern "C" int st_gemm_swiglu_decode_variant(int variant, const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* out,
                                             int64_t ldc, int M, int I, int K, st_stream_t stream) {
    if (!A || !gate_up_w || !out || M <= 0 || M > ST_DECODE_MAX_ROWS || I <= 0 || K <= 0 || (K % 64) || (lda & 7) || (ldb & 7) || lda < K || ldb < K ||
        ldc < I || (((uintptr_t)A) & 15) || (((uintptr_t)gate_up_w) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 40:                                             // the 4-wave training tile with the K-split SwiGLU tail (needs st_gemm_set_workspace for the split)
            if (lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
            return st_gemm_asm4_swiglu(A, lda, gate_up_w, ldb, nullptr, 0, out, ldc, M, I, K, s, 1);
        case 512: return st_gemm_swiglu512_launch(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 8: case 9: return st_gemm_swiglu_small(variant, A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 1: return launch_tile_swiglu<256, 160, 4, 2, 3, false, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 2: return launch_tile_swiglu<256, 192, 4, 2, 2>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 3: return launch_tile_swiglu<256, 256, 4, 2, 2>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 4: return launch_tile_swiglu<256, 256, 4, 2, 2, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 5: return launch_tile_swiglu<256, 192, 4, 2, 2, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 6: return launch_tile_swiglu<128, 128, 2, 2, 3>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 7: return launch_tile_swiglu<64, 128, 1, 4, 3>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        default: return ST_EINVAL;
    }
}

// Tile choice of the decode gate/up + SwiGLU GEMM (ids of st_gemm_swiglu_decode_variant).
// One row tile (M <= 256) or two (257..512): the only freedom is the column tile.  Cost ~ rounds over 256 CUs x tile width; the 7B
// gate/up (I = 18944) gives 148 tiles at 128 output columns (0.58 of the CUs) and 198 at 96 (-10 % time).  A 256x160 tile as 8x1
// waves (237 tiles, one full round) measured SLOWER (+6 % decode step): each wave re-reads every B fragment from LDS.
// 256x160 with the 8-column interleave (id 1): 3-slot ring (two K-tiles of weights in flight: the 2-slot variants sit parked on HBM
// latency half of the time) and 237 workgroups for I = 18944: 90 us vs 102 us (256x192) on MI355X.
// Round 5, <= 128 rows: a decode tile runs at the rate its CU stages bytes into the LDS (profiles/r05_notes.md §1b), so the launch takes what
// the BUSIEST CU stages: rounds of tiles over the CUs x (rows + weight rows) of a tile.  7B: 296 tiles of 128 weight rows put two tiles on 40
// CUs (64 rows: 61-68 us, 128 rows: 79-84 us); 237 tiles of 160 (ids 8 / 9, the 8-column interleave) put one on each: 44-47 / 53-55 us,
// weights at 5.8-6.1 TB/s.  3B (138 vs 172 tiles, both under one round) keeps the narrower tile.
static int swiglu_decode_plan(int M, int I) {
    if (M <= 128) {
        const int ncu = st_num_cus(), bm = M <= 64 ? 64 : 128;
        const int c128 = st_cdiv(st_cdiv(2 * I, 128), ncu) * (bm + 128), c160 = st_cdiv(st_cdiv(2 * I, 160), ncu) * (bm + 160);
        static const bool wide_ok = [] { const char* e = getenv("ST_DECODE_GU_SMALL160"); return !e || e[0] != '0'; }();
        if (wide_ok && c160 < c128) return M <= 64 ? 8 : 9;
        return M <= 64 ? 7 : 6;
    }
    auto cost = [&](int cols) { const int t = st_cdiv(M, 256) * st_cdiv(I, cols); return (double)st_cdiv(t, 256) * (cols + 24); };
    if (cost(80) <= cost(96) && cost(80) <= cost(128)) return 1;
    if (cost(96) <= cost(128)) return 2;
    return 3;
}
// 257..512 rows (two row tiles) on the 4-wave training tile with the SwiGLU epilogue and a K-split tail (round 4; plan id 40) once it
// fills at least one round of CUs — 7B: 2 x 148 = 296 tiles, 256 whole + 40 cut into K-slices.  OPT-IN (ST_DECODE_GU_ASM4=1, or variant 40
// of st_gemm_swiglu_decode_variant): measured 1 % faster per decode iteration (10.42 vs 10.53 ms at 512 rows, same box) but 4 GB MORE HBM
// traffic per iteration (the fp32 K-slices of the tail: 16.1 vs 12.0 GB for gate/up, PMC) — the 256x160 decode tile stays the default.
static bool swiglu_decode_on_asm4(int M, int I) {
    static const bool gu_asm4 = [] { const char* e = getenv("ST_DECODE_GU_ASM4"); return e && e[0] == '1'; }();
    return gu_asm4 && M > 256 && g_train_variant == 40 && (int64_t)st_cdiv(M, 256) * st_cdiv(I, 128) >= st_num_cus();
}
extern "C" int st_gemm_swiglu_decode_plan(int M, int I, int* variant_out) {
    if (M <= 0 || M > ST_DECODE_MAX_ROWS || I <= 0 || !variant_out) return ST_EINVAL;
    *variant_out = swiglu_decode_on_asm4(M, I) ? 40 : (swiglu_decode_on_512(M) ? 512 : swiglu_decode_plan(M, I));
    return 0;
}

extern "C" int st_gemm_swiglu_decode(const st_bf16* A, int64_t lda, const st_bf16* gate_up_w, int64_t ldb, st_bf16* out, int64_t ldc,
                                     int M, int I, int K, st_stream_t stream) {
    if (!A || !gate_up_w || !out || M <= 0 || M > ST_DECODE_MAX_ROWS || I <= 0 || K <= 0 || (K % 64) || (lda & 7) || (ldb & 7) || lda < K || ldb < K ||
        ldc < I || (((uintptr_t)A) & 15) || (((uintptr_t)gate_up_w) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (swiglu_decode_on_asm4(M, I) && lda < (1 << 22) && ldb < (1 << 22))
        return st_gemm_asm4_swiglu(A, lda, gate_up_w, ldb, nullptr, 0, out, ldc, M, I, K, s, 1);
    if (swiglu_decode_on_512(M)) return st_gemm_swiglu512_launch(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
    const int plan = swiglu_decode_plan(M, I);
    if (plan == 8 || plan == 9) return st_gemm_swiglu_small(plan, A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
    if (g_decode_nt >= 2 || (g_decode_nt && M <= 256)) {     // one row tile: the weights are read once — non-temporal stream (2: always, A/B)
        if (plan == 7) return launch_tile_swiglu<64, 128, 1, 4, 3, false, false, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        if (plan == 6 && M <= 128) return launch_tile_swiglu<128, 128, 2, 2, 3, false, false, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        if (plan == 1) return launch_tile_swiglu<256, 160, 4, 2, 3, false, true, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
    }
    switch (plan) {
        case 7: return launch_tile_swiglu<64, 128, 1, 4, 3>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 6: return launch_tile_swiglu<128, 128, 2, 2, 3>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 1: return launch_tile_swiglu<256, 160, 4, 2, 3, false, true>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        case 2: return launch_tile_swiglu<256, 192, 4, 2, 2>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
        default: return launch_tile_swiglu<256, 256, 4, 2, 2>(A, lda, gate_up_w, ldb, out, ldc, M, I, K, s);
    }
}

