// gemm_tiles_layout.hip — contraction-major operand forms (dX = dY W, dW = dY^T X) of the 256x256 tile.
#include "gemm_tile_kernel.h"

extern int g_train_variant;       // gemm.hip: 40 = the 4-wave hand-scheduled tile (gemm_asm4.hip), which also has the contraction-major forms
int st_gemm_asm4_nn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, int64_t ldc, int M, int N, int K, hipStream_t s);
int st_gemm_asm4_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s);

/* out[M,N] (bf16) = A[M,K] B[K,N] with B contraction-major (row pitch ldb): the dX = dY W form.  K % 64 == 0, N % 8 == 0. */
extern "C" int st_gemm_nn(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, st_bf16* out, int64_t ldc, int M, int N, int K,
                          st_stream_t stream) {
    if (!A || !B || !out || M <= 0 || N < 8 || K <= 0 || (K % 64) || (N & 7) || (lda & 7) || (ldb & 7) || lda < K || ldb < N || ldc < N ||
        (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM, s, 2.0 * (double)M * (double)N * (double)K, st_prof_tag(1, 0, M, N, K));
    if (g_train_variant == 40 && M > 256 && lda < (1 << 22) && ldb < (1 << 22)) return st_gemm_asm4_nn(A, lda, B, ldb, out, ldc, M, N, K, s);   // 32-bit buffer offsets: larger pitches take the 8-wave tile
    return launch_tile_layout<false, true, true, false>(A, lda, B, ldb, out, nullptr, ldc, M, N, K, s);
}

/* out_f32[M,N] (+)= A[K,M]^T B[K,N], BOTH operands contraction-major: the dW = dY^T X form.  K % 64 == 0, M % 8 == 0, N % 8 == 0. */
extern "C" int st_gemm_tn(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, float* out_f32, int64_t ldc, int accumulate, int M,
                          int N, int K, st_stream_t stream) {
    if (!A || !B || !out_f32 || M < 8 || N < 8 || K <= 0 || (K % 64) || (M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || lda < M || ldb < N ||
        ldc < N || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM, s, 2.0 * (double)M * (double)N * (double)K, st_prof_tag(2, 4 | (accumulate ? 8 : 0), M, N, K));
    if (g_train_variant == 40 && lda < (1 << 22) && ldb < (1 << 22)) return st_gemm_asm4_tn(A, lda, B, ldb, out_f32, ldc, accumulate, M, N, K, s);
    if (accumulate) return launch_tile_layout<true, true, false, true>(A, lda, B, ldb, nullptr, out_f32, ldc, M, N, K, s);
    return launch_tile_layout<true, true, false, false>(A, lda, B, ldb, nullptr, out_f32, ldc, M, N, K, s);
}

