// gemm_tiles.hip — workspace registration and the explicit-variant entry of the tile family (kernel: gemm_tile_kernel.h; the
// instantiations are spread over gemm_tiles_{train,decode,swiglu,layout}.hip so that they compile in parallel).
#include "gemm_tile_kernel.h"

float* g_tail_ws = nullptr;
int64_t g_tail_ws_bytes = 0;
int g_decode_nt = [] { const char* e = getenv("ST_DECODE_NT"); return e ? atoi(e) : 1; }();

extern "C" int st_gemm_set_workspace(void* ws, int64_t bytes) {
    g_tail_ws = reinterpret_cast<float*>(ws);
    g_tail_ws_bytes = ws ? bytes : 0;
    return 0;
}


int st_gemm_tile_dispatch(int variant, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                          const uint16_t* res, int64_t ldr, uint16_t* Cb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K,
                          hipStream_t s);

extern "C" int st_gemm_nt_variant(int variant, const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
                                  const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, float* out_f32, int64_t ldc, int accumulate,
                                  int M, int N, int K, st_stream_t stream) {
    if (!A || !B || M <= 0 || N <= 0 || K <= 0 || (K % 64) || (lda & 7) || (ldb & 7) || ((out_bf16 == nullptr) == (out_f32 == nullptr)))
        return ST_EINVAL;
    if (out_f32 && (bias || residual)) return ST_EINVAL;
    return st_gemm_tile_dispatch(variant, A, lda, B, ldb, bias, residual, ldr, out_bf16, out_f32, ldc, accumulate, M, N, K,
                                 (hipStream_t)stream);
}
