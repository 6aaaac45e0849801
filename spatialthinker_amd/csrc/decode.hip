// decode.hip — rollout-side kernels (decode attention against the KV cache, token sampling).
#include "common.h"

extern "C" {
int st_decode_attn(const st_bf16* q, int64_t ldq, const st_bf16* kp, const st_bf16* vp, const int64_t* prompt_off,
                   const int32_t* prompt_len, const st_bf16* kg, const st_bf16* vg, int64_t gen_stride,
                   const int32_t* gen_len, int B, int n_q, int n_kv, int D, float scale, st_bf16* out, int64_t ldo,
                   st_stream_t stream) { return -38; }
int st_sample(const st_bf16* logits, int64_t ldl, int B, int V, float temperature, int top_k, float top_p,
              uint64_t seed, uint64_t step, int32_t* out_ids, float* scratch, st_stream_t stream) { return -38; }
}
