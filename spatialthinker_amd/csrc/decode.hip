// decode.hip — rollout-side kernels: KV-cache append, merge of the (shared prompt | own generated) attention
// partials, and token sampling (Gumbel-max == exact multinomial at temperature T, argmax at T = 0).
// Decode attention itself reuses the MFMA flash kernel through st_attn_fwd_ranges: for every prompt the
// G rollouts x (n_q/n_kv) query heads of a KV head form ONE query tile against the prompt's keys, so the
// prompt KV is streamed once per group instead of once per sample (attention.hip).
#include "common.h"

// kg/vg: (B, Rmax, width) bf16; row b gets the K/V slice of qkv row b at slot gen_len[b]; gen_len[b] += 1.
__global__ void kv_append_kernel(const uint16_t* __restrict__ qkv, int64_t ld, int col_k, int col_v, int width,
                                 uint16_t* __restrict__ kg, uint16_t* __restrict__ vg, int64_t gen_stride,
                                 const int32_t* __restrict__ gen_len, const int32_t* __restrict__ active, int B) {
    const int chunks = width >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * chunks) return;
    const int b = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    if (active && !active[b]) return;
    const int64_t dst = (int64_t)b * gen_stride + (int64_t)gen_len[b] * width + c;
    *reinterpret_cast<uint4*>(kg + dst) = *reinterpret_cast<const uint4*>(qkv + (int64_t)b * ld + col_k + c);
    *reinterpret_cast<uint4*>(vg + dst) = *reinterpret_cast<const uint4*>(qkv + (int64_t)b * ld + col_v + c);
}
__global__ void inc_len_kernel(int32_t* __restrict__ gen_len, const int32_t* __restrict__ active, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (!active || active[b])) gen_len[b] += 1;
}

// out[r, h, :] = sum_p w_p * part[p][r, h, :],  w_p = exp(lse[h][p*rows + r] - m) / sum_p exp(...)  (flash-decoding merge of
// P partial attentions over disjoint key sets).  parts: (P*rows, heads*D) bf16, lse: (heads, P*rows) fp32.
__global__ void attn_merge_kernel(const uint16_t* __restrict__ parts, int64_t ldp, const float* __restrict__ lse, int P,
                                  uint16_t* __restrict__ out, int64_t ldo, int rows, int heads, int D, int qgroup) {
    const int chunks = D >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * heads * chunks) return;
    const int c = (int)(idx % chunks) * 8;
    const int h = (int)((idx / chunks) % heads);
    const int r = (int)(idx / ((int64_t)chunks * heads));
    const float* lp = lse + (int64_t)h * P * rows + r;
    float m = -INFINITY;
    for (int p = 0; p < P; ++p) m = fmaxf(m, lp[(int64_t)p * rows]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    if (m > -INFINITY) {
        for (int p = 0; p < P; ++p) {
            const float l = lp[(int64_t)p * rows];
            if (l == -INFINITY) continue;
            const float w = __expf(l - m);
            den += w;
            float x[8];
            unpack8(*reinterpret_cast<const uint4*>(parts + ((int64_t)p * rows + r) * ldp + h * D + c), x);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * x[j];
        }
        const float inv = 1.f / den;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= inv;
    }
    // qgroup > 0: grouped row r = (sample r / g, group-local head r % g) -> final (B, n_q*D) layout, query head h*g + r%g
    uint16_t* op = qgroup > 0 ? out + (int64_t)(r / qgroup) * ldo + ((int64_t)h * qgroup + r % qgroup) * D + c : out + (int64_t)r * ldo + h * D + c;
    *reinterpret_cast<uint4*>(op) = pack8(acc);
}

// ---- sampling --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint64_t x) {          // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 16);
}
// argmax_i ( z_i / T + Gumbel_i ),  Gumbel_i = -log(-log(u_i)), u_i from a counter hash.  Each row is split over gridDim.y
// workgroups (partials -> scratch), reduced by sample_finish_kernel; with scratch == NULL one workgroup does the whole row.
__global__ __launch_bounds__(256) void sample_kernel(const uint16_t* __restrict__ logits, int64_t ldl, int V, float inv_temp,
                                                    int greedy, uint64_t seed, uint64_t step_host, const int64_t* __restrict__ step_dev,
                                                    const int32_t* __restrict__ forced, int32_t* __restrict__ out_ids,
                                                    float* __restrict__ scratch) {
    const int row = blockIdx.x;
    const uint16_t* x = logits + (int64_t)row * ldl;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    const uint64_t step = step_dev ? (uint64_t)step_dev[0] : step_host;
    const uint64_t key = (seed * 0x100000001B3ull) ^ (step << 32) ^ ((uint64_t)row * 0x9E3779B1ull);
    const int per = (V + gridDim.y - 1) / gridDim.y;
    const int v0 = blockIdx.y * per, v1 = min(V, v0 + per);
    for (int i = v0 + threadIdx.x; i < v1; i += 256) {
        float z = bf2f(x[i]) * inv_temp;
        if (!greedy) {
            const uint32_t r = mix32(key + (uint64_t)i * 0xD6E8FEB86659FD93ull);
            const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);          // (0,1)
            z -= __logf(-__logf(u));
        }
        if (z > best || (z == best && i < besti)) { best = z; besti = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    __shared__ float sb[4];
    __shared__ int si[4];
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (sb[k] > best || (sb[k] == best && si[k] < besti)) { best = sb[k]; besti = si[k]; }
        if (scratch) {
            scratch[((int64_t)row * gridDim.y + blockIdx.y) * 2] = best;
            scratch[((int64_t)row * gridDim.y + blockIdx.y) * 2 + 1] = __int_as_float(besti);
        } else {
            if (forced && forced[row] >= 0) besti = forced[row];
            out_ids[row] = besti;
        }
    }
}
__global__ void sample_finish_kernel(const float* __restrict__ scratch, int splits, const int32_t* __restrict__ forced,
                                     int32_t* __restrict__ out_ids, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int k = 0; k < splits; ++k) {
        const float b = scratch[((int64_t)row * splits + k) * 2];
        const int i = __float_as_int(scratch[((int64_t)row * splits + k) * 2 + 1]);
        if (b > best || (b == best && i < besti)) { best = b; besti = i; }
    }
    if (forced && forced[row] >= 0) besti = forced[row];
    out_ids[row] = besti;
}

extern "C" {

int st_kv_append(const st_bf16* qkv, int64_t ld, int col_k, int col_v, int width, st_bf16* kg, st_bf16* vg, int64_t gen_stride,
                 int32_t* gen_len, const int32_t* active, int B, int increment, st_stream_t stream) {
    if (!qkv || !kg || !vg || !gen_len || B <= 0 || width <= 0 || (width & 7) || (ld & 7) || (col_k & 7) || (col_v & 7)) return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(kv_append_kernel, dim3(st_cdiv((int64_t)B * (width / 8), 256)), dim3(256), 0, s, qkv, ld, col_k, col_v, width, kg,
                       vg, gen_stride, gen_len, active, B);
    if (increment) hipLaunchKernelGGL(inc_len_kernel, dim3(st_cdiv(B, 256)), dim3(256), 0, s, gen_len, active, B);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_attn_merge(const st_bf16* parts, int64_t ldp, const float* lse, int n_parts, st_bf16* out, int64_t ldo, int rows, int heads,
                  int D, int q_group, st_stream_t stream) {
    if (!parts || !lse || !out || n_parts <= 0 || rows <= 0 || heads <= 0 || (D & 7) || (ldp & 7) || (ldo & 7)) return ST_EINVAL;
    hipLaunchKernelGGL(attn_merge_kernel, dim3(st_cdiv((int64_t)rows * heads * (D / 8), 256)), dim3(256), 0, (hipStream_t)stream, parts,
                       ldp, lse, n_parts, out, ldo, rows, heads, D, q_group);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_sample(const st_bf16* logits, int64_t ldl, int B, int V, float temperature, int top_k, float top_p, uint64_t seed,
              uint64_t step, const int64_t* step_dev, const int32_t* forced, int32_t* out_ids, float* scratch, st_stream_t stream) {
    if (!logits || !out_ids || B <= 0 || V <= 0 || temperature < 0.f) return ST_EINVAL;
    if (top_k > 0 || top_p < 1.f) return -38;        // top-k / top-p filtering: not built yet (shipped configs use -1 / 1.0)
    const int greedy = temperature == 0.f;
    const int splits = scratch ? 16 : 1;                       // scratch: B * 16 * 2 floats
    hipLaunchKernelGGL(sample_kernel, dim3(B, splits), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, greedy ? 1.f : 1.f / temperature,
                       greedy, seed, step, step_dev, forced, out_ids, scratch);
    if (scratch)
        hipLaunchKernelGGL(sample_finish_kernel, dim3(st_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, scratch, splits, forced, out_ids, B);
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
