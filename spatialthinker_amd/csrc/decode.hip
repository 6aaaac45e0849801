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
// ---- top-k / top-p (vLLM sampler semantics: top-k mask first, then top-p on the renormalised rest; temperature applied
// before both).  Logits are bf16, so there are only 65536 distinct values: the cut is found EXACTLY by a two-level histogram over
// an order-preserving 16-bit key (count for top-k, integer-scaled probability mass for top-p — integer LDS atomics keep the
// sums order-independent).  Result: one threshold key per row; sample_kernel skips tokens whose key is below it.  All tokens
// that tie with the threshold value are kept.
__device__ __forceinline__ uint32_t bf16_order_key(uint16_t b) { return (b & 0x8000u) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000u); }

__global__ __launch_bounds__(256) void sample_filter_kernel(const uint16_t* __restrict__ logits, int64_t ldl, int V, float inv_temp,
                                                           int top_k, float top_p, uint32_t* __restrict__ thr_out) {
    __shared__ uint32_t cnt[256];
    __shared__ unsigned long long mass[256];
    __shared__ uint32_t s_max, s_sel[2];
    __shared__ double s_target;
    const int row = blockIdx.x, tid = threadIdx.x;
    const uint16_t* x = logits + (int64_t)row * ldl;
    // row maximum (as key)
    uint32_t mk = 0;
    for (int i = tid; i < V; i += 256) {
        const uint16_t b = x[i];
        if ((b & 0x7fffu) > 0x7f80u) continue;                        // NaN never wins
        mk = max(mk, bf16_order_key(b));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mk = max(mk, (uint32_t)__shfl_xor((int)mk, o, 64));
    if (tid == 0) s_max = 0;
    __syncthreads();
    if ((tid & 63) == 0) atomicMax(&s_max, mk);
    __syncthreads();
    const uint32_t maxkey = s_max;
    const uint16_t maxbits = (maxkey & 0x8000u) ? (uint16_t)(maxkey & 0x7fffu) : (uint16_t)~maxkey;
    const float zmax = bf2f(maxbits) * inv_temp;
    auto weight = [&](uint16_t b) -> unsigned long long {              // exp(z - zmax) scaled to 2^44, >= 1 so no token vanishes
        const float e = __expf(bf2f(b) * inv_temp - zmax);
        return (unsigned long long)(e * 17592186044416.0f) + 1ull;
    };
    uint32_t key_k = 0;
    // ---------------- top-k: the k-th largest key
    if (top_k > 0 && top_k < V) {
        cnt[tid] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += 256) atomicAdd(&cnt[bf16_order_key(x[i]) >> 8], 1u);
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0; int hb = 255;
            for (; hb > 0; --hb) { if (cum + cnt[hb] >= (uint32_t)top_k) break; cum += cnt[hb]; }
            s_sel[0] = hb; s_sel[1] = (uint32_t)top_k - cum;           // rank still needed inside bucket hb
        }
        __syncthreads();
        const uint32_t hb = s_sel[0], need = s_sel[1];
        __syncthreads();
        cnt[tid] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += 256) { const uint32_t k = bf16_order_key(x[i]); if ((k >> 8) == hb) atomicAdd(&cnt[k & 255], 1u); }
        __syncthreads();
        if (tid == 0) {
            uint32_t cum = 0; int lb = 255;
            for (; lb > 0; --lb) { cum += cnt[lb]; if (cum >= need) break; }
            s_sel[0] = (hb << 8) | (uint32_t)lb;
        }
        __syncthreads();
        key_k = s_sel[0];
        __syncthreads();
    }
    uint32_t thr = key_k;
    // ---------------- top-p over the tokens that survived top-k
    if (top_p < 1.f) {
        mass[tid] = 0ull;
        __syncthreads();
        for (int i = tid; i < V; i += 256) {
            const uint16_t b = x[i];
            const uint32_t k = bf16_order_key(b);
            if (k >= key_k && (b & 0x7fffu) <= 0x7f80u) atomicAdd(&mass[k >> 8], weight(b));
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int h = 0; h < 256; ++h) tot += (double)mass[h];
            const double target = (double)top_p * tot;
            double cum = 0.0; int hb = 255;
            for (; hb > 0; --hb) { if (cum + (double)mass[hb] >= target) break; cum += (double)mass[hb]; }
            s_sel[0] = hb; s_target = target - cum;                    // mass still needed inside bucket hb
        }
        __syncthreads();
        const uint32_t hb = s_sel[0];
        __syncthreads();
        mass[tid] = 0ull;
        __syncthreads();
        for (int i = tid; i < V; i += 256) {
            const uint16_t b = x[i];
            const uint32_t k = bf16_order_key(b);
            if (k >= key_k && (k >> 8) == hb && (b & 0x7fffu) <= 0x7f80u) atomicAdd(&mass[k & 255], weight(b));
        }
        __syncthreads();
        if (tid == 0) {
            double cum = 0.0; int lb = 255;
            for (; lb > 0; --lb) { cum += (double)mass[lb]; if (cum >= s_target) break; }
            s_sel[0] = (hb << 8) | (uint32_t)lb;
        }
        __syncthreads();
        thr = max(thr, s_sel[0]);
    }
    if (tid == 0) thr_out[row] = thr;
}

// argmax_i ( z_i / T + Gumbel_i ),  Gumbel_i = -log(-log(u_i)), u_i from a counter hash.  Each row is split over gridDim.y
// workgroups (partials -> scratch), reduced by sample_finish_kernel; with scratch == NULL one workgroup does the whole row.
__global__ __launch_bounds__(256) void sample_kernel(const uint16_t* __restrict__ logits, int64_t ldl, int V, float inv_temp,
                                                    int greedy, uint64_t seed, uint64_t step_host, const int64_t* __restrict__ step_dev,
                                                    const int32_t* __restrict__ forced, int32_t* __restrict__ out_ids,
                                                    float* __restrict__ scratch, const int32_t* __restrict__ row_ids,
                                                    const uint32_t* __restrict__ thr, const int32_t* __restrict__ row_steps,
                                                    float* __restrict__ lse_part) {
    const int row = blockIdx.x;
    const uint16_t* x = logits + (int64_t)row * ldl;
    const uint32_t min_key = thr ? thr[row] : 0u;                      // top-k / top-p cut (0 = keep everything)
    float best = -INFINITY;
    int besti = 0x7fffffff;
    // lse_part (optional): this split's (max, sum exp(z - max)) of the UNFILTERED z = logit / T — st_decode_step turns the 16 partials
    // into the log-probability of the token it records (the rollout's own old-policy log-probs, same pass over the logits)
    float lm = -INFINITY, ls = 0.f;
    // response index of this row's token: per-row (rows of one launch may be at different positions once survivors of several
    // waves are decoded together), else a device counter, else the host value
    const uint64_t step = row_steps ? (uint64_t)row_steps[row] : (step_dev ? (uint64_t)step_dev[0] : step_host);
    // the stream is keyed by the sample's identity, not by its row in this launch: compacting the decode batch (dropping
    // finished samples) does not change what the survivors sample
    const uint64_t rid = row_ids ? (uint64_t)row_ids[row] : (uint64_t)row;
    const uint64_t key = (seed * 0x100000001B3ull) ^ (step << 32) ^ (rid * 0x9E3779B1ull);
    const int per = (V + gridDim.y - 1) / gridDim.y;
    const int v0 = blockIdx.y * per, v1 = min(V, v0 + per);
    for (int i = v0 + threadIdx.x; i < v1; i += 256) {
        if (lse_part) {
            const float zz = bf2f(x[i]) * inv_temp;
            if (zz > lm) { ls = ls * __expf(lm - zz) + 1.f; lm = zz; } else ls += __expf(zz - lm);
        }
        if (min_key && bf16_order_key(x[i]) < min_key) continue;
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
    __shared__ float sm[4], ss[4];
    if (lse_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(lm, o, 64), os = __shfl_xor(ls, o, 64);
            const float nm = fmaxf(lm, om);
            ls = (lm == -INFINITY ? 0.f : ls * __expf(lm - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
            lm = nm;
        }
        if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = lm; ss[threadIdx.x >> 6] = ls; }
    }
    if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (sb[k] > best || (sb[k] == best && si[k] < besti)) { best = sb[k]; besti = si[k]; }
        if (lse_part) {
            float m = sm[0], sacc = ss[0];
            for (int k = 1; k < 4; ++k) {
                const float nm = fmaxf(m, sm[k]);
                sacc = (m == -INFINITY ? 0.f : sacc * __expf(m - nm)) + (sm[k] == -INFINITY ? 0.f : ss[k] * __expf(sm[k] - nm));
                m = nm;
            }
            lse_part[((int64_t)row * gridDim.y + blockIdx.y) * 2] = m;
            lse_part[((int64_t)row * gridDim.y + blockIdx.y) * 2 + 1] = sacc;
        }
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

// ---------------------------------------------------------------------------------------------------------------------
// Fused decode epilogues.  The decode GEMMs leave fp32 split-K slabs [split][M][N]; instead of a finish kernel followed by
// separate RMSNorm / RoPE / KV-append launches (each ~5 us of mostly launch latency per layer), one workgroup per row sums
// the slabs in the fixed split order and applies everything that follows, with the same bf16 rounding points as the
// unfused kernels (st_gemm_nt_skinny finish -> st_rmsnorm_fwd; finish -> st_rope_apply -> st_kv_append).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_sum8(const float* __restrict__ slabs, int splits, int64_t slab_stride, int64_t off, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    // loads of four slabs are issued together (one latency instead of four); the additions keep the split order
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(slabs + (s + u) * slab_stride + off);
            b[u] = *reinterpret_cast<const float4*>(slabs + (s + u) * slab_stride + off + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w; v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
        }
    }
    for (; s < splits; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(slabs + s * slab_stride + off);
        const float4 b = *reinterpret_cast<const float4*>(slabs + s * slab_stride + off + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
}

// x_out = bf16(sum slabs + residual);  h_out = rmsnorm(x_out) * norm_w (skipped when norm_w == NULL).  N <= 4096.
// One workgroup per row; every global load of the row (all S slabs of both column chunks, residual, norm weight) is issued
// before the first use, so the kernel costs one memory round trip instead of a chain of dependent ones (slab lines come
// from other XCDs' write-backs: ~2 us each).
template <int S>
__global__ __launch_bounds__(256) void decode_finish_norm_kernel(const float* __restrict__ slabs, const uint16_t* __restrict__ res,
                                                                int64_t ldr, uint16_t* __restrict__ x_out, int64_t ldx,
                                                                const uint16_t* __restrict__ norm_w, float eps,
                                                                uint16_t* __restrict__ h_out, int64_t ldh, int M, int N) {
    __shared__ float part[4];
    const int row = blockIdx.x;
    const int64_t slab_stride = (int64_t)M * N;
    const int i0 = threadIdx.x * 8, i1 = i0 + 2048;
    const bool ok[2] = {i0 < N, i1 < N};
    const int col[2] = {i0, i1};
    float4 sa[2][S], sb[2][S];
    uint4 rr[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, ww[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!ok[c]) continue;
        const float* sp = slabs + (int64_t)row * N + col[c];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sa[c][s] = *reinterpret_cast<const float4*>(sp + s * slab_stride);
            sb[c][s] = *reinterpret_cast<const float4*>(sp + s * slab_stride + 4);
        }
        if (res) rr[c] = *reinterpret_cast<const uint4*>(res + (int64_t)row * ldr + col[c]);
        if (norm_w) ww[c] = *reinterpret_cast<const uint4*>(norm_w + col[c]);
    }
    float f[2][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!ok[c]) continue;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) {                       // fixed split order: same sums as gemm_skinny_finish
            v[0] += sa[c][s].x; v[1] += sa[c][s].y; v[2] += sa[c][s].z; v[3] += sa[c][s].w;
            v[4] += sb[c][s].x; v[5] += sb[c][s].y; v[6] += sb[c][s].z; v[7] += sb[c][s].w;
        }
        if (res) {
            float r[8];
            unpack8(rr[c], r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
        const uint4 packed = pack8(v);
        *reinterpret_cast<uint4*>(x_out + (int64_t)row * ldx + col[c]) = packed;
        unpack8(packed, f[c]);                              // the norm sees the bf16-rounded stream value
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[c][j] * f[c][j];
    }
    if (!norm_w) return;
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf((part[0] + part[1] + part[2] + part[3]) / (float)N + eps);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!ok[c]) continue;
        float wf[8];
        unpack8(ww[c], wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[c][j] = wf[j] * bfround(f[c][j] * rstd);
        *reinterpret_cast<uint4*>(h_out + (int64_t)row * ldh + col[c]) = pack8(f[c]);
    }
}

// qkv row = bf16(sum slabs + bias); RoPE on the q and k heads; q -> q_out[b], k/v -> kg/vg[b, gen_len[b]].
__global__ __launch_bounds__(256) void decode_finish_qkv_kernel(const float* __restrict__ slabs, int splits,
                                                               const uint16_t* __restrict__ bias, const float* __restrict__ cosb,
                                                               const float* __restrict__ sinb, uint16_t* __restrict__ q_out,
                                                               int64_t ldq, uint16_t* __restrict__ kg, uint16_t* __restrict__ vg,
                                                               int64_t gen_stride, const int32_t* __restrict__ gen_len,
                                                               const int32_t* __restrict__ row_map, int B,
                                                               int M, int n_q, int n_kv, int D) {
    const int b = blockIdx.x;
    const int N = (n_q + 2 * n_kv) * D;
    const int64_t slab_stride = (int64_t)M * N;
    const int half = D >> 1, chunks = half >> 3;
    const int width = n_kv * D;
    const int64_t cache_row = (int64_t)(row_map ? row_map[b] : b) * gen_stride + (int64_t)gen_len[b] * width;   // row b holds sample row_map[b]
    // rotary heads: item = (head, chunk of 8 inside the first half); partner chunk at +half
    for (int it = threadIdx.x; it < (n_q + n_kv) * chunks; it += 256) {
        const int hd = it / chunks, ch = it % chunks;
        const int col = hd * D + ch * 8;
        float a[8], bb[8], ba[8], bbv[8];
        slab_sum8(slabs, splits, slab_stride, (int64_t)b * N + col, a);
        slab_sum8(slabs, splits, slab_stride, (int64_t)b * N + col + half, bb);
        if (bias) {
            unpack8(*reinterpret_cast<const uint4*>(bias + col), ba);
            unpack8(*reinterpret_cast<const uint4*>(bias + col + half), bbv);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] += ba[j]; bb[j] += bbv[j]; }
        }
        float c[8], sn[8], o1[8], o2[8];
        const float* cp = cosb + (int64_t)b * half + ch * 8;
        const float* sp = sinb + (int64_t)b * half + ch * 8;
        *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(cp);
        *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(cp + 4);
        *reinterpret_cast<float4*>(sn) = *reinterpret_cast<const float4*>(sp);
        *reinterpret_cast<float4*>(sn + 4) = *reinterpret_cast<const float4*>(sp + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x1 = bfround(a[j]), x2 = bfround(bb[j]);             // the projection output is bf16 before the rotation
            o1[j] = x1 * c[j] - x2 * sn[j];
            o2[j] = x2 * c[j] + x1 * sn[j];
        }
        uint16_t* dst = hd < n_q ? q_out + (int64_t)b * ldq + col : kg + cache_row + (col - n_q * D);
        *reinterpret_cast<uint4*>(dst) = pack8(o1);
        *reinterpret_cast<uint4*>(dst + half) = pack8(o2);
    }
    // value heads: plain finish into the cache
    for (int it = threadIdx.x; it < width / 8; it += 256) {
        const int col = (n_q + n_kv) * D + it * 8;
        float v[8], bv[8];
        slab_sum8(slabs, splits, slab_stride, (int64_t)b * N + col, v);
        if (bias) {
            unpack8(*reinterpret_cast<const uint4*>(bias + col), bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bv[j];
        }
        *reinterpret_cast<uint4*>(vg + cache_row + it * 8) = pack8(v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// decode_step_kernel: everything between two decode forwards in ONE launch, one workgroup per live row (the captured iteration used
// to spend ~25 one-element-per-row torch kernels on it): finish the split sampler (argmax over the 16 partials, forced EOS), record
// the token, update the live flag / response index / cache slot / generated-key range ends, build the row's M-RoPE cos/sin for
// the position of this token and advance the position, gather the token's embedding row.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_step_kernel(const float* __restrict__ partials, int splits, const int32_t* __restrict__ forced_len,
                                                         int forced_token, const int64_t* __restrict__ eos_ids, int n_eos, int ignore_eos,
                                                         int32_t* __restrict__ gen_len, int32_t* __restrict__ active, int64_t* __restrict__ out_tokens,
                                                         int R, int32_t* __restrict__ tok_out, int32_t* __restrict__ slot_out,
                                                         const int32_t* __restrict__ k_base, const int32_t* __restrict__ kb_gen,
                                                         int32_t* __restrict__ ke_gen, int n_chunks, int chunk_keys, int32_t* __restrict__ pos,
                                                         const float* __restrict__ inv_freq, int half, int s0, int s1, float* __restrict__ cosb,
                                                         float* __restrict__ sinb, const uint16_t* __restrict__ embed, int64_t ld_embed,
                                                         uint16_t* __restrict__ x_out, int64_t ldx, int H, int B,
                                                         const float* __restrict__ lse_part, const uint16_t* __restrict__ logits, int64_t ldl,
                                                         float inv_temp, float* __restrict__ logp_out) {
    __shared__ int s_tok, s_pos[3];
    const int row = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        float best = -INFINITY;
        int besti = 0x7fffffff;
        for (int k = 0; k < splits; ++k) {                   // same order and tie-break as sample_finish_kernel
            const float b = partials[((int64_t)row * splits + k) * 2];
            const int i = __float_as_int(partials[((int64_t)row * splits + k) * 2 + 1]);
            if (b > best || (b == best && i < besti)) { best = b; besti = i; }
        }
        const int j = gen_len[row];                          // response index of this token
        if (forced_len && forced_len[row] == j + 1) besti = forced_token;
        const bool live = active[row] != 0;
        const int col = j < R - 1 ? j : R - 1;               // finished rows at the cap rewrite their last slot
        if (live) out_tokens[(int64_t)row * R + col] = (int64_t)besti;
        if (logp_out && live) {                              // log softmax(logits / T)[token]: 16 (max, sum) partials in fixed order
            float m = -INFINITY, sacc = 0.f;
            for (int k = 0; k < splits; ++k) {
                const float pm = lse_part[((int64_t)row * splits + k) * 2], ps = lse_part[((int64_t)row * splits + k) * 2 + 1];
                const float nm = fmaxf(m, pm);
                sacc = (m == -INFINITY ? 0.f : sacc * __expf(m - nm)) + (pm == -INFINITY ? 0.f : ps * __expf(pm - nm));
                m = nm;
            }
            logp_out[(int64_t)row * R + col] = bf2f(logits[(int64_t)row * ldl + besti]) * inv_temp - (m + __logf(sacc));
        }
        bool stop = j + 1 >= R;                              // the length cap ends a sample like an EOS
        if (!ignore_eos)
            for (int e = 0; e < n_eos; ++e) stop = stop || ((int64_t)besti == eos_ids[e]);
        const bool live_next = live && !stop;
        active[row] = live_next ? 1 : 0;
        tok_out[row] = besti;
        slot_out[row] = col;
        gen_len[row] = live ? j + 1 : j;                     // a finished row keeps its response length (round 5: it used to keep counting)
        // A finished row stays in the phase's GEMM tiles until the phase is re-batched, but its output is never read again: its generated-key
        // ranges are EMPTY from here on, so the attention launch streams no K/V for it (round 5; in the bench's 512-row phase the finished
        // rows were 15-17 % of the generated-K/V traffic, 27-69 % in the later phases).  Its prompt partial keeps the merged row finite.
        const int kend = k_base[row] + col + 1;              // one past the cache row this token's K/V will occupy
        for (int c = 0; c < n_chunks; ++c) {
            const int kb = kb_gen[c * B + row];
            int ke = kb + chunk_keys < kend ? kb + chunk_keys : kend;
            ke_gen[c * B + row] = (live && ke > kb) ? ke : kb;   // (the forward of a row's LAST token still attends: tests tap its logits)
        }
        s_tok = besti;
    }
    if (tid < 3) s_pos[tid] = pos[tid * B + row];
    __syncthreads();
    if (tid < half) {
        const int sec = tid < s0 ? 0 : (tid < s0 + s1 ? 1 : 2);
        const float ang = (float)s_pos[sec] * inv_freq[tid];  // as mrope_table_kernel
        float sn, cs;
        sincosf(ang, &sn, &cs);
        cosb[(int64_t)row * half + tid] = cs;
        sinb[(int64_t)row * half + tid] = sn;
    }
    if (tid < 3) pos[tid * B + row] = s_pos[tid] + 1;
    const uint16_t* src = embed + (int64_t)s_tok * ld_embed;
    for (int c = tid; c < (H >> 3); c += 256)
        *reinterpret_cast<uint4*>(x_out + (int64_t)row * ldx + c * 8) = *reinterpret_cast<const uint4*>(src + c * 8);
}

extern "C" {

int st_decode_finish_norm(const float* slabs, int splits, const st_bf16* residual, int64_t ldr, st_bf16* x_out, int64_t ldx,
                          const st_bf16* norm_w, float eps, st_bf16* h_out, int64_t ldh, int M, int N, st_stream_t stream) {
    if (!slabs || splits <= 0 || !x_out || M <= 0 || N <= 0 || N > 4096 || (N & 7) || (ldx & 7) || (residual && (ldr & 7)) ||
        (norm_w && (!h_out || (ldh & 7))))
        return ST_EINVAL;
#define ST_FN(S) hipLaunchKernelGGL(decode_finish_norm_kernel<S>, dim3(M), dim3(256), 0, (hipStream_t)stream, slabs, residual, ldr, x_out, ldx, norm_w, eps, h_out, ldh, M, N)
    switch (splits) {
        case 1: ST_FN(1); break; case 2: ST_FN(2); break; case 3: ST_FN(3); break; case 4: ST_FN(4); break;
        case 5: ST_FN(5); break; case 6: ST_FN(6); break; case 7: ST_FN(7); break; case 8: ST_FN(8); break;
        case 9: ST_FN(9); break; case 10: ST_FN(10); break; case 11: ST_FN(11); break; case 12: ST_FN(12); break;
        default: return ST_EINVAL;                          // decode_plan never splits deeper than 12
    }
#undef ST_FN
    ST_CHECK_LAUNCH();
    return 0;
}

int st_decode_finish_qkv(const float* slabs, int splits, const st_bf16* bias, const float* cos_tab, const float* sin_tab,
                         st_bf16* q_out, int64_t ldq, st_bf16* kg, st_bf16* vg, int64_t gen_stride, const int32_t* gen_len,
                         const int32_t* row_map, int B, int M, int n_q, int n_kv, int D, st_stream_t stream) {
    if (!slabs || splits <= 0 || !cos_tab || !sin_tab || !q_out || !kg || !vg || !gen_len || B <= 0 || B > M || n_q <= 0 || n_kv <= 0 ||
        (D & 15) || (ldq & 7))
        return ST_EINVAL;
    hipLaunchKernelGGL(decode_finish_qkv_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, slabs, splits, bias, cos_tab, sin_tab, q_out,
                       ldq, kg, vg, gen_stride, gen_len, row_map, B, M, n_q, n_kv, D);
    ST_CHECK_LAUNCH();
    return 0;
}

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
              uint64_t step, const int64_t* step_dev, const int32_t* forced, const int32_t* row_ids, const int32_t* row_steps,
              int32_t* out_ids, float* scratch, st_stream_t stream) {
    if (!logits || !out_ids || B <= 0 || V <= 0 || temperature < 0.f) return ST_EINVAL;
    const int greedy = temperature == 0.f;
    const int splits = scratch ? 16 : 1;                       // scratch: B * 16 * 2 floats (+ B threshold words when filtering)
    const bool filter = !greedy && ((top_k > 0 && top_k < V) || top_p < 1.f);
    if (filter && (!scratch || top_p <= 0.f)) return ST_EINVAL;
    uint32_t* thr = filter ? reinterpret_cast<uint32_t*>(scratch + (int64_t)B * 32) : nullptr;
    if (filter)
        hipLaunchKernelGGL(sample_filter_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, 1.f / temperature, top_k, top_p, thr);
    hipLaunchKernelGGL(sample_kernel, dim3(B, splits), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, greedy ? 1.f : 1.f / temperature,
                       greedy, seed, step, step_dev, forced, out_ids, scratch, row_ids, thr, row_steps, (float*)nullptr);
    if (scratch)
        hipLaunchKernelGGL(sample_finish_kernel, dim3(st_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, scratch, splits, forced, out_ids, B);
    ST_CHECK_LAUNCH();
    return 0;
}

/* The sampler without its last stage: partial (value, index) pairs [row][16][2] stay in `scratch`; st_decode_step finishes them. */
int st_sample_partials(const st_bf16* logits, int64_t ldl, int B, int V, float temperature, int top_k, float top_p, uint64_t seed,
                       uint64_t step, const int64_t* step_dev, const int32_t* row_ids, const int32_t* row_steps, float* scratch,
                       float* lse_partials, st_stream_t stream) {
    if (!logits || !scratch || B <= 0 || V <= 0 || temperature < 0.f) return ST_EINVAL;
    const int greedy = temperature == 0.f;
    const bool filter = !greedy && ((top_k > 0 && top_k < V) || top_p < 1.f);
    if (filter && top_p <= 0.f) return ST_EINVAL;
    uint32_t* thr = filter ? reinterpret_cast<uint32_t*>(scratch + (int64_t)B * 32) : nullptr;
    if (filter)
        hipLaunchKernelGGL(sample_filter_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, 1.f / temperature, top_k, top_p, thr);
    hipLaunchKernelGGL(sample_kernel, dim3(B, 16), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, greedy ? 1.f : 1.f / temperature, greedy,
                       seed, step, step_dev, (const int32_t*)nullptr, (int32_t*)nullptr, scratch, row_ids, thr, row_steps, lse_partials);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_decode_step(const float* sample_scratch, const int32_t* forced_len, int32_t forced_token, const int64_t* eos_ids, int n_eos,
                   int ignore_eos, int32_t* gen_len, int32_t* active, int64_t* out_tokens, int R, int32_t* tok_out, int32_t* slot_out,
                   const int32_t* k_base, const int32_t* kb_gen, int32_t* ke_gen, int n_chunks, int chunk_keys, int32_t* pos,
                   const float* inv_freq, int D, int s0, int s1, int s2, float* cos_out, float* sin_out, const st_bf16* embed,
                   int64_t ld_embed, st_bf16* x_out, int64_t ldx, int H, int B, const float* lse_partials, const st_bf16* logits, int64_t ldl,
                   float temperature, float* logp_out, st_stream_t stream) {
    if (!sample_scratch || !gen_len || !active || !out_tokens || !tok_out || !slot_out || !k_base || !kb_gen || !ke_gen || !pos || !inv_freq ||
        !cos_out || !sin_out || !embed || !x_out || B <= 0 || R <= 0 || n_chunks < 0 || D <= 0 || (D & 1) || D / 2 > 256 || s0 + s1 + s2 != D / 2 ||
        H <= 0 || (H & 7) || (ld_embed & 7) || (ldx & 7) || (n_eos > 0 && !eos_ids) || (logp_out && (!lse_partials || !logits || temperature < 0.f)))
        return ST_EINVAL;
    hipLaunchKernelGGL(decode_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sample_scratch, 16, forced_len, (int)forced_token, eos_ids,
                       n_eos, ignore_eos, gen_len, active, out_tokens, R, tok_out, slot_out, k_base, kb_gen, ke_gen, n_chunks, chunk_keys, pos,
                       inv_freq, D / 2, s0, s1, cos_out, sin_out, embed, ld_embed, x_out, ldx, H, B, lse_partials, logits, ldl,
                       temperature > 0.f ? 1.f / temperature : 1.f, logp_out);
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
