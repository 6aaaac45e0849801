// rowwise.hip — HBM-bound row kernels of the GRPO hot path for gfx950:
// fused log-prob forward/backward over the vocabulary, RMSNorm forward/backward.
// All reductions use 64-lane wave shuffles; global access is 16 B per lane.
#include "common.h"
#include <algorithm>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// ------------------------------------------------------------------------------------------
// log-prob forward: one 256-thread workgroup per token row, online (max, sum) in base 2.
// Algorithmic bytes: 2*V per row read once.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logprob_fwd_kernel(const uint16_t* __restrict__ logits, int64_t ldl,
                                                         const int64_t* __restrict__ labels, float inv_temp,
                                                         float* __restrict__ logp, float* __restrict__ lse_out, int V) {
    const int row = blockIdx.x;
    const uint16_t* x = logits + (int64_t)row * ldl;
    const float sc = inv_temp * LOG2E;                  // z2 = x * sc  (base-2 scaled logit)
    float m = -INFINITY, s = 0.f;
    const bool vec_ok = ((ldl & 7) == 0) && ((((uintptr_t)logits) & 15) == 0);
    const int V8 = vec_ok ? (V & ~7) : 0;
    for (int i = threadIdx.x * 8; i < V8; i += 256 * 8) {
        const uint4 r = *reinterpret_cast<const uint4*>(x + i);
        float f[8];
        unpack8(r, f);
        float vm = f[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) vm = fmaxf(vm, f[j]);
        vm *= sc;                                        // sc > 0
        if (vm > m) { s *= exp2f(m - vm); m = vm; }
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += exp2f(f[j] * sc - m);
        s += a;
    }
    for (int i = V8 + threadIdx.x; i < V; i += 256) {   // scalar tail / unaligned fallback
        const float z = bf2f(x[i]) * sc;
        if (z > m) { s *= exp2f(m - z); m = z; }
        s += exp2f(z - m);
    }
    // wave then workgroup combine of (m, s)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, mo);
        s = (m == -INFINITY ? 0.f : s * exp2f(m - mn)) + (mo == -INFINITY ? 0.f : so * exp2f(mo - mn));
        m = mn;
    }
    __shared__ float sm[4], ss[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = m; ss[w] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], S = ss[0];
        for (int k = 1; k < 4; ++k) {
            const float mn = fmaxf(M, sm[k]);
            S = (M == -INFINITY ? 0.f : S * exp2f(M - mn)) + (sm[k] == -INFINITY ? 0.f : ss[k] * exp2f(sm[k] - mn));
            M = mn;
        }
        const float lse = (M + log2f(S)) * LN2;          // natural-log logsumexp of z
        const int64_t lab = labels[row];
        float lp = 0.f;
        if (lab >= 0 && lab < V) lp = bf2f(x[lab]) * inv_temp - lse;
        logp[row] = lp;
        lse_out[row] = lse;
    }
}

// in-place backward: dlogits = g * inv_temp * (onehot - exp(z - lse)); rows with g == 0 are zero-filled
// without being read.  Algorithmic bytes: 2*V read + 2*V written per live row.
__global__ __launch_bounds__(256) void logprob_bwd_kernel(uint16_t* __restrict__ logits, int64_t ldl,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ lse, const float* __restrict__ g,
                                                         float inv_temp, int V) {
    const int row = blockIdx.x;
    uint16_t* x = logits + (int64_t)row * ldl;
    const float gr = g[row] * inv_temp;
    const float sc = inv_temp * LOG2E;
    const float l2 = lse[row] * LOG2E;
    const int64_t lab = labels[row];
    const bool vec_ok = ((ldl & 7) == 0) && ((((uintptr_t)logits) & 15) == 0);
    const int V8 = vec_ok ? (V & ~7) : 0;
    if (gr == 0.f) {
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x * 8; i < V8; i += 256 * 8) *reinterpret_cast<uint4*>(x + i) = z;
        for (int i = V8 + threadIdx.x; i < V; i += 256) x[i] = 0;
        return;
    }
    for (int i = threadIdx.x * 8; i < V8; i += 256 * 8) {
        const uint4 r = *reinterpret_cast<const uint4*>(x + i);
        float f[8];
        unpack8(r, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = exp2f(f[j] * sc - l2);
            f[j] = gr * (((int64_t)(i + j) == lab ? 1.f : 0.f) - p);
        }
        *reinterpret_cast<uint4*>(x + i) = pack8(f);
    }
    for (int i = V8 + threadIdx.x; i < V; i += 256) {
        const float p = exp2f(bf2f(x[i]) * sc - l2);
        x[i] = f2bf(gr * (((int64_t)i == lab ? 1.f : 0.f) - p));
    }
}

// ------------------------------------------------------------------------------------------
// RMSNorm forward: one wave per row, 4 rows per 256-thread workgroup.
// y = w * bf16(x * rstd)  (HF rounding points, modeling_qwen2_5_vl.py:74-79)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                         const uint16_t* __restrict__ w, float eps,
                                                         uint16_t* __restrict__ y, int64_t ldy,
                                                         float* __restrict__ rstd_out, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    uint16_t* yr = y + (int64_t)row * ldy;
    float ss = 0.f;
    for (int i = lane * 8; i < H; i += 512) {
        const uint4 r = *reinterpret_cast<const uint4*>(xr + i);
        float f[8];
        unpack8(r, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)H + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
    for (int i = lane * 8; i < H; i += 512) {
        const uint4 r = *reinterpret_cast<const uint4*>(xr + i);
        const uint4 wr = *reinterpret_cast<const uint4*>(w + i);
        float f[8], wf[8];
        unpack8(r, f);
        unpack8(wr, wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = wf[j] * bfround(f[j] * rstd);
        *reinterpret_cast<uint4*>(yr + i) = pack8(f);
    }
}

// Register-resident form of rmsnorm_fwd_kernel for H <= 512 * NCH (round 5): a lane issues ALL its 16-byte loads of the row first
// (NCH independent loads in flight per lane instead of one dependent load per loop trip) and keeps them for the second pass — the row
// is read once.  Same per-lane summation order as the loop form: bit-identical rstd and y.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_reg_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                             const uint16_t* __restrict__ w, float eps,
                                                             uint16_t* __restrict__ y, int64_t ldy,
                                                             float* __restrict__ rstd_out, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    uint16_t* yr = y + (int64_t)row * ldy;
    uint4 r[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane * 8 + j * 512;
        r[j] = i < H ? *reinterpret_cast<const uint4*>(xr + i) : make_uint4(0, 0, 0, 0);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (lane * 8 + j * 512 < H) {
            float f[8];
            unpack8(r[j], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)H + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane * 8 + j * 512;
        if (i < H) {
            float f[8], wf[8];
            unpack8(r[j], f);
            unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = wf[e] * bfround(f[e] * rstd);
            *reinterpret_cast<uint4*>(yr + i) = pack8(f);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Critic value head (dp_critic.py:52-125 reads `output.logits` of a token-classification model: score = nn.Linear(H, 1) on the final hidden
// state, in the model's bf16): v[t] = bf16(sum_h hn[t][h] * w[h] + b).  One wave per row.
// Backward: dhn[t][:] = bf16(dv[t] * w[:]); dw[:] += sum_t dv[t] * hn[t][:] (fp32, fixed row order per column: deterministic); db += sum dv.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void value_head_fwd_kernel(const uint16_t* __restrict__ hn, int64_t ldh, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ bias, float* __restrict__ out, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float acc = 0.f;
    for (int i = lane * 8; i < H; i += 512) {
        float f[8], wf[8];
        unpack8(*reinterpret_cast<const uint4*>(hn + (int64_t)row * ldh + i), f);
        unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += f[j] * wf[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = bfround(acc + (bias ? bf2f(bias[0]) : 0.f));
}

__global__ __launch_bounds__(256) void value_head_bwd_dx_kernel(const uint16_t* __restrict__ w, const float* __restrict__ dv, uint16_t* __restrict__ dhn,
                                                               int64_t lddh, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float d = dv[row];
    for (int i = lane * 8; i < H; i += 512) {
        float wf[8];
        unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] *= d;
        *reinterpret_cast<uint4*>(dhn + (int64_t)row * lddh + i) = pack8(wf);
    }
}

// one workgroup per 64 columns: thread (c = t & 63, lane group r = t >> 6) walks rows r, r + 4, ...; the four partial sums are added in a
// fixed order.  Workgroup 0 also sums dv for the bias.
__global__ __launch_bounds__(256) void value_head_bwd_dw_kernel(const uint16_t* __restrict__ hn, int64_t ldh, const float* __restrict__ dv,
                                                               float* __restrict__ dw, float* __restrict__ db, int T, int H) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), r0 = threadIdx.x >> 6;
    float acc = 0.f, accb = 0.f;
    if (c < H)
        for (int t = r0; t < T; t += 4) {
            const float d = dv[t];
            acc += d * bf2f(hn[(int64_t)t * ldh + c]);
            accb += d;
        }
    part[r0][threadIdx.x & 63] = acc;
    __syncthreads();
    if (r0 == 0 && c < H) dw[c] += (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    if (blockIdx.x == 0 && db) {                               // every column thread of a lane group holds the same bias partial
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[r0][0] = accb;
        __syncthreads();
        if (threadIdx.x == 0) db[0] += (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
    }
}

// RMSNorm forward feeding an MX-fp8 GEMM (config #5): the same row arithmetic, and the bf16 result is quantised in the same pass (e4m3 +
// e8m0 block scales in the layout of st_mxfp8_quantize: bit-identical to st_rmsnorm_fwd followed by st_mxfp8_quantize); the bf16 result
// itself is written only when the caller keeps it (y != nullptr: the weight-gradient GEMM of a pass with gradients).  H % 128 == 0:
// a lane's 8 consecutive columns never straddle an MX block, 4 lanes = one block, 16 lanes = one 128-column scale dword.
__global__ __launch_bounds__(256) void rmsnorm_mxfp8_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ w, float eps,
                                                           uint16_t* __restrict__ y, int64_t ldy, uint8_t* __restrict__ q, int64_t ldq,
                                                           uint32_t* __restrict__ scales, int64_t scale_rows, float* __restrict__ rstd_out,
                                                           int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    float ss = 0.f;
    for (int i = lane * 8; i < H; i += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + i), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)H + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
    for (int i0 = 0; i0 < H; i0 += 512) {                         // uniform trip count: the block reductions below need every lane
        const int i = i0 + lane * 8;
        const bool live = i < H;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
            float wf[8];
            unpack8(*reinterpret_cast<const uint4*>(xr + i), f);
            unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bfround(wf[j] * bfround(f[j] * rstd));
            if (y) *reinterpret_cast<uint4*>(y + (int64_t)row * ldy + i) = pack8(f);
        }
        uint32_t wq[2];
        const int e = mx_quant8(f, wq);
        if (live) *reinterpret_cast<uint2*>(q + (int64_t)row * ldq + i) = make_uint2(wq[0], wq[1]);
        const uint32_t sd = mx_scale_dword(e, lane);
        if (live && (lane & 15) == 0) scales[(int64_t)(i >> 7) * scale_rows + row] = sd;
    }
}

// Small-T variant (decode: one token per live sequence): one 256-thread workgroup per row so a 64-row call still puts 64
// workgroups in flight and the row is touched once per thread (row kept in registers between the two passes).
__global__ __launch_bounds__(256) void rmsnorm_fwd_row_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                             const uint16_t* __restrict__ w, float eps,
                                                             uint16_t* __restrict__ y, int64_t ldy,
                                                             float* __restrict__ rstd_out, int H) {
    __shared__ float part[4];
    const int row = blockIdx.x;
    const uint16_t* xr = x + (int64_t)row * ldx;
    uint16_t* yr = y + (int64_t)row * ldy;
    float f[2][8];
    float ss = 0.f;
    int cnt = 0;
    for (int i = threadIdx.x * 8; i < H && cnt < 2; i += 2048, ++cnt) {
        unpack8(*reinterpret_cast<const uint4*>(xr + i), f[cnt]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[cnt][j] * f[cnt][j];
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float rstd = rsqrtf((part[0] + part[1] + part[2] + part[3]) / (float)H + eps);
    if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
    cnt = 0;
    for (int i = threadIdx.x * 8; i < H && cnt < 2; i += 2048, ++cnt) {
        float wf[8];
        unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[cnt][j] = wf[j] * bfround(f[cnt][j] * rstd);
        *reinterpret_cast<uint4*>(yr + i) = pack8(f[cnt]);
    }
}

// RMSNorm backward, two kernels so that both are bandwidth-shaped:
//   dx: one wave per row (T waves in flight): pass 1 dot = sum(dy*w*xhat) by wave shuffles, pass 2 (row re-read from L1/L2)
//       dx = rstd*(dy*w - xhat*dot/H) (+ dres);
//   dw: column-slab reduction dw[c] += sum_t dy[t,c]*bf16(x[t,c]*rstd[t]) — 64 columns x 256 rows per workgroup, coalesced
//       128-byte row segments, one fp32 atomicAdd per column per workgroup.
__global__ __launch_bounds__(256) void rmsnorm_bwd_dx_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                            const uint16_t* __restrict__ w, const float* __restrict__ rstd,
                                                            const uint16_t* __restrict__ dy, int64_t lddy,
                                                            const uint16_t* __restrict__ dres, int64_t lddres,
                                                            uint16_t* __restrict__ dx, int64_t lddx, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float rs = rstd[row];
    const uint16_t* xr = x + (int64_t)row * ldx;
    const uint16_t* dyr = dy + (int64_t)row * lddy;
    float acc = 0.f;
    for (int i = lane * 8; i < H; i += 512) {
        float xf[8], df[8], wf[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + i), xf);
        unpack8(*reinterpret_cast<const uint4*>(dyr + i), df);
        unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(__fmul_rn(df[j], wf[j]), __fmul_rn(xf[j], rs), acc);      // explicit: the one-pass kernel forms the same bits
    }
    const float mdot = wave_sum(acc) / (float)H;
    for (int i = lane * 8; i < H; i += 512) {
        float xf[8], df[8], wf[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + i), xf);
        unpack8(*reinterpret_cast<const uint4*>(dyr + i), df);
        unpack8(*reinterpret_cast<const uint4*>(w + i), wf);
        if (dres) unpack8(*reinterpret_cast<const uint4*>(dres + (int64_t)row * lddres + i), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = __fmul_rn(rs, fmaf(-__fmul_rn(xf[j], rs), mdot, __fmul_rn(df[j], wf[j])));
            o[j] = dres ? (o[j] + d) : d;
        }
        *reinterpret_cast<uint4*>(dx + (int64_t)row * lddx + i) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void rmsnorm_bwd_dw_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ rstd, const uint16_t* __restrict__ dy,
                                                            int64_t lddy, float* __restrict__ dw, int T, int H, int rows_per_block) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(T, r0 + rows_per_block);
    float acc = 0.f;
    if (c < H)
        for (int r = r0 + ty; r < r1; r += 4)
            acc += bf2f(dy[(int64_t)r * lddy + c]) * bfround(bf2f(x[(int64_t)r * ldx + c]) * rstd[r]);
    part[ty][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ty == 0 && c < H) atomicAdd(dw + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

static int rmsnorm_bwd_two_kernels(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy, int64_t lddy,
                                   const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx, float* dw_accum, int T, int H, hipStream_t s);

// RMSNorm backward in ONE pass over x / dy (round 6): a wave owns a row at a time — the row's x and dy pieces stay in
// registers between the dot product and the dx store — and keeps this wave's share of dw (sum over its rows of dy * bf16(x * rstd)) in
// registers; the four waves of a workgroup add theirs in wave order through LDS and the workgroup leaves ONE fp32 partial row in
// `part` [gridDim.x][H], which rmsnorm_bwd_dw_finish_kernel adds to dw in block order: no atomics, the same bits on every run (the
// two-kernel form above adds its workgroups' column sums with fp32 atomics, i.e. in whatever order they retire).  NP = ceil(H / 512).
template <int NP>
__global__ __launch_bounds__(256, 2) void rmsnorm_bwd_fused_kernel(const uint16_t* __restrict__ x, int64_t ldx, const uint16_t* __restrict__ w,
                                                               const float* __restrict__ rstd, const uint16_t* __restrict__ dy, int64_t lddy,
                                                               const uint16_t* __restrict__ dres, int64_t lddres, uint16_t* __restrict__ dx,
                                                               int64_t lddx, float* __restrict__ part, int T, int H) {
    extern __shared__ float red[];                             // [3][H]: the partial sums of waves 1..3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 wv[NP];                                              // the weight stays packed (28 registers at H = 3584): two workgroups per CU fit
    float acc[NP][8];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = lane * 8 + k * 512;
        wv[k] = make_uint4(0, 0, 0, 0);
        if (i < H) wv[k] = *reinterpret_cast<const uint4*>(w + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    }
    for (int row = blockIdx.x * 4 + wave; row < T; row += gridDim.x * 4) {
        const float rs = rstd[row];
        const uint16_t* xr = x + (int64_t)row * ldx;
        const uint16_t* dyr = dy + (int64_t)row * lddy;
        uint4 xv[NP], dv[NP], rv[NP];                           // the whole row of all three inputs in flight at once
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = lane * 8 + k * 512;
            if (i < H) {
                xv[k] = *reinterpret_cast<const uint4*>(xr + i); dv[k] = *reinterpret_cast<const uint4*>(dyr + i);
                if (dres) rv[k] = *reinterpret_cast<const uint4*>(dres + (int64_t)row * lddres + i);
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = lane * 8 + k * 512;
            if (i < H) {
                float xf[8], df[8], wf[8];
                unpack8(xv[k], xf); unpack8(dv[k], df); unpack8(wv[k], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = __fmul_rn(xf[j], rs);
                    dot = fmaf(__fmul_rn(df[j], wf[j]), xh, dot);
                    acc[k][j] = fmaf(df[j], bfround(xh), acc[k][j]);
                }
            }
        }
        const float mdot = wave_sum(dot) / (float)H;
        // the packed row passes THROUGH an empty asm statement: without it the compiler keeps the first loop's unpacked fp32 copies of x and
        // dy alive for the second loop (112 registers at H = 3584: 368 bytes of scratch per lane at the 256-register bound of two workgroups per CU)
#pragma unroll
        for (int k = 0; k < NP; ++k)
            asm volatile("" : "+v"(xv[k].x), "+v"(xv[k].y), "+v"(xv[k].z), "+v"(xv[k].w), "+v"(dv[k].x), "+v"(dv[k].y), "+v"(dv[k].z), "+v"(dv[k].w));
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = lane * 8 + k * 512;
            if (i < H) {
                float xf[8], df[8], wf[8], o[8];
                unpack8(xv[k], xf); unpack8(dv[k], df); unpack8(wv[k], wf);
                if (dres) unpack8(rv[k], o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = __fmul_rn(rs, fmaf(-__fmul_rn(xf[j], rs), mdot, __fmul_rn(df[j], wf[j])));
                    o[j] = dres ? (o[j] + d) : d;
                }
                *reinterpret_cast<uint4*>(dx + (int64_t)row * lddx + i) = pack8(o);
            }
        }
    }
    if (!part) return;
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = lane * 8 + k * 512;
            if (i < H) {
                *reinterpret_cast<float4*>(red + (wave - 1) * H + i) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
                *reinterpret_cast<float4*>(red + (wave - 1) * H + i + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = lane * 8 + k * 512;
            if (i < H) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k][j] = ((acc[k][j] + red[i + j]) + red[H + i + j]) + red[2 * H + i + j];
                float* pp = part + (int64_t)blockIdx.x * H + i;
                *reinterpret_cast<float4*>(pp) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
                *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
            }
        }
    }
}

// dw[c] += sum over the partial rows in block order (64 columns per workgroup, four interleaved row walkers added in walker order)
__global__ __launch_bounds__(256) void rmsnorm_bwd_dw_finish_kernel(const float* __restrict__ part, int nb, float* __restrict__ dw, int H) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < H)
        for (int b = ty; b < nb; b += 4) acc += part[(int64_t)b * H + c];
    red[ty][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ty == 0 && c < H) dw[c] += ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

extern "C" {

int st_logprob_fwd(const st_bf16* logits, int64_t ldl, const int64_t* labels, float inv_temperature, float* logp,
                   float* lse, int T, int V, st_stream_t stream) {
    if (!logits || !labels || !logp || !lse || T < 0 || V <= 0 || ldl < V || !(inv_temperature > 0.f)) return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_LOGPROB, s, 2.0 * (double)T * (double)V);
    hipLaunchKernelGGL(logprob_fwd_kernel, dim3(T), dim3(256), 0, s, logits, ldl, labels, inv_temperature, logp, lse, V);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_logprob_bwd(st_bf16* logits, int64_t ldl, const int64_t* labels, const float* lse, const float* g,
                   float inv_temperature, int T, int V, st_stream_t stream) {
    if (!logits || !labels || !lse || !g || T < 0 || V <= 0 || ldl < V || !(inv_temperature > 0.f)) return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_LOGPROB, s, 4.0 * (double)T * (double)V);
    hipLaunchKernelGGL(logprob_bwd_kernel, dim3(T), dim3(256), 0, s, logits, ldl, labels, lse, g, inv_temperature, V);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_rmsnorm_fwd(const st_bf16* x, int64_t ldx, const st_bf16* w, float eps, st_bf16* y, int64_t ldy, float* rstd,
                   int T, int H, st_stream_t stream) {
    if (!x || !w || !y || T < 0 || H <= 0 || (H & 7) || (ldx & 7) || (ldy & 7)) return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_RMSNORM, s, 4.0 * (double)T * (double)H);
    if (T <= 1024 && H <= 4096)
        hipLaunchKernelGGL(rmsnorm_fwd_row_kernel, dim3(T), dim3(256), 0, s, x, ldx, w, eps, y, ldy, rstd, H);
    else if (H <= 2048)
        hipLaunchKernelGGL(rmsnorm_fwd_reg_kernel<4>, dim3(st_cdiv(T, 4)), dim3(256), 0, s, x, ldx, w, eps, y, ldy, rstd, T, H);
    else if (H <= 4096)
        hipLaunchKernelGGL(rmsnorm_fwd_reg_kernel<8>, dim3(st_cdiv(T, 4)), dim3(256), 0, s, x, ldx, w, eps, y, ldy, rstd, T, H);
    else
        hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(st_cdiv(T, 4)), dim3(256), 0, s, x, ldx, w, eps, y, ldy, rstd, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_value_head_fwd(const st_bf16* hn, int64_t ldh, const st_bf16* w, const st_bf16* bias, float* out, int T, int H, st_stream_t stream) {
    if (!hn || !w || !out || T < 0 || H <= 0 || (H & 7) || (ldh & 7) || ldh < H) return ST_EINVAL;
    if (T == 0) return 0;
    hipLaunchKernelGGL(value_head_fwd_kernel, dim3(st_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, hn, ldh, w, bias, out, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_value_head_bwd(const st_bf16* hn, int64_t ldh, const st_bf16* w, const float* dv, st_bf16* dhn, int64_t lddh, float* dw_accum,
                      float* db_accum, int T, int H, st_stream_t stream) {
    if (!hn || !w || !dv || !dhn || !dw_accum || T < 0 || H <= 0 || (H & 7) || (ldh & 7) || (lddh & 7) || ldh < H || lddh < H) return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(value_head_bwd_dx_kernel, dim3(st_cdiv(T, 4)), dim3(256), 0, s, w, dv, dhn, lddh, T, H);
    hipLaunchKernelGGL(value_head_bwd_dw_kernel, dim3(st_cdiv(H, 64)), dim3(256), 0, s, hn, ldh, dv, dw_accum, db_accum, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_rmsnorm_mxfp8(const st_bf16* x, int64_t ldx, const st_bf16* w, float eps, st_bf16* y, int64_t ldy, uint8_t* q, int64_t ldq,
                     uint32_t* scales, int64_t scale_rows, float* rstd, int T, int H, st_stream_t stream) {
    if (!x || !w || !q || !scales || T < 0 || H <= 0 || (H % 128) || (ldx & 7) || (y && (ldy & 7)) || (ldq & 7) || ldq < H || scale_rows < T ||
        (((uintptr_t)q) & 7))
        return ST_EINVAL;
    if (T == 0) return 0;
    hipLaunchKernelGGL(rmsnorm_mxfp8_kernel, dim3(st_cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, w, eps, y, ldy, q, ldq, scales,
                       scale_rows, rstd, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_rmsnorm_bwd(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy, int64_t lddy,
                   const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx, float* dw_accum, int T, int H,
                   st_stream_t stream) {
    if (!x || !w || !rstd || !dy || !dx || T < 0 || H <= 0 || (H & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7) ||
        (dres && (lddres & 7)))
        return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    return rmsnorm_bwd_two_kernels(x, ldx, w, rstd, dy, lddy, dres, lddres, dx, lddx, dw_accum, T, H, s);
}

#define RB_WGS_PER_CU 2       /* 8 waves per CU (<= 256 registers each), each with one whole row of x, dy and dres in flight; 512 partial rows of dw at 256 CUs */
/* one pass over x / dy, deterministic dw (see rmsnorm_bwd_fused_kernel).  workspace: st_rmsnorm_bwd_workspace_bytes(T, H) bytes of scratch
 * (contents irrelevant); H <= 4096 and 16-byte aligned rows, otherwise (or with workspace == NULL while dw_accum != NULL) ST_EINVAL. */
int64_t st_rmsnorm_bwd_workspace_bytes(int T, int H) {
    const int nb = (int)std::min<int64_t>(st_cdiv(T, 4), RB_WGS_PER_CU * (int64_t)st_num_cus());
    return (int64_t)std::max(nb, 1) * H * (int64_t)sizeof(float);
}
int st_rmsnorm_bwd_fused(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy, int64_t lddy,
                         const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx, float* dw_accum, void* workspace,
                         int64_t workspace_bytes, int T, int H, st_stream_t stream) {
    if (!x || !w || !rstd || !dy || !dx || T < 0 || H <= 0 || H > 4096 || (H & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7) || (dres && (lddres & 7)) ||
        (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)w | (uintptr_t)dres) & 15))
        return ST_EINVAL;
    if (dw_accum && (!workspace || workspace_bytes < st_rmsnorm_bwd_workspace_bytes(T, H) || (((uintptr_t)workspace) & 15))) return ST_EINVAL;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)std::min<int64_t>(st_cdiv(T, 4), RB_WGS_PER_CU * (int64_t)st_num_cus());
    float* part = dw_accum ? (float*)workspace : nullptr;
    const size_t lds = (size_t)3 * H * sizeof(float);
    const int np = st_cdiv(H, 512);
#define RB_GO(NP) hipLaunchKernelGGL((rmsnorm_bwd_fused_kernel<NP>), dim3(nb), dim3(256), lds, s, x, ldx, w, rstd, dy, lddy, dres, lddres, dx, lddx, part, T, H)
    switch (np) {
        case 1: RB_GO(1); break; case 2: RB_GO(2); break; case 3: RB_GO(3); break; case 4: RB_GO(4); break;
        case 5: RB_GO(5); break; case 6: RB_GO(6); break; case 7: RB_GO(7); break; default: RB_GO(8); break;
    }
#undef RB_GO
    if (dw_accum) hipLaunchKernelGGL(rmsnorm_bwd_dw_finish_kernel, dim3(st_cdiv(H, 64)), dim3(256), 0, s, part, nb, dw_accum, H);
    ST_CHECK_LAUNCH();
    return 0;
}

static int rmsnorm_bwd_two_kernels(const st_bf16* x, int64_t ldx, const st_bf16* w, const float* rstd, const st_bf16* dy, int64_t lddy,
                                   const st_bf16* dres, int64_t lddres, st_bf16* dx, int64_t lddx, float* dw_accum, int T, int H, hipStream_t s) {
    // dw first: dx may alias dy (in-place), and dw needs the original dy
    if (dw_accum)
        hipLaunchKernelGGL(rmsnorm_bwd_dw_kernel, dim3(st_cdiv(H, 64), st_cdiv(T, 256)), dim3(256), 0, s, x, ldx, rstd, dy, lddy, dw_accum,
                           T, H, 256);
    hipLaunchKernelGGL(rmsnorm_bwd_dx_kernel, dim3(st_cdiv(T, 4)), dim3(256), 0, s, x, ldx, w, rstd, dy, lddy, dres, lddres, dx, lddx, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
