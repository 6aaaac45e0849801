// elementwise.hip — HBM-bound elementwise / gather / small-reduction kernels (gfx950):
// rotary tables + apply, SwiGLU, GELU, fused AdamW-Kahan, grad-norm, packed-layout gather/scatter,
// GRPO loss + advantage, transposes and column sums.  16-byte accesses per lane throughout.
#include "common.h"

// ------------------------------------------------------------------------------ rotary
// M-RoPE table: cos/sin (T, D/2) fp32; band i takes its position from row sec(i) of pos (3,T).
__global__ void mrope_table_kernel(const int32_t* __restrict__ pos, const float* __restrict__ inv_freq, int T, int half,
                                   int s0, int s1, float* __restrict__ cosb, float* __restrict__ sinb) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * half) return;
    const int t = (int)(idx / half), i = (int)(idx % half);
    const int sec = i < s0 ? 0 : (i < s0 + s1 ? 1 : 2);
    const float ang = (float)pos[(int64_t)sec * T + t] * inv_freq[i];   // fp32 product, as HF :531-535
    float s, c;
    sincosf(ang, &s, &c);
    cosb[idx] = c;
    sinb[idx] = s;
}

// x (T, ld): the first n_rot heads of head_dim D in every row are rotated in place:
//   y[i] = x[i]*c - x[i+h]*s ; y[i+h] = x[i+h]*c + x[i]*s   (rotate_half, HF :153-157); inverse: s -> -s
__global__ void rope_apply_kernel(uint16_t* __restrict__ x, int64_t ld, const float* __restrict__ cosb,
                                  const float* __restrict__ sinb, int T, int n_rot, int D, float sgn) {
    const int half = D >> 1, chunks = half >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)T * n_rot * chunks;
    if (idx >= total) return;
    const int ch = (int)(idx % chunks);
    const int hd = (int)((idx / chunks) % n_rot);
    const int t = (int)(idx / ((int64_t)chunks * n_rot));
    uint16_t* p = x + (int64_t)t * ld + (int64_t)hd * D + ch * 8;
    float a[8], b[8], c[8], s[8];
    unpack8(*reinterpret_cast<const uint4*>(p), a);
    unpack8(*reinterpret_cast<const uint4*>(p + half), b);
    const float* cp = cosb + (int64_t)t * half + ch * 8;
    const float* sp = sinb + (int64_t)t * half + ch * 8;
    *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(cp);
    *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(cp + 4);
    *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(sp);
    *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(sp + 4);
    float o1[8], o2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sj = s[j] * sgn;
        o1[j] = a[j] * c[j] - b[j] * sj;
        o2[j] = b[j] * c[j] + a[j] * sj;
    }
    *reinterpret_cast<uint4*>(p) = pack8(o1);
    *reinterpret_cast<uint4*>(p + half) = pack8(o2);
}

// ------------------------------------------------------------------------------ SwiGLU / GELU

__global__ void swiglu_fwd_kernel(const uint16_t* __restrict__ gu, int64_t ldgu, uint16_t* __restrict__ out, int64_t ldo,
                                  int T, int I) {
    const int chunks = I >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * chunks) return;
    const int t = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    float g[8], u[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + c), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + I + c), u);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bfround(g[j] * sigmoidf_(g[j])) * u[j];   // HF rounds act_fn(gate) first
    *reinterpret_cast<uint4*>(out + (int64_t)t * ldo + c) = pack8(o);
}

// SwiGLU forward feeding an MX-fp8 GEMM (config #5): out (optional) as swiglu_fwd_kernel, plus its MX-fp8 quantisation in the same pass
// (bit-identical to st_swiglu_fwd followed by st_mxfp8_quantize).  One wave per (row, 512 columns), a lane = 8 consecutive columns.
__global__ __launch_bounds__(256) void swiglu_mxfp8_kernel(const uint16_t* __restrict__ gu, int64_t ldgu, uint16_t* __restrict__ out, int64_t ldo,
                                                          uint8_t* __restrict__ q, int64_t ldq, uint32_t* __restrict__ scales, int64_t scale_rows,
                                                          int T, int I) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = (I + 511) / 512;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= (int64_t)T * chunks) return;
    const int t = (int)(item / chunks), c = (int)(item % chunks) * 512 + lane * 8;
    const bool live = c < I;                                  // I is a multiple of 128: a lane is all-in or all-out
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
        float g[8], u[8];
        unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + c), g);
        unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + I + c), u);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bfround(bfround(g[j] * sigmoidf_(g[j])) * u[j]);
        if (out) *reinterpret_cast<uint4*>(out + (int64_t)t * ldo + c) = pack8(o);
    }
    uint32_t w[2];
    const int e = mx_quant8(o, w);
    if (live) *reinterpret_cast<uint2*>(q + (int64_t)t * ldq + c) = make_uint2(w[0], w[1]);
    const uint32_t sd = mx_scale_dword(e, lane);
    if (live && (lane & 15) == 0) scales[(int64_t)(c >> 7) * scale_rows + t] = sd;
}

// m_out (optional): the forward's result silu(gate) * up, recomputed here from the gate | up values this kernel reads anyway — the
// recompute_light backward needs it for the down projection's weight gradient and used to run st_swiglu_fwd (one more pass over gu) for it
__global__ void swiglu_bwd_kernel(const uint16_t* __restrict__ gu, int64_t ldgu, const uint16_t* __restrict__ dout,
                                  int64_t lddo, uint16_t* __restrict__ dgu, int64_t lddgu, uint16_t* __restrict__ m_out, int64_t ldm, int T, int I) {
    const int chunks = I >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * chunks) return;
    const int t = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + c), g);
    unpack8(*reinterpret_cast<const uint4*>(gu + (int64_t)t * ldgu + I + c), u);
    unpack8(*reinterpret_cast<const uint4*>(dout + (int64_t)t * lddo + c), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sg = sigmoidf_(g[j]);
        du[j] = d[j] * (g[j] * sg);
        dg[j] = d[j] * u[j] * (sg * (1.f + g[j] * (1.f - sg)));
    }
    if (m_out) {                                              // before the stores below: dgu may alias gu
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bfround(g[j] * sigmoidf_(g[j])) * u[j];       // the expression of swiglu_fwd_kernel: bit-identical
        *reinterpret_cast<uint4*>(m_out + (int64_t)t * ldm + c) = pack8(o);
    }
    *reinterpret_cast<uint4*>(dgu + (int64_t)t * lddgu + c) = pack8(dg);
    *reinterpret_cast<uint4*>(dgu + (int64_t)t * lddgu + I + c) = pack8(du);
}

__global__ void gelu_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ out,
                            int64_t n8) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n8) return;
    float f[8], d[8], o[8];
    unpack8(reinterpret_cast<const uint4*>(x)[idx], f);
    if (dy) unpack8(reinterpret_cast<const uint4*>(dy)[idx], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float cdf = 0.5f * (1.f + erff(f[j] * 0.70710678118654752f));
        if (dy) o[j] = d[j] * (cdf + f[j] * 0.3989422804014327f * __expf(-0.5f * f[j] * f[j]));
        else o[j] = f[j] * cdf;
    }
    reinterpret_cast<uint4*>(out)[idx] = pack8(o);
}

// ------------------------------------------------------------------------------ optimizer
// AnyPrecisionAdamW (verl/utils/torch_functional.py:253-329), one pass: 8 params per lane.
// Rounding points follow torch's GPU elementwise kernels (fp32 opmath, one bf16 rounding per op;
// a + alpha*b and a + v*b*c lower to fused multiply-adds) — see oracle/rl_math.py AdamWKahanBF16("gpu").
// Algorithmic bytes per parameter: read p,m,v,c (bf16) + grad (fp32) = 12, write p,m,v,c = 8.
__global__ __launch_bounds__(256) void adamw_kahan_kernel(uint16_t* __restrict__ p, const float* __restrict__ grad,
                                                         uint16_t* __restrict__ m, uint16_t* __restrict__ v,
                                                         uint16_t* __restrict__ c, int64_t n, float wd_mul, float b1,
                                                         float one_m_b1, float b2, float one_m_b2, float eps,
                                                         float neg_step, float dc, const float* __restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    const int64_t n8 = n >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * blockDim.x) {
        float pf[8], mf[8], vf[8], cf[8], gf[8];
        unpack8(reinterpret_cast<const uint4*>(p)[idx], pf);
        unpack8(reinterpret_cast<const uint4*>(m)[idx], mf);
        unpack8(reinterpret_cast<const uint4*>(v)[idx], vf);
        unpack8(reinterpret_cast<const uint4*>(c)[idx], cf);
        const float4 g0 = reinterpret_cast<const float4*>(grad)[idx * 2], g1 = reinterpret_cast<const float4*>(grad)[idx * 2 + 1];
        gf[0] = g0.x; gf[1] = g0.y; gf[2] = g0.z; gf[3] = g0.w; gf[4] = g1.x; gf[5] = g1.y; gf[6] = g1.z; gf[7] = g1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = bfround(gf[j] * gs);
            float pw = wd_mul != 1.f ? bfround(pf[j] * wd_mul) : pf[j];
            const float mn = bfround(__fmaf_rn(one_m_b1, g, bfround(mf[j] * b1)));
            const float vn = bfround(__fmaf_rn(one_m_b2, g * g, bfround(vf[j] * b2)));
            const float cv = bfround(bfround(bfround(__fsqrt_rn(vn)) / dc) + eps);
            const float cn = bfround(__fmaf_rn(neg_step, mn / cv, cf[j]));
            const float pn = bfround(pw + cn);
            cf[j] = bfround(cn + bfround(pw - pn));
            pf[j] = pn; mf[j] = mn; vf[j] = vn;
        }
        reinterpret_cast<uint4*>(p)[idx] = pack8(pf);
        reinterpret_cast<uint4*>(m)[idx] = pack8(mf);
        reinterpret_cast<uint4*>(v)[idx] = pack8(vf);
        reinterpret_cast<uint4*>(c)[idx] = pack8(cf);
    }
    // scalar tail
    if (blockIdx.x == 0) {
        for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
            const float g = bfround(grad[i] * gs);
            const float pf0 = bf2f(p[i]);
            const float pw = wd_mul != 1.f ? bfround(pf0 * wd_mul) : pf0;
            const float mn = bfround(__fmaf_rn(one_m_b1, g, bfround(bf2f(m[i]) * b1)));
            const float vn = bfround(__fmaf_rn(one_m_b2, g * g, bfround(bf2f(v[i]) * b2)));
            const float cv = bfround(bfround(bfround(__fsqrt_rn(vn)) / dc) + eps);
            const float cn = bfround(__fmaf_rn(neg_step, mn / cv, bf2f(c[i])));
            const float pn = bfround(pw + cn);
            c[i] = f2bf(bfround(cn + bfround(pw - pn)));
            p[i] = f2bf(pn); m[i] = f2bf(mn); v[i] = f2bf(vn);
        }
    }
}

// torch.optim.AdamW(fused=True) on bf16 parameters with bf16 exp_avg / exp_avg_sq — what the reference builds for
// worker.actor.optim.strategy=adamw (verl/workers/fsdp_workers.py:284-291).  One pass, no compensation buffer: every quantity is
// widened to fp32 ("opmath"), the python-double hyper-parameters enter the arithmetic as doubles exactly where torch's
// fused kernel mixes them in, and each tensor is rounded to bf16 once when stored.  Algorithmic bytes per parameter:
// read p,m,v (bf16) + grad (fp32) = 10, write p,m,v = 6.
__device__ __forceinline__ void adamw_plain_math(float& p, float g, float& m, float& v, double lr, double b1, double b2, double wd,
                                                double eps, float bc1, float bc2_sqrt) {
    if (wd != 0.0) p = (float)((double)p - lr * wd * (double)p);
    const float w = (float)(1.0 - b1);                                     // lerp(exp_avg, grad, 1 - beta1), weight < 0.5 branch
    m = w < 0.5f ? __fmaf_rn(w, g - m, m) : g - (g - m) * (1.f - w);
    v = (float)(b2 * (double)v + (1.0 - b2) * (double)g * (double)g);
    const float step_size = (float)(lr / (double)bc1);
    const float denom = (float)((double)(__fsqrt_rn(v) / bc2_sqrt) + eps);
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_plain_kernel(uint16_t* __restrict__ p, const float* __restrict__ grad,
                                                         uint16_t* __restrict__ m, uint16_t* __restrict__ v, int64_t n, double lr,
                                                         double b1, double b2, double wd, double eps, float bc1, float bc2_sqrt,
                                                         const float* __restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    const int64_t n8 = n >> 3;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * blockDim.x) {
        float pf[8], mf[8], vf[8], gf[8];
        unpack8(reinterpret_cast<const uint4*>(p)[idx], pf);
        unpack8(reinterpret_cast<const uint4*>(m)[idx], mf);
        unpack8(reinterpret_cast<const uint4*>(v)[idx], vf);
        const float4 g0 = reinterpret_cast<const float4*>(grad)[idx * 2], g1 = reinterpret_cast<const float4*>(grad)[idx * 2 + 1];
        gf[0] = g0.x; gf[1] = g0.y; gf[2] = g0.z; gf[3] = g0.w; gf[4] = g1.x; gf[5] = g1.y; gf[6] = g1.z; gf[7] = g1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) adamw_plain_math(pf[j], bfround(gf[j] * gs), mf[j], vf[j], lr, b1, b2, wd, eps, bc1, bc2_sqrt);
        reinterpret_cast<uint4*>(p)[idx] = pack8(pf);
        reinterpret_cast<uint4*>(m)[idx] = pack8(mf);
        reinterpret_cast<uint4*>(v)[idx] = pack8(vf);
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
            float pf = bf2f(p[i]), mf = bf2f(m[i]), vf = bf2f(v[i]);
            adamw_plain_math(pf, bfround(grad[i] * gs), mf, vf, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
            p[i] = f2bf(pf); m[i] = f2bf(mf); v[i] = f2bf(vf);
        }
    }
}

// fp32 master weights (the reference's default actor: worker.actor.fsdp.torch_dtype unset -> fp32 parameters under FSDP
// MixedPrecision(param_dtype=bf16), torch.optim.AdamW(fused=True) on the fp32 shards, verl/workers/fsdp_workers.py:186-189,284-291):
// the same fused-AdamW arithmetic on fp32 p / exp_avg / exp_avg_sq, and the bf16 working copy the kernels compute with is re-rounded
// from the master in the same pass (what FSDP's next all-gather in param_dtype would produce).  28 B read + 14 B written per parameter.
// On fp32 moments the last bit of the exp_avg update is visible (the bf16 kernels above round it away): torch's fused kernel forms
// beta1 * exp_avg + (1 - beta1) * grad with the python-double betas, i.e. in double, rounded once (measured against
// torch.optim.AdamW(fused=True) on this image: exp_avg / exp_avg_sq bit-identical, tests/test_gpu_train_options.py).
__device__ __forceinline__ void adamw_master_math(float& p, float g, float& m, float& v, double lr, double b1, double b2, double wd,
                                                 double eps, float bc1, float bc2_sqrt) {
    if (wd != 0.0) p = (float)((double)p - lr * wd * (double)p);
    m = (float)(b1 * (double)m + (1.0 - b1) * (double)g);
    v = (float)(b2 * (double)v + (1.0 - b2) * (double)g * (double)g);
    const float step_size = (float)(lr / (double)bc1);
    const float denom = (float)((double)(__fsqrt_rn(v) / bc2_sqrt) + eps);
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(256) void adamw_master_kernel(float* __restrict__ p, uint16_t* __restrict__ pw, const float* __restrict__ grad,
                                                          float* __restrict__ m, float* __restrict__ v, int64_t n, double lr, double b1,
                                                          double b2, double wd, double eps, float bc1, float bc2_sqrt,
                                                          const float* __restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.f;
    const int64_t n4 = n >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (int64_t)gridDim.x * blockDim.x) {
        float4 p4 = reinterpret_cast<float4*>(p)[idx], m4 = reinterpret_cast<float4*>(m)[idx], v4 = reinterpret_cast<float4*>(v)[idx];
        const float4 g4 = reinterpret_cast<const float4*>(grad)[idx];
        adamw_master_math(p4.x, g4.x * gs, m4.x, v4.x, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
        adamw_master_math(p4.y, g4.y * gs, m4.y, v4.y, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
        adamw_master_math(p4.z, g4.z * gs, m4.z, v4.z, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
        adamw_master_math(p4.w, g4.w * gs, m4.w, v4.w, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
        reinterpret_cast<float4*>(p)[idx] = p4; reinterpret_cast<float4*>(m)[idx] = m4; reinterpret_cast<float4*>(v)[idx] = v4;
        reinterpret_cast<uint2*>(pw)[idx] = make_uint2(f2bf2(p4.x, p4.y), f2bf2(p4.z, p4.w));
    }
    if (blockIdx.x == 0) {
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            float pf = p[i], mf = m[i], vf = v[i];
            adamw_master_math(pf, grad[i] * gs, mf, vf, lr, b1, b2, wd, eps, bc1, bc2_sqrt);
            p[i] = pf; m[i] = mf; v[i] = vf; pw[i] = f2bf(pf);
        }
    }
}

// deterministic sum of squares: stage 1 -> scratch[blocks], stage 2 (one block) -> out
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ x, int64_t n, float* __restrict__ scratch) {
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 f = reinterpret_cast<const float4*>(x)[i];
        acc += (double)f.x * f.x + (double)f.y * f.y + (double)f.z * f.z + (double)f.w * f.w;
    }
    if (blockIdx.x == 0) for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += (double)x[i] * x[i];
    acc = wave_sum_d(acc);
    __shared__ double sd[4];
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = (float)(sd[0] + sd[1] + sd[2] + sd[3]);
}
__global__ __launch_bounds__(256) void sumsq_stage2(const float* __restrict__ scratch, int nb, float* __restrict__ out,
                                                   int accumulate) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)scratch[i];
    acc = wave_sum_d(acc);
    __shared__ double sd[4];
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float r = (float)(sd[0] + sd[1] + sd[2] + sd[3]);
        out[0] = accumulate ? out[0] + r : r;
    }
}

// ------------------------------------------------------------------------------ gather / scatter
__global__ void rows_gather_kernel(const uint16_t* __restrict__ src, int64_t lds, const int32_t* __restrict__ rows,
                                   uint16_t* __restrict__ dst, int64_t ldd, int n_rows, int H) {
    const int chunks = H >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_rows * chunks) return;
    const int r = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    *reinterpret_cast<uint4*>(dst + (int64_t)r * ldd + c) = *reinterpret_cast<const uint4*>(src + (int64_t)rows[r] * lds + c);
}
__global__ void rows_scatter_kernel(const uint16_t* __restrict__ src, int64_t lds, const int32_t* __restrict__ rows,
                                    uint16_t* __restrict__ dst, int64_t ldd, int n_rows, int H, int add) {
    const int chunks = H >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_rows * chunks) return;
    const int r = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)r * lds + c);
    uint16_t* d = dst + (int64_t)rows[r] * ldd + c;
    if (add) {
        float a[8], b[8];
        unpack8(v, a);
        unpack8(*reinterpret_cast<const uint4*>(d), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
        v = pack8(a);
    }
    *reinterpret_cast<uint4*>(d) = v;
}
// dst[u] = bf16( sum_r src[idx[u*k + r]] ) in fp32, fixed order r = 0..k-1, entries < 0 skipped (gradient of a row gather
// with repeated sources: image features shared by the rollouts of one prompt)
__global__ void rows_gather_sum_kernel(const uint16_t* __restrict__ src, int64_t lds, const int32_t* __restrict__ idx, int k,
                                       uint16_t* __restrict__ dst, int64_t ldd, int n_out, int H) {
    const int chunks = H >> 3;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)n_out * chunks) return;
    const int u = (int)(id / chunks), c = (int)(id % chunks) * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r) {
        const int32_t row = idx[(int64_t)u * k + r];
        if (row < 0) continue;
        float b[8];
        unpack8(*reinterpret_cast<const uint4*>(src + (int64_t)row * lds + c), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += b[j];
    }
    *reinterpret_cast<uint4*>(dst + (int64_t)u * ldd + c) = pack8(a);
}
__global__ void embed_grad_kernel(const uint16_t* __restrict__ dx, int64_t ldx, const int32_t* __restrict__ ids,
                                  float* __restrict__ dtable, int64_t ldt, int T, int H) {
    const int chunks = H >> 3;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * chunks) return;
    const int t = (int)(idx / chunks), c = (int)(idx % chunks) * 8;
    const int32_t id = ids[t];
    if (id < 0) return;                                   // rows that carry image features: no embedding grad
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(dx + (int64_t)t * ldx + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(dtable + (int64_t)id * ldt + c + j, f[j]);
}
__global__ void cast_pad_kernel(const float* __restrict__ in, int64_t ldin, uint16_t* __restrict__ out, int64_t ldout, int R,
                                int C_in, int C_out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R * C_out) return;
    const int r = (int)(idx / C_out), c = (int)(idx % C_out);
    out[(int64_t)r * ldout + c] = c < C_in ? f2bf(in[(int64_t)r * ldin + c]) : (uint16_t)0;
}
__global__ void add_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ out,
                                int64_t n8) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n8) return;
    float x[8], y[8];
    unpack8(reinterpret_cast<const uint4*>(a)[idx], x);
    unpack8(reinterpret_cast<const uint4*>(b)[idx], y);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] += y[j];
    reinterpret_cast<uint4*>(out)[idx] = pack8(x);
}

// 64x64 tile transpose through LDS: 16-byte global loads (8 lanes cover one 128-byte row segment), 16-byte global stores (8 lanes
// cover one 128-byte segment of an output row); the column gather happens on the LDS side (8 two-byte reads per 16-byte store).
// Ragged edges and unaligned leading dimensions fall back to element-wise access.
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* __restrict__ in, int64_t ldin,
                                                       uint16_t* __restrict__ out, int64_t ldout, int R, int C) {
    constexpr int PITCH = 72;                               // u16 per LDS row: 64 + 8 (16-byte aligned rows, 36-dword stride)
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * PITCH];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bool fast = (r0 + 64 <= R) && (c0 + 64 <= C) && ((ldin & 7) == 0) && ((ldout & 7) == 0) &&
                      ((reinterpret_cast<uintptr_t>(in) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (fast) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = (threadIdx.x >> 3) + 32 * it, c = (threadIdx.x & 7) * 8;
            *reinterpret_cast<uint4*>(tile + r * PITCH + c) = *reinterpret_cast<const uint4*>(in + (int64_t)(r0 + r) * ldin + c0 + c);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int j = (threadIdx.x >> 3) + 32 * it, rc = (threadIdx.x & 7) * 8;     // output row j (= input column), 8 input rows
            uint16_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = tile[(rc + i) * PITCH + j];
            uint4 o;
            o.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); o.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
            o.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); o.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
            *reinterpret_cast<uint4*>(out + (int64_t)(c0 + j) * ldout + r0 + rc) = o;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i * PITCH + tx] = (r < R && c < C) ? in[(int64_t)r * ldin + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(int64_t)c * ldout + r] = tile[tx * PITCH + i];
    }
}

// column sums of a (R, C) bf16 matrix into fp32 (bias gradients): each block owns 64 columns x a row slab
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* __restrict__ in, int64_t ldin, float* __restrict__ out,
                                                    int R, int C, int rows_per_block) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float acc = 0.f;
    if (c < C) for (int r = r0 + ty; r < r1; r += 4) acc += bf2f(in[(int64_t)r * ldin + c]);
    part[ty][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ty == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------ GRPO
// One deterministic 1024-thread workgroup (n = micro-batch * response_length, <= a few 10^4).
__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += sh[k];
    return t;
}

__global__ __launch_bounds__(1024) void grpo_loss_kernel(const float* __restrict__ logp, const float* __restrict__ oldp,
                                                        const float* __restrict__ refp, const float* __restrict__ adv,
                                                        const int64_t* __restrict__ mask, int n, float lo, float hi,
                                                        float dual, int kl_kind, float kl_coef, float inv_accum,
                                                        float* __restrict__ g, float* __restrict__ metrics) {
    __shared__ float sh[16];
    float msum = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) msum += (float)mask[i];
    const float M = block_sum_1024(msum, sh);
    const float inv = 1.f / (M + 1e-8f);
    float a_pg = 0.f, a_hi = 0.f, a_lo = 0.f, a_kl = 0.f, a_ent = 0.f, a_klloss = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float mk = (float)mask[i];
        const float lp = logp[i], d = lp - oldp[i], A = adv[i];
        const float ratio = expf(d);
        const float clipped = expf(fminf(fmaxf(d, lo), hi));
        const float l1 = -A * ratio, l2 = -A * clipped, l3 = -A * dual;
        const float higher = fmaxf(l1, l2);
        const float inside = (d >= lo && d <= hi) ? 1.f : 0.f;
        const float g1 = -A * ratio, g2 = -A * clipped * inside;
        const float sel1 = l1 > l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
        const float g_hi = sel1 * g1 + (1.f - sel1) * g2;
        const float selh = higher < l3 ? 1.f : (higher == l3 ? 0.5f : 0.f);
        float loss, gi;
        if (A < 0.f) { loss = fminf(higher, l3); gi = selh * g_hi; } else { loss = higher; gi = g_hi; }
        a_pg += loss * mk;
        a_hi += (l1 < l2 ? 1.f : 0.f) * mk;
        a_lo += ((higher > l3 && A < 0.f) ? 1.f : 0.f) * mk;
        a_kl += (-d) * mk;
        a_ent += lp * mk;
        if (refp) {
            const float dr = refp[i] - lp;       // ref - logp
            float kl, dk;
            if (kl_kind == 0) { kl = -dr; dk = 1.f; }
            else if (kl_kind == 1) { kl = fabsf(dr); dk = (dr < 0.f) ? 1.f : (dr > 0.f ? -1.f : 0.f); }
            else if (kl_kind == 2) { kl = 0.5f * dr * dr; dk = -dr; }
            else if (kl_kind == 3) {
                const float e = expf(dr), raw = e - dr - 1.f;
                kl = fminf(fmaxf(raw, -10.f), 10.f);
                dk = (raw >= -10.f && raw <= 10.f) ? (1.f - e) : 0.f;
            } else {
                const float r = expf(dr), raw = (r - 1.f) * (r - 1.f);
                kl = fminf(fmaxf(raw, 0.f), 20.f);
                dk = (raw >= 0.f && raw <= 20.f) ? (-2.f * (r - 1.f) * r) : 0.f;
            }
            a_klloss += kl * mk;
            gi += kl_coef * dk;
        }
        g[i] = gi * mk * inv * inv_accum;
    }
    const float pg = block_sum_1024(a_pg, sh) * inv;
    const float fh = block_sum_1024(a_hi, sh) * inv;
    const float fl = block_sum_1024(a_lo, sh) * inv;
    const float pk = block_sum_1024(a_kl, sh) * inv;
    const float en = -block_sum_1024(a_ent, sh) * inv;
    const float kls = block_sum_1024(a_klloss, sh) * inv;
    if (threadIdx.x == 0) {
        metrics[0] = refp ? pg + kls * kl_coef : pg;
        metrics[1] = fh; metrics[2] = fl; metrics[3] = pk; metrics[4] = en; metrics[5] = kls; metrics[6] = M; metrics[7] = 0.f;
    }
}

// Clipped value loss of the critic (core_algos.py:356-392 compute_value_loss, called from dp_critic.py:196-203):
//   loss = 0.5 * masked_mean(max((v - R)^2, (clamp(v, V - c, V + c) - R)^2)), clipfrac = masked_mean((v - R)^2 < (clamped - R)^2),
// masked_mean(x) = sum(x * m) / (sum(m) + 1e-8); g = d(loss / grad_accum) / dv with torch's conventions (clamp passes the gradient inside
// its bounds incl. the bounds, max splits it evenly on a tie — inside the clip range both branches are the same function).
// metrics: [vf_loss, vf_clipfrac, masked_mean(v), sum(mask)].  Single workgroup: deterministic.
__global__ __launch_bounds__(1024) void value_loss_kernel(const float* __restrict__ vpred, const float* __restrict__ returns,
                                                         const float* __restrict__ values, const int64_t* __restrict__ mask, int n, float clip,
                                                         float inv_accum, float* __restrict__ g, float* __restrict__ metrics) {
    __shared__ float sh[16];
    float msum = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) msum += (float)mask[i];
    const float M = block_sum_1024(msum, sh);
    const float inv = 1.f / (M + 1e-8f);
    float a_loss = 0.f, a_clip = 0.f, a_v = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float mk = (float)mask[i];
        const float v = vpred[i], R = returns[i], lo = values[i] - clip, hi = values[i] + clip;
        const float vc = fminf(fmaxf(v, lo), hi);
        const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
        const float inside = (v >= lo && v <= hi) ? 1.f : 0.f;
        const float g1 = v - R, g2 = (vc - R) * inside;                   // d(0.5 l1)/dv, d(0.5 l2)/dv
        const float sel1 = l1 > l2 ? 1.f : (l1 == l2 ? 0.5f : 0.f);
        a_loss += fmaxf(l1, l2) * mk;
        a_clip += (l1 < l2 ? 1.f : 0.f) * mk;
        a_v += v * mk;
        g[i] = (sel1 * g1 + (1.f - sel1) * g2) * mk * inv * inv_accum;
    }
    const float L = 0.5f * block_sum_1024(a_loss, sh) * inv;
    const float C = block_sum_1024(a_clip, sh) * inv;
    const float V = block_sum_1024(a_v, sh) * inv;
    if (threadIdx.x == 0) { metrics[0] = L; metrics[1] = C; metrics[2] = V; metrics[3] = M; }
}

// GRPO outcome advantage.  Stage A: one thread per row -> score.  Stage B: one thread per group scans the rows
// in order (sequential fp32 sum for the mean, fp64 Welford for the unbiased std — torch CPU semantics).
// Stage C: broadcast over the response mask.  Single workgroup per stage-B chunk keeps it deterministic.
__global__ void grpo_scores_kernel(const float* __restrict__ rewards, int N, int R, float* __restrict__ scores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s = 0.f;
    for (int t = 0; t < R; ++t) s += rewards[(int64_t)i * R + t];
    scores[i] = s;
}
__global__ void grpo_group_stats_kernel(const float* __restrict__ scores, const int32_t* __restrict__ group, int N,
                                        int n_groups, float* __restrict__ mean_out, float* __restrict__ std_out,
                                        int32_t* __restrict__ status) {
    const int gidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (gidx >= n_groups) return;
    float sum = 0.f;
    int cnt = 0;
    double wm = 0.0, m2 = 0.0;
    for (int i = 0; i < N; ++i) {
        if (group[i] != gidx) continue;
        const float x = scores[i];
        sum += x;
        ++cnt;
        const double delta = (double)x - wm;
        wm += delta / (double)cnt;
        m2 += delta * ((double)x - wm);
    }
    if (cnt < 2) { if (status) atomicExch(status, -1); mean_out[gidx] = 0.f; std_out[gidx] = 0.f; return; }
    mean_out[gidx] = sum / (float)cnt;
    std_out[gidx] = (float)sqrt(m2 / (double)(cnt - 1));
}
__global__ void grpo_broadcast_kernel(const float* __restrict__ scores, const int32_t* __restrict__ group,
                                      const float* __restrict__ mean, const float* __restrict__ stdv,
                                      const int64_t* __restrict__ mask, int N, int R, float eps, float* __restrict__ adv) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * R) return;
    const int i = (int)(idx / R);
    const int gi = group[i];
    const float a = (scores[i] - mean[gi]) / (stdv[gi] + eps);
    adv[idx] = a * (float)mask[idx];
}

// ------------------------------------------------------------------------------ C ABI
extern "C" {

int st_mrope_table(const int32_t* pos, const float* inv_freq, int T, int D, int s0, int s1, int s2, float* cos_out,
                   float* sin_out, st_stream_t stream) {
    if (!pos || !inv_freq || !cos_out || !sin_out || T < 0 || D <= 0 || (D & 1) || s0 + s1 + s2 != D / 2) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * (D / 2);
    hipLaunchKernelGGL(mrope_table_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, inv_freq, T, D / 2,
                       s0, s1, cos_out, sin_out);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_rope_apply(st_bf16* x, int64_t ld, const float* cos_tab, const float* sin_tab, int T, int n_rot_heads, int D,
                  int inverse, st_stream_t stream) {
    if (!x || !cos_tab || !sin_tab || T < 0 || n_rot_heads <= 0 || D <= 0 || (D % 16) || (ld & 7)) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * n_rot_heads * (D / 16);
    hipLaunchKernelGGL(rope_apply_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, cos_tab, sin_tab,
                       T, n_rot_heads, D, inverse ? -1.f : 1.f);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_swiglu_fwd(const st_bf16* gu, int64_t ldgu, st_bf16* out, int64_t ldo, int T, int I, st_stream_t stream) {
    if (!gu || !out || T < 0 || I <= 0 || (I & 7) || (ldgu & 7) || (ldo & 7)) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * (I / 8);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gu, ldgu, out, ldo, T, I);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_swiglu_mxfp8(const st_bf16* gu, int64_t ldgu, st_bf16* out, int64_t ldo, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows,
                    int T, int I, st_stream_t stream) {
    if (!gu || !q || !scales || T < 0 || I <= 0 || (I % 128) || (ldgu & 7) || (out && (ldo & 7)) || (ldq & 7) || ldq < I || scale_rows < T ||
        (((uintptr_t)q) & 7))
        return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t items = (int64_t)T * ((I + 511) / 512);
    hipLaunchKernelGGL(swiglu_mxfp8_kernel, dim3(st_cdiv(items, 4)), dim3(256), 0, (hipStream_t)stream, gu, ldgu, out, ldo, q, ldq, scales,
                       scale_rows, T, I);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_swiglu_bwd(const st_bf16* gu, int64_t ldgu, const st_bf16* dout, int64_t lddo, st_bf16* dgu, int64_t lddgu, int T,
                  int I, st_stream_t stream) {
    return st_swiglu_bwd_m(gu, ldgu, dout, lddo, dgu, lddgu, nullptr, 0, T, I, stream);
}

int st_swiglu_bwd_m(const st_bf16* gu, int64_t ldgu, const st_bf16* dout, int64_t lddo, st_bf16* dgu, int64_t lddgu, st_bf16* m_out,
                    int64_t ldm, int T, int I, st_stream_t stream) {
    if (!gu || !dout || !dgu || T < 0 || I <= 0 || (I & 7) || (ldgu & 7) || (lddo & 7) || (lddgu & 7) || (m_out && ((ldm & 7) || ldm < I))) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * (I / 8);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gu, ldgu, dout, lddo, dgu,
                       lddgu, m_out, ldm, T, I);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_gelu_fwd(const st_bf16* x, st_bf16* y, int64_t n, st_stream_t stream) {
    if (!x || !y || n < 0 || (n & 7)) return ST_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3(st_cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (const uint16_t*)nullptr, y,
                       n / 8);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_gelu_bwd(const st_bf16* x, const st_bf16* dy, st_bf16* dx, int64_t n, st_stream_t stream) {
    if (!x || !dy || !dx || n < 0 || (n & 7)) return ST_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(gelu_kernel, dim3(st_cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n / 8);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_adamw_kahan_step(st_bf16* p, const float* grad, st_bf16* m, st_bf16* v, st_bf16* c, int64_t n, double lr, double beta1,
                        double beta2, double eps, double weight_decay, float step_size, float denom_corr,
                        const float* grad_scale, st_stream_t stream) {
    if (!p || !grad || !m || !v || !c || n < 0) return ST_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // the python-side scalars, formed the way torch forms them: double arithmetic, then one cast to fp32
    const float wd_mul = weight_decay != 0.0 ? (float)(1.0 - lr * weight_decay) : 1.f;
    const float one_m_b1 = (float)(1.0 - beta1), one_m_b2 = (float)(1.0 - beta2);
    StProfScope ps(ST_K_ADAMW, s, 20.0 * (double)n);
    int blocks = st_cdiv(n / 8 + 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_kahan_kernel, dim3(blocks), dim3(256), 0, s, p, grad, m, v, c, n, wd_mul, (float)beta1, one_m_b1,
                       (float)beta2, one_m_b2, (float)eps, -step_size, denom_corr, grad_scale);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_adamw_step(st_bf16* p, const float* grad, st_bf16* m, st_bf16* v, int64_t n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, float bias_correction1, float bias_correction2_sqrt, const float* grad_scale, st_stream_t stream) {
    if (!p || !grad || !m || !v || n < 0) return ST_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_ADAMW, s, 16.0 * (double)n);
    int blocks = st_cdiv(n / 8 + 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_plain_kernel, dim3(blocks), dim3(256), 0, s, p, grad, m, v, n, lr, beta1, beta2, weight_decay, eps,
                       bias_correction1, bias_correction2_sqrt, grad_scale);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_adamw_master_step(float* master, st_bf16* p_bf16, const float* grad, float* m, float* v, int64_t n, double lr, double beta1,
                         double beta2, double eps, double weight_decay, float bias_correction1, float bias_correction2_sqrt,
                         const float* grad_scale, st_stream_t stream) {
    if (!master || !p_bf16 || !grad || !m || !v || n < 0 || (((uintptr_t)master | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) ||
        (((uintptr_t)p_bf16) & 7))
        return ST_EINVAL;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_ADAMW, s, 42.0 * (double)n);
    int blocks = st_cdiv(n / 4 + 1, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(adamw_master_kernel, dim3(blocks), dim3(256), 0, s, master, p_bf16, grad, m, v, n, lr, beta1, beta2, weight_decay, eps,
                       bias_correction1, bias_correction2_sqrt, grad_scale);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_sumsq_f32(const float* x, int64_t n, float* scratch, float* out, int accumulate, st_stream_t stream) {
    if (!x || !scratch || !out || n < 0) return ST_EINVAL;
    int nb = st_cdiv(n / 4 + 1, 256 * 8);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sumsq_stage1, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, out, accumulate);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_embed_gather(const st_bf16* table, int64_t ldt, const int32_t* ids, st_bf16* out, int64_t ldo, int T, int H,
                    st_stream_t stream) {
    if (!table || !ids || !out || T < 0 || H <= 0 || (H & 7) || (ldt & 7) || (ldo & 7)) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * (H / 8);
    hipLaunchKernelGGL(rows_gather_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, table, ldt, ids, out, ldo, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_rows_gather(const st_bf16* src, int64_t lds, const int32_t* rows, st_bf16* dst, int64_t ldd, int n_rows, int H,
                   st_stream_t stream) {
    return st_embed_gather(src, lds, rows, dst, ldd, n_rows, H, stream);
}
int st_rows_scatter(const st_bf16* src, int64_t lds, const int32_t* rows, st_bf16* dst, int64_t ldd, int n_rows, int H, int add,
                    st_stream_t stream) {
    if (!src || !rows || !dst || n_rows < 0 || H <= 0 || (H & 7) || (lds & 7) || (ldd & 7)) return ST_EINVAL;
    if (n_rows == 0) return 0;
    const int64_t n = (int64_t)n_rows * (H / 8);
    hipLaunchKernelGGL(rows_scatter_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, lds, rows, dst, ldd,
                       n_rows, H, add);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_rows_gather_sum(const st_bf16* src, int64_t lds, const int32_t* idx, int k, st_bf16* dst, int64_t ldd, int n_out, int H,
                       st_stream_t stream) {
    if (!src || !idx || !dst || k <= 0 || n_out < 0 || H <= 0 || (H & 7) || (lds & 7) || (ldd & 7)) return ST_EINVAL;
    if (n_out == 0) return 0;
    const int64_t n = (int64_t)n_out * (H / 8);
    hipLaunchKernelGGL(rows_gather_sum_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, lds, idx, k, dst, ldd,
                       n_out, H);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_embed_grad(const st_bf16* dx, int64_t ldx, const int32_t* ids, float* dtable, int64_t ldt, int T, int H,
                  st_stream_t stream) {
    if (!dx || !ids || !dtable || T < 0 || H <= 0 || (H & 7) || (ldx & 7)) return ST_EINVAL;
    if (T == 0) return 0;
    const int64_t n = (int64_t)T * (H / 8);
    hipLaunchKernelGGL(embed_grad_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dx, ldx, ids, dtable, ldt, T, H);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_cast_pad_f32_bf16(const float* in, int64_t ldin, st_bf16* out, int64_t ldout, int R, int C_in, int C_out,
                         st_stream_t stream) {
    if (!in || !out || R < 0 || C_in <= 0 || C_out < C_in) return ST_EINVAL;
    if (R == 0) return 0;
    const int64_t n = (int64_t)R * C_out;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(st_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in, ldin, out, ldout, R, C_in, C_out);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_add_bf16(const st_bf16* a, const st_bf16* b, st_bf16* out, int64_t n, st_stream_t stream) {
    if (!a || !b || !out || n < 0 || (n & 7)) return ST_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(st_cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 8);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_transpose(const st_bf16* in, int64_t ldin, st_bf16* out, int64_t ldout, int R, int C, st_stream_t stream) {
    if (!in || !out || R < 0 || C < 0) return ST_EINVAL;
    if (R == 0 || C == 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3(st_cdiv(C, 64), st_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, in, ldin, out,
                       ldout, R, C);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_colsum(const st_bf16* in, int64_t ldin, float* out_f32, int accumulate, int R, int C, st_stream_t stream) {
    if (!in || !out_f32 || R < 0 || C <= 0) return ST_EINVAL;
    if (!accumulate) hipMemsetAsync(out_f32, 0, sizeof(float) * (size_t)C, (hipStream_t)stream);
    if (R == 0) return 0;
    const int rpb = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(st_cdiv(C, 64), st_cdiv(R, rpb)), dim3(256), 0, (hipStream_t)stream, in, ldin, out_f32,
                       R, C, rpb);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_grpo_loss(const float* logp, const float* old_logp, const float* ref_logp, const float* adv, const int64_t* mask, int n,
                 double clip_low, double clip_high, double clip_dual, int kl_kind, double kl_coef, double grad_accum, float* g,
                 float* metrics, st_stream_t stream) {
    if (!logp || !old_logp || !adv || !mask || !g || !metrics || n <= 0 || kl_kind < 0 || kl_kind > 4 || !(grad_accum > 0.0))
        return ST_EINVAL;
    // clip bounds exactly as the reference forms them: np.log(1 -/+ clip) in double, used as fp32 (core_algos.py:334-336)
    const float lo = (float)log(1.0 - clip_low), hi = (float)log(1.0 + clip_high);
    hipLaunchKernelGGL(grpo_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, old_logp, ref_logp, adv, mask, n, lo,
                       hi, (float)clip_dual, kl_kind, (float)kl_coef, (float)(1.0 / grad_accum), g, metrics);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_value_loss(const float* vpreds, const float* returns, const float* values, const int64_t* mask, int n, double cliprange_value,
                  double grad_accum, float* g, float* metrics, st_stream_t stream) {
    if (!vpreds || !returns || !values || !mask || !g || !metrics || n <= 0 || !(grad_accum > 0.0)) return ST_EINVAL;
    hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, vpreds, returns, values, mask, n, (float)cliprange_value,
                       (float)(1.0 / grad_accum), g, metrics);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_grpo_advantage(const float* rewards, const int64_t* mask, const int32_t* group, int N, int R, int n_groups, double eps,
                      float* adv, float* scratch, int32_t* status, st_stream_t stream) {
    // scratch: N + 2*n_groups floats
    if (!rewards || !mask || !group || !adv || !scratch || N <= 0 || R <= 0 || n_groups <= 0) return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* scores = scratch;
    float* mean = scratch + N;
    float* stdv = mean + n_groups;
    hipLaunchKernelGGL(grpo_scores_kernel, dim3(st_cdiv(N, 256)), dim3(256), 0, s, rewards, N, R, scores);
    hipLaunchKernelGGL(grpo_group_stats_kernel, dim3(st_cdiv(n_groups, 64)), dim3(64), 0, s, scores, group, N, n_groups, mean, stdv,
                       status);
    hipLaunchKernelGGL(grpo_broadcast_kernel, dim3(st_cdiv((int64_t)N * R, 256)), dim3(256), 0, s, scores, group, mean, stdv, mask,
                       N, R, (float)eps, adv);
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
