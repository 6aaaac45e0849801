// attention.hip — varlen flash attention for gfx950 (causal GQA for the LM, bidirectional for the ViT).
//
// Layout / mapping (wave = 64 lanes, MFMA 32x32x16 bf16, fp32 accumulate):
//   * workgroup = 4 waves = 128 query rows of ONE sequence and ONE query head; wave w owns rows 32w..32w+31;
//   * K/V tiles of 64 keys are staged HBM -> registers -> LDS once per workgroup and shared by its 4 waves:
//       K tile  [64][D]      row pitch 2D+16 B  (16-byte pad => conflict-free ds_read_b128 fragment reads)
//       V tile  [DP][64]^T   row pitch 136 B    (transposed while staging so PV fragments are key-contiguous)
//   * scores are computed TRANSPOSED, S^T = K Q^T, so a lane holds one query column: the softmax row
//     reductions are in-lane plus one cross-half `__shfl_xor 32` (no LDS, no serial lanes);
//   * P^T feeds the PV MFMA straight from registers: the accumulator rows a lane holds
//     (keys {0-3,8-11}+4*half, {16-19,24-27}+4*half per 32-key block) are used as the MFMA k-index, and the
//     V^T fragment is read with the same key permutation, so no cross-lane movement is needed;
//   * O^T (d x q) accumulates in registers; repeat_kv of the reference is never materialised: the kv head is
//     h / (n_q / n_kv).
// Roofline: MFMA-bound, 4*D*Lq*Lk flop per (sequence, head) pair (half of that when causal).
#include "common.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define KV_TILE 64
#define Q_TILE 128
#define VT_PITCH 136            // bytes per V^T row (64 keys * 2 B + 8 pad)

template <int D> struct AttnCfg {
    static constexpr int DP = (D + 31) / 32 * 32;       // PV output rows padded to MFMA 32
    static constexpr int KS = D / 16;                   // QK^T k-steps
    static constexpr int KPITCH = D * 2 + 16;           // bytes per K row in LDS
    static constexpr int K_BYTES = KV_TILE * KPITCH;
    static constexpr int V_BYTES = DP * VT_PITCH;
};

__device__ __forceinline__ uint32_t pack2bf(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

// Stage one K tile (natural) and one V tile (transposed) into LDS.  256 threads.
template <int D>
__device__ __forceinline__ void stage_kv(const uint16_t* __restrict__ k, int64_t ldk, const uint16_t* __restrict__ v, int64_t ldv,
                                         int64_t row0, int kt0, int L, int kv_col, char* ks, char* vt) {
    constexpr int CH = D / 8;                           // 16-byte chunks per row
    constexpr int KP = AttnCfg<D>::KPITCH;
    for (int c = threadIdx.x; c < KV_TILE * CH; c += 256) {
        const int key = c / CH, dc = c % CH;
        uint4 kk = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (kt0 + key < L) {
            kk = *reinterpret_cast<const uint4*>(k + (row0 + kt0 + key) * ldk + kv_col + dc * 8);
            vv = *reinterpret_cast<const uint4*>(v + (row0 + kt0 + key) * ldv + kv_col + dc * 8);
        }
        *reinterpret_cast<uint4*>(ks + key * KP + dc * 16) = kk;
        uint16_t* vdst = reinterpret_cast<uint16_t*>(vt + (dc * 8) * VT_PITCH) + key;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vdst[(2 * j) * (VT_PITCH / 2)] = (uint16_t)(w[j] & 0xffffu);
            vdst[(2 * j + 1) * (VT_PITCH / 2)] = (uint16_t)(w[j] >> 16);
        }
    }
    if (AttnCfg<D>::DP != D) {                          // zero the padded V^T rows once per tile
        for (int c = threadIdx.x; c < (AttnCfg<D>::DP - D) * (VT_PITCH / 8); c += 256)
            *reinterpret_cast<uint2*>(vt + D * VT_PITCH + c * 8) = make_uint2(0, 0);
    }
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                      const uint16_t* __restrict__ k, int64_t ldk,
                                                      const uint16_t* __restrict__ v, int64_t ldv,
                                                      const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end,
                                                      const int32_t* __restrict__ k_beg, const int32_t* __restrict__ k_end,
                                                      const int32_t* __restrict__ o_beg, int qgroup, int T, int n_q, int n_kv,
                                                      float scale_log2, uint16_t* __restrict__ out, int64_t ldo,
                                                      float* __restrict__ lse) {
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) char smem[C::K_BYTES + C::V_BYTES];
    char* ks = smem;
    char* vt = smem + C::K_BYTES;

    // sequence `seq`: query rows [q_beg, q_end) of the q tensor attend to key rows [k_beg, k_end) of k/v.
    // Prefill/training: both ranges are cu_seqlens[seq], cu_seqlens[seq+1]; decode: Lq = group rows, Lk = cache length.
    const int seq = blockIdx.z, h = blockIdx.y;
    const int s0 = q_beg[seq], Lq = q_end[seq] - s0;
    const int sk = k_beg[seq], L = k_end[seq] - sk;
    const int q_base = blockIdx.x * Q_TILE;
    if (q_base >= Lq) return;
    const int kvh = h / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qc = lane & 31, half = lane >> 5;
    const int q_idx = q_base + wave * 32 + qc;           // row inside the sequence
    const bool q_ok = q_idx < Lq;

    // Q fragments (MFMA B operand: n = q row, k = 8 contiguous d), resident for the whole kernel
    bf16x8 qf[C::KS];
    {
        // qgroup > 0 (decode): query row R of this launch is (sample b = R / g, group-local head R % g) of KV head h, read
        // straight from the (B, n_q*D) projection output — no permuted copy of q is ever made.
        const int64_t R = s0 + (q_ok ? q_idx : 0);
        const uint16_t* qp = qgroup > 0 ? q + (R / qgroup) * ldq + ((int64_t)h * qgroup + R % qgroup) * D + half * 8
                                        : q + R * ldq + (int64_t)h * D + half * 8;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            uint4 r = q_ok ? *reinterpret_cast<const uint4*>(qp + s * 16) : make_uint4(0, 0, 0, 0);
            qf[s] = *reinterpret_cast<bf16x8*>(&r);
        }
    }
    f32x16 o[C::DP / 32];
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    const int kv_end = CAUSAL ? min(L, q_base + Q_TILE) : L;
    for (int kt0 = 0; kt0 < kv_end; kt0 += KV_TILE) {
        __syncthreads();
        stage_kv<D>(k, ldk, v, ldv, (int64_t)sk, kt0, L, kvh * D, ks, vt);
        __syncthreads();
        // wave-uniform skip of tiles entirely above this wave's diagonal
        if (CAUSAL && kt0 > q_base + wave * 32 + 31) continue;

        // ---- S^T = K Q^T : two 32-key blocks
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            const char* kp = ks + (kb * 32 + qc) * C::KPITCH + half * 16;
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + s * 32);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (base 2).  Accumulator row r <-> key (r&3) + 8*(r>>2) + 4*half
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool ok = (key < L) && (!CAUSAL || key <= q_idx);
                const float sv = ok ? sacc[kb][r] * scale_log2 : -INFINITY;
                sacc[kb][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_i, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_i - m_use);          // m_i = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(sacc[kb][r] - m_use);
                sacc[kb][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_i = l_i * alpha + rs;
        m_i = m_new;
#pragma unroll
        for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        // ---- O^T += V^T P^T : k-slot ks2 covers accumulator rows (ks2&1)*8 .. +7 of block ks2>>1
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ++ks2) {
            const int kb = ks2 >> 1, rb = (ks2 & 1) * 8;
            uint4 pw;
            pw.x = pack2bf(sacc[kb][rb + 0], sacc[kb][rb + 1]);
            pw.y = pack2bf(sacc[kb][rb + 2], sacc[kb][rb + 3]);
            pw.z = pack2bf(sacc[kb][rb + 4], sacc[kb][rb + 5]);
            pw.w = pack2bf(sacc[kb][rb + 6], sacc[kb][rb + 7]);
            const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
            const int key0 = kb * 32 + (ks2 & 1) * 16 + 4 * half;      // keys key0..+3 and key0+8..+11
#pragma unroll
            for (int b = 0; b < C::DP / 32; ++b) {
                const char* vp = vt + (b * 32 + qc) * VT_PITCH + key0 * 2;
                uint4 vw;
                const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
                vw.x = lo.x; vw.y = lo.y; vw.z = hi.x; vw.w = hi.y;
                const bf16x8 vf = *reinterpret_cast<bf16x8*>(&vw);
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[b], 0, 0, 0);
            }
        }
    }
    if (!q_ok) return;
    const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;
    const int64_t orow = (o_beg ? o_beg[seq] : s0) + q_idx;     // decode partials land in their own slab
    uint16_t* op = out + orow * ldo + (int64_t)h * D;
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            if (d < D) {
                uint2 w;
                w.x = pack2bf(o[b][4 * g + 0] * inv_l, o[b][4 * g + 1] * inv_l);
                w.y = pack2bf(o[b][4 * g + 2] * inv_l, o[b][4 * g + 3] * inv_l);
                *reinterpret_cast<uint2*>(op + d) = w;
            }
        }
    if (half == 0) lse[(int64_t)h * T + orow] = l_i > 0.f ? (m_i + log2f(l_i)) * LN2 : -INFINITY;
}


// =============================================================================================
// Backward.  Two deterministic kernels (no atomics):
//   attn_bwd_dq_kernel  — same decomposition as the forward (128 q rows x one head per workgroup); per KV
//                         tile recomputes S^T, forms dP^T = V dO^T, dS^T = P^T o (dP^T - delta) * scale and
//                         accumulates dQ^T = K^T dS^T in registers;
//   attn_bwd_dkv_kernel — one workgroup per 128 keys of one KV head (wave = 32 keys, K/V fragments register
//                         resident); streams 32-row Q/dO tiles of every query head of the group through LDS
//                         (natural + transposed images), accumulating dK^T and dV^T in registers, so the GQA
//                         sum over the group's query heads needs no atomics.
// delta[h][t] = sum_d dO*O is produced by attn_bwd_delta_kernel.
// Flop: 7 matmuls of 2*D*Lq*Lk (S and dP are computed in both kernels) vs 5 for an atomic-dQ variant.
// =============================================================================================
__global__ void attn_bwd_delta_kernel(const uint16_t* __restrict__ o, int64_t ldo, const uint16_t* __restrict__ dout,
                                      int64_t lddo, int T, int n_q, int D, float* __restrict__ delta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)T * n_q) return;
    const int t = (int)(idx / n_q), h = (int)(idx % n_q);
    const uint16_t* op = o + (int64_t)t * ldo + (int64_t)h * D;
    const uint16_t* dp = dout + (int64_t)t * lddo + (int64_t)h * D;
    float acc = 0.f;
    for (int c = 0; c < D; c += 8) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const uint4*>(op + c), a);
        unpack8(*reinterpret_cast<const uint4*>(dp + c), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j] * b[j];
    }
    delta[(int64_t)h * T + t] = acc;
}

// natural [ROWS][D] (pitch 2D+16) and optionally transposed [D][ROWS] (pitch TP) images of a (ROWS x D) tile
template <int D, int ROWS, int TP, bool WITH_T>
__device__ __forceinline__ void stage_tile_nt(const uint16_t* __restrict__ src, int64_t ld, int64_t row0, int r0, int L, int col,
                                              char* nat, char* tr) {
    constexpr int CH = D / 8;
    constexpr int NP = D * 2 + 16;
    for (int c = threadIdx.x; c < ROWS * CH; c += 256) {
        const int r = c / CH, dc = c % CH;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (r0 + r < L) x = *reinterpret_cast<const uint4*>(src + (row0 + r0 + r) * ld + col + dc * 8);
        *reinterpret_cast<uint4*>(nat + r * NP + dc * 16) = x;
        if (WITH_T) {
            uint16_t* d = reinterpret_cast<uint16_t*>(tr + (dc * 8) * TP) + r;
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[(2 * j) * (TP / 2)] = (uint16_t)(w[j] & 0xffffu);
                d[(2 * j + 1) * (TP / 2)] = (uint16_t)(w[j] >> 16);
            }
        }
    }
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                         const uint16_t* __restrict__ k, int64_t ldk,
                                                         const uint16_t* __restrict__ v, int64_t ldv,
                                                         const uint16_t* __restrict__ dout, int64_t lddo,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         const int32_t* __restrict__ cu, int T, int n_q, int n_kv,
                                                         float scale, uint16_t* __restrict__ dq, int64_t lddq) {
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::K_BYTES + C::V_BYTES];
    char* ks = smem;                       // K natural  [64][D]
    char* vs = smem + C::K_BYTES;          // V natural  [64][D]
    char* kt = smem + 2 * C::K_BYTES;      // K^T        [DP][64]
    const int seq = blockIdx.z, h = blockIdx.y;
    const int s0 = cu[seq], L = cu[seq + 1] - s0;
    const int q_base = blockIdx.x * Q_TILE;
    if (q_base >= L) return;
    const int kvh = h / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qc = lane & 31, half = lane >> 5;
    const int q_idx = q_base + wave * 32 + qc;
    const bool q_ok = q_idx < L;
    const float scale_log2 = scale * LOG2E;

    bf16x8 qf[C::KS], dof[C::KS];
    {
        const int64_t row = s0 + (q_ok ? q_idx : 0);
        const uint16_t* qp = q + row * ldq + (int64_t)h * D + half * 8;
        const uint16_t* dp = dout + row * lddo + (int64_t)h * D + half * 8;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            uint4 a = q_ok ? *reinterpret_cast<const uint4*>(qp + s * 16) : make_uint4(0, 0, 0, 0);
            uint4 b = q_ok ? *reinterpret_cast<const uint4*>(dp + s * 16) : make_uint4(0, 0, 0, 0);
            qf[s] = *reinterpret_cast<bf16x8*>(&a);
            dof[s] = *reinterpret_cast<bf16x8*>(&b);
        }
    }
    const float lse2 = q_ok ? lse[(int64_t)h * T + s0 + q_idx] * LOG2E : 0.f;
    const float dlt = q_ok ? delta[(int64_t)h * T + s0 + q_idx] : 0.f;
    f32x16 acc[C::DP / 32];
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int kv_end = CAUSAL ? min(L, q_base + Q_TILE) : L;
    for (int kt0 = 0; kt0 < kv_end; kt0 += KV_TILE) {
        __syncthreads();
        stage_tile_nt<D, KV_TILE, VT_PITCH, true>(k, ldk, (int64_t)s0, kt0, L, kvh * D, ks, kt);
        stage_tile_nt<D, KV_TILE, VT_PITCH, false>(v, ldv, (int64_t)s0, kt0, L, kvh * D, vs, nullptr);
        __syncthreads();
        if (CAUSAL && kt0 > q_base + wave * 32 + 31) continue;
        f32x16 sacc[2], pacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[kb][r] = 0.f; pacc[kb][r] = 0.f; }
            const char* kp = ks + (kb * 32 + qc) * C::KPITCH + half * 16;
            const char* vp = vs + (kb * 32 + qc) * C::KPITCH + half * 16;
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + s * 32);
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vp + s * 32);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);     // S^T  = K Q^T
                pacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[s], pacc[kb], 0, 0, 0);    // dP^T = V dO^T
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool ok = q_ok && (key < L) && (!CAUSAL || key <= q_idx);
                const float p = ok ? exp2f(sacc[kb][r] * scale_log2 - lse2) : 0.f;
                sacc[kb][r] = p * (pacc[kb][r] - dlt) * scale;                                         // dS^T
            }
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ++ks2) {
            const int kb = ks2 >> 1, rb = (ks2 & 1) * 8;
            uint4 pw;
            pw.x = pack2bf(sacc[kb][rb + 0], sacc[kb][rb + 1]);
            pw.y = pack2bf(sacc[kb][rb + 2], sacc[kb][rb + 3]);
            pw.z = pack2bf(sacc[kb][rb + 4], sacc[kb][rb + 5]);
            pw.w = pack2bf(sacc[kb][rb + 6], sacc[kb][rb + 7]);
            const bf16x8 df = *reinterpret_cast<bf16x8*>(&pw);
            const int key0 = kb * 32 + (ks2 & 1) * 16 + 4 * half;
#pragma unroll
            for (int b = 0; b < C::DP / 32; ++b) {
                const char* tp = kt + (b * 32 + qc) * VT_PITCH + key0 * 2;
                uint4 tw;
                const uint2 lo = *reinterpret_cast<const uint2*>(tp);
                const uint2 hi = *reinterpret_cast<const uint2*>(tp + 16);
                tw.x = lo.x; tw.y = lo.y; tw.z = hi.x; tw.w = hi.y;
                const bf16x8 tf = *reinterpret_cast<bf16x8*>(&tw);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf, df, acc[b], 0, 0, 0);             // dQ^T += K^T dS^T
            }
        }
    }
    if (!q_ok) return;
    uint16_t* op = dq + (int64_t)(s0 + q_idx) * lddq + (int64_t)h * D;
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            if (d < D) {
                uint2 w;
                w.x = pack2bf(acc[b][4 * g + 0], acc[b][4 * g + 1]);
                w.y = pack2bf(acc[b][4 * g + 2], acc[b][4 * g + 3]);
                *reinterpret_cast<uint2*>(op + d) = w;
            }
        }
}

#define QT_ROWS 32
#define QT_PITCH 72            // bytes per row of the transposed [D][32] images (32*2 + 8 pad)
template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                          const uint16_t* __restrict__ k, int64_t ldk,
                                                          const uint16_t* __restrict__ v, int64_t ldv,
                                                          const uint16_t* __restrict__ dout, int64_t lddo,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          const int32_t* __restrict__ cu, int T, int n_q, int n_kv,
                                                          float scale, uint16_t* __restrict__ dk, int64_t lddk,
                                                          uint16_t* __restrict__ dv, int64_t lddv) {
    using C = AttnCfg<D>;
    constexpr int NAT = QT_ROWS * C::KPITCH;
    constexpr int TRB = C::DP * QT_PITCH;
    __shared__ __attribute__((aligned(16))) char smem[2 * NAT + 2 * TRB + 2 * QT_ROWS * 4];
    char* qs = smem;                 // Q natural   [32][D]
    char* dos = smem + NAT;          // dO natural  [32][D]
    char* qt = smem + 2 * NAT;       // Q^T         [DP][32]
    char* dot = qt + TRB;            // dO^T        [DP][32]
    float* s_lse = reinterpret_cast<float*>(dot + TRB);
    float* s_dlt = s_lse + QT_ROWS;

    const int seq = blockIdx.z, kvh = blockIdx.y;
    const int s0 = cu[seq], L = cu[seq + 1] - s0;
    const int k_base = blockIdx.x * 128;
    if (k_base >= L) return;
    const int group = n_q / n_kv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kc = lane & 31, half = lane >> 5;
    const int key_idx = k_base + wave * 32 + kc;
    const bool k_ok = key_idx < L;
    const float scale_log2 = scale * LOG2E;

    bf16x8 kf[C::KS], vf[C::KS];      // MFMA B operands: n = key, k = 8 contiguous d
    {
        const int64_t row = s0 + (k_ok ? key_idx : 0);
        const uint16_t* kp = k + row * ldk + (int64_t)kvh * D + half * 8;
        const uint16_t* vp = v + row * ldv + (int64_t)kvh * D + half * 8;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
            uint4 a = k_ok ? *reinterpret_cast<const uint4*>(kp + s * 16) : make_uint4(0, 0, 0, 0);
            uint4 b = k_ok ? *reinterpret_cast<const uint4*>(vp + s * 16) : make_uint4(0, 0, 0, 0);
            kf[s] = *reinterpret_cast<bf16x8*>(&a);
            vf[s] = *reinterpret_cast<bf16x8*>(&b);
        }
    }
    f32x16 dka[C::DP / 32], dva[C::DP / 32];
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[b][r] = 0.f; dva[b][r] = 0.f; }

    const int q_start = CAUSAL ? (k_base / QT_ROWS) * QT_ROWS : 0;
    for (int hq = kvh * group; hq < (kvh + 1) * group; ++hq) {
        for (int qt0 = q_start; qt0 < L; qt0 += QT_ROWS) {
            __syncthreads();
            stage_tile_nt<D, QT_ROWS, QT_PITCH, true>(q, ldq, (int64_t)s0, qt0, L, hq * D, qs, qt);
            stage_tile_nt<D, QT_ROWS, QT_PITCH, true>(dout, lddo, (int64_t)s0, qt0, L, hq * D, dos, dot);
            if (threadIdx.x < QT_ROWS) {
                const int qi = qt0 + threadIdx.x;
                s_lse[threadIdx.x] = qi < L ? lse[(int64_t)hq * T + s0 + qi] * LOG2E : 0.f;
                s_dlt[threadIdx.x] = qi < L ? delta[(int64_t)hq * T + s0 + qi] : 0.f;
            }
            __syncthreads();
            if (CAUSAL && qt0 + QT_ROWS - 1 < k_base + wave * 32) continue;      // whole q tile is before this wave's keys
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
            const char* qp = qs + kc * C::KPITCH + half * 16;     // A operand rows = q_local = lane&31
            const char* dp = dos + kc * C::KPITCH + half * 16;
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(qp + s * 32);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(dp + s * 32);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[s], sacc, 0, 0, 0);      // S  = Q K^T   (rows q, cols key)
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, vf[s], pacc, 0, 0, 0);      // dP = dO V^T
            }
            float pv[16], dsv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int qi = qt0 + ql;
                const bool ok = k_ok && (qi < L) && (!CAUSAL || key_idx <= qi);
                const float p = ok ? exp2f(sacc[r] * scale_log2 - s_lse[ql]) : 0.f;
                pv[r] = p;
                dsv[r] = p * (pacc[r] - s_dlt[ql]) * scale;
            }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                const int rb = ks2 * 8;
                uint4 pw, dw;
                pw.x = pack2bf(pv[rb + 0], pv[rb + 1]); pw.y = pack2bf(pv[rb + 2], pv[rb + 3]);
                pw.z = pack2bf(pv[rb + 4], pv[rb + 5]); pw.w = pack2bf(pv[rb + 6], pv[rb + 7]);
                dw.x = pack2bf(dsv[rb + 0], dsv[rb + 1]); dw.y = pack2bf(dsv[rb + 2], dsv[rb + 3]);
                dw.z = pack2bf(dsv[rb + 4], dsv[rb + 5]); dw.w = pack2bf(dsv[rb + 6], dsv[rb + 7]);
                const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
                const bf16x8 df = *reinterpret_cast<bf16x8*>(&dw);
                const int q0 = ks2 * 16 + 4 * half;                 // q rows q0..+3 and q0+8..+11
#pragma unroll
                for (int b = 0; b < C::DP / 32; ++b) {
                    const char* t1 = dot + (b * 32 + kc) * QT_PITCH + q0 * 2;
                    const char* t2 = qt + (b * 32 + kc) * QT_PITCH + q0 * 2;
                    uint4 aw, bw;
                    uint2 lo = *reinterpret_cast<const uint2*>(t1), hi = *reinterpret_cast<const uint2*>(t1 + 16);
                    aw.x = lo.x; aw.y = lo.y; aw.z = hi.x; aw.w = hi.y;
                    lo = *reinterpret_cast<const uint2*>(t2); hi = *reinterpret_cast<const uint2*>(t2 + 16);
                    bw.x = lo.x; bw.y = lo.y; bw.z = hi.x; bw.w = hi.y;
                    dva[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&aw), pf, dva[b], 0, 0, 0);  // dV^T += dO^T P
                    dka[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&bw), df, dka[b], 0, 0, 0);  // dK^T += Q^T dS
                }
            }
        }
    }
    if (!k_ok) return;
    uint16_t* kp = dk + (int64_t)(s0 + key_idx) * lddk + (int64_t)kvh * D;
    uint16_t* vp = dv + (int64_t)(s0 + key_idx) * lddv + (int64_t)kvh * D;
#pragma unroll
    for (int b = 0; b < C::DP / 32; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            if (d < D) {
                uint2 w;
                w.x = pack2bf(dka[b][4 * g + 0], dka[b][4 * g + 1]);
                w.y = pack2bf(dka[b][4 * g + 2], dka[b][4 * g + 3]);
                *reinterpret_cast<uint2*>(kp + d) = w;
                w.x = pack2bf(dva[b][4 * g + 0], dva[b][4 * g + 1]);
                w.y = pack2bf(dva[b][4 * g + 2], dva[b][4 * g + 3]);
                *reinterpret_cast<uint2*>(vp + d) = w;
            }
        }
}

extern "C" {

static int attn_fwd_launch(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                           const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end,
                           const int32_t* o_beg, int qgroup, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                           st_bf16* out, int64_t ldo, float* lse, int max_q, int klass, st_stream_t stream) {
    if (!q || !k || !v || !q_beg || !q_end || !k_beg || !k_end || !out || !lse || n_seq <= 0 || T <= 0 || n_q <= 0 || n_kv <= 0 ||
        (n_q % n_kv) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || max_q <= 0)
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(st_cdiv(max_q, Q_TILE), n_q, n_seq);
    const float sl2 = scale * LOG2E;
    StProfScope ps(klass, s, 0.0);
#define ST_FWD(DD, CC) hipLaunchKernelGGL((attn_fwd_kernel<DD, CC>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, qgroup, T, n_q, n_kv, sl2, out, ldo, lse)
    if (D == 128 && causal) ST_FWD(128, true);
    else if (D == 128) ST_FWD(128, false);
    else if (D == 80 && !causal) ST_FWD(80, false);
    else if (D == 80) ST_FWD(80, true);
    else return ST_EINVAL;
#undef ST_FWD
    ST_CHECK_LAUNCH();
    return 0;
}

int st_attn_fwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal, st_bf16* out,
                int64_t ldo, float* lse, int max_seqlen, st_stream_t stream) {
    if (!cu_seqlens) return ST_EINVAL;
    return attn_fwd_launch(q, ldq, k, ldk, v, ldv, cu_seqlens, cu_seqlens + 1, cu_seqlens, cu_seqlens + 1, nullptr, 0, n_seq, T, n_q, n_kv, D,
                           scale, causal, out, ldo, lse, max_seqlen, D == 128 ? ST_K_ATTN_FWD : ST_K_VIT_ATTN, stream);
}

int st_attn_fwd_ranges(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                       const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end,
                       const int32_t* o_beg, int q_group, int n_seq, int T_out, int n_q, int n_kv, int D, float scale, st_bf16* out,
                       int64_t ldo, float* lse, int max_q, st_stream_t stream) {
    return attn_fwd_launch(q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, q_group, n_seq, T_out, n_q, n_kv, D, scale, 0, out, ldo,
                           lse, max_q, ST_K_DECODE_ATTN, stream);
}

int st_attn_bwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                st_bf16* dq, int64_t lddq, st_bf16* dk, int64_t lddk, st_bf16* dv, int64_t lddv, float* delta,
                int max_seqlen, st_stream_t stream) {
    if (!q || !k || !v || !out || !dout || !lse || !cu_seqlens || !dq || !dk || !dv || !delta || n_seq <= 0 || T <= 0 ||
        n_q <= 0 || n_kv <= 0 || (n_q % n_kv) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) ||
        (lddk & 3) || (lddv & 3) || max_seqlen <= 0 || (D != 128 && D != 80))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(D == 128 ? ST_K_ATTN_BWD : ST_K_VIT_ATTN, s, 0.0);
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(st_cdiv((int64_t)T * n_q, 256)), dim3(256), 0, s, out, ldo, dout, lddo, T, n_q, D, delta);
    const dim3 gq(st_cdiv(max_seqlen, Q_TILE), n_q, n_seq), gk(st_cdiv(max_seqlen, 128), n_kv, n_seq);
#define ST_BWD(DD, CC)                                                                                                         \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DD, CC>), gq, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse, delta,       \
                       cu_seqlens, T, n_q, n_kv, scale, dq, lddq);                                                             \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DD, CC>), gk, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse, delta,      \
                       cu_seqlens, T, n_q, n_kv, scale, dk, lddk, dv, lddv)
    if (D == 128 && causal) { ST_BWD(128, true); }
    else if (D == 128) { ST_BWD(128, false); }
    else if (causal) { ST_BWD(80, true); }
    else { ST_BWD(80, false); }
#undef ST_BWD
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
