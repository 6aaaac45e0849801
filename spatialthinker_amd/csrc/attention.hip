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
                                                      const int32_t* __restrict__ cu, int T, int n_q, int n_kv,
                                                      float scale_log2, uint16_t* __restrict__ out, int64_t ldo,
                                                      float* __restrict__ lse) {
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) char smem[C::K_BYTES + C::V_BYTES];
    char* ks = smem;
    char* vt = smem + C::K_BYTES;

    const int seq = blockIdx.z, h = blockIdx.y;
    const int s0 = cu[seq], L = cu[seq + 1] - s0;
    const int q_base = blockIdx.x * Q_TILE;
    if (q_base >= L) return;
    const int kvh = h / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qc = lane & 31, half = lane >> 5;
    const int q_idx = q_base + wave * 32 + qc;           // row inside the sequence
    const bool q_ok = q_idx < L;

    // Q fragments (MFMA B operand: n = q row, k = 8 contiguous d), resident for the whole kernel
    bf16x8 qf[C::KS];
    {
        const uint16_t* qp = q + (int64_t)(s0 + (q_ok ? q_idx : 0)) * ldq + (int64_t)h * D + half * 8;
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
        stage_kv<D>(k, ldk, v, ldv, (int64_t)s0, kt0, L, kvh * D, ks, vt);
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
    uint16_t* op = out + (int64_t)(s0 + q_idx) * ldo + (int64_t)h * D;
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
    if (half == 0) lse[(int64_t)h * T + s0 + q_idx] = (m_i + log2f(l_i)) * LN2;
}

extern "C" {

int st_attn_fwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal, st_bf16* out,
                int64_t ldo, float* lse, int max_seqlen, st_stream_t stream) {
    if (!q || !k || !v || !cu_seqlens || !out || !lse || n_seq <= 0 || T <= 0 || n_q <= 0 || n_kv <= 0 || (n_q % n_kv) ||
        (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || max_seqlen <= 0)
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(st_cdiv(max_seqlen, Q_TILE), n_q, n_seq);
    const float sl2 = scale * LOG2E;
    StProfScope ps(D == 128 ? ST_K_ATTN_FWD : ST_K_VIT_ATTN, s, 0.0);
    if (D == 128 && causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, cu_seqlens, T, n_q, n_kv, sl2, out, ldo, lse);
    else if (D == 128) hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, cu_seqlens, T, n_q, n_kv, sl2, out, ldo, lse);
    else if (D == 80 && !causal) hipLaunchKernelGGL((attn_fwd_kernel<80, false>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, cu_seqlens, T, n_q, n_kv, sl2, out, ldo, lse);
    else if (D == 80) hipLaunchKernelGGL((attn_fwd_kernel<80, true>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, cu_seqlens, T, n_q, n_kv, sl2, out, ldo, lse);
    else return ST_EINVAL;
    ST_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
