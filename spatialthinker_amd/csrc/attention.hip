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
#include <stdlib.h>

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define KV_TILE 64
#define Q_TILE 128
#define QT_ROWS 32
#define VT_PITCH 136            // bytes per V^T row (64 keys * 2 B + 8 pad)

template <int D> struct AttnCfg {
    static constexpr int DP = (D + 31) / 32 * 32;       // PV output rows padded to MFMA 32
    static constexpr int KS = D / 16;                   // QK^T k-steps
    static constexpr int KPITCH = D * 2 + 16;           // bytes per K row in LDS
    static constexpr int K_BYTES = KV_TILE * KPITCH;
    static constexpr int V_BYTES = DP * VT_PITCH;
};

__device__ __forceinline__ uint32_t pack2bf(float a, float b) { return f2bf2(a, b); }

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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
// D = 128 forward, asynchronous staging (the LM path: training/prefill and the decode partials).
// Same MFMA mapping as attn_fwd_kernel (S^T = K Q^T, P^T fed from the accumulator registers), but the K/V tiles are
// copied HBM -> LDS by LDS-DMA (no register pass), double-buffered one tile ahead behind ONE barrier per tile:
//   K image  [64 keys][256 B]; 16-byte chunk c of row r sits at chunk position c ^ (r & 15): the 16 rows a ds_read_b128
//            lane group touches land on 16 distinct bank slots;
//   V image  32 sub-tiles of [8 keys][32 d] (512 B, 64-B rows), sub-tile (key>>3)*4 + (d>>5): the PV operand V^T is read
//            with ds_read_b64_tr_b16 — each 32-lane half of the instruction covers 4 keys x 32 d = one 256-B bank row —
//            so V is never transposed by stores (the old kernel issued 8 ds_write_b16 per 16 bytes).
// The swizzles live in the SOURCE addresses of the DMA (the LDS side of global_load_lds is lane-linear).
// =============================================================================================
// Eight transpose reads of one 16-key k-slot (4 d-blocks x 2 key groups) as ONE asm statement: hipcc puts `s_waitcnt
// vmcnt(0)` in front of the builtin form (it cannot tell the read from the LDS-DMA prefetch in flight), which would drain
// the next tile's DMA in the middle of every tile.  The results are valid only after tr_wait8 on the same registers.
__device__ __forceinline__ void tr_issue8(uint2 (&f)[8], uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                 "ds_read_b64_tr_b16 %1, %8 offset:2048\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:512\n\t"
                 "ds_read_b64_tr_b16 %3, %8 offset:2560\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:1024\n\t"
                 "ds_read_b64_tr_b16 %5, %8 offset:3072\n\t"
                 "ds_read_b64_tr_b16 %6, %8 offset:1536\n\t"
                 "ds_read_b64_tr_b16 %7, %8 offset:3584"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ void tr_wait8(uint2 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                 :
                 : "memory");
}
__device__ __forceinline__ bf16x8 tr_frag(const uint2& lo, const uint2& hi) {
    uint4 w; w.x = lo.x; w.y = lo.y; w.z = hi.x; w.w = hi.y;
    return *reinterpret_cast<bf16x8*>(&w);
}

#define F2_STAGE 32768
#define F2_V_OFF 16384
template <bool CAUSAL, bool NT = false>
__global__ __launch_bounds__(256, 2) void attn_fwd128_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                            const uint16_t* __restrict__ k, int64_t ldk,
                                                            const uint16_t* __restrict__ v, int64_t ldv,
                                                            const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end,
                                                            const int32_t* __restrict__ k_beg, const int32_t* __restrict__ k_end,
                                                            const int32_t* __restrict__ o_beg, int qgroup, int T, int n_q, int n_kv,
                                                            float scale_log2, uint16_t* __restrict__ out, int64_t ldo,
                                                            float* __restrict__ lse, const int32_t* __restrict__ pre_beg,
                                                            const int32_t* __restrict__ pre_end,
                                                            const uint16_t* __restrict__ k_pre, int64_t ldk_pre,
                                                            const uint16_t* __restrict__ v_pre, int64_t ldv_pre) {
    constexpr int D = 128;
#ifdef FWD_ONE_PER_CU
    __shared__ __attribute__((aligned(1024))) char smem[2 * F2_STAGE + 32768];      // experiment: one workgroup per CU
#else
    __shared__ __attribute__((aligned(1024))) char smem[2 * F2_STAGE];
#endif
    const int seq = blockIdx.z, h = blockIdx.y;
    const int s0 = q_beg[seq], Lq = q_end[seq] - s0;
    const int sk = k_beg[seq], L = k_end[seq] - sk;
    // causal work grows with the q tile index: launch the heavy tiles first
    const int q_base = (CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * Q_TILE;
    if (q_base >= Lq) return;
    const int kvh = h / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qc = lane & 31, half = lane >> 5;
    const int q_idx = q_base + wave * 32 + qc;
    const bool q_ok = q_idx < Lq;
    // An item without keys (decode: the chunks of a sample's cache beyond its current length — three of four items of the generated
    // partials for most of a rollout) leaves lse = -inf, which is all st_attn_merge looks at, and is gone before it touches Q: the
    // launch used to spend most of its time on ~2.5 us of Q loads and zero stores per empty workgroup, 16 of them per CU.
    if ((CAUSAL ? min(L, q_base + Q_TILE) : L) <= 0 && (!pre_beg || pre_end[seq] - pre_beg[seq] <= 0)) {
        if (q_ok && half == 0) lse[(int64_t)h * T + (o_beg ? o_beg[seq] : s0) + q_idx] = -INFINITY;
        return;
    }

    f32x16 o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;

    // keys = an optional shared PREFIX range (rows [pre_beg, pre_end) of k/v: the prompt the G rollouts of a group have in common,
    // fully visible) followed by the sequence's own range (causal).  Tiles 0..n_pre-1 walk the prefix, the rest the own keys.
    const int pb = pre_beg ? pre_beg[seq] : 0, Lp = pre_beg ? pre_end[seq] - pb : 0;
    const int n_pre = (Lp + KV_TILE - 1) / KV_TILE;
    const int kv_end = CAUSAL ? min(L, q_base + Q_TILE) : L;
    const int n_tiles = n_pre + (kv_end + KV_TILE - 1) / KV_TILE;
    // the prefix may live in other tensors than the own keys (k_pre / v_pre: the prompt K/V cache the generator filled)
    const uint16_t* kbase = k + kvh * D;
    const uint16_t* vbase = v + kvh * D;
    const uint16_t* kpbase = (k_pre ? k_pre : k) + kvh * D;
    const uint16_t* vpbase = (v_pre ? v_pre : v) + kvh * D;
    const int64_t ldkp = k_pre ? ldk_pre : ldk, ldvp = v_pre ? ldv_pre : ldv;

    // NT (decode launches): every K/V byte is read by this workgroup alone, once per decode iteration — the non-temporal copy
    auto cp16 = [](const void* src, char* dst_) __attribute__((always_inline)) { if constexpr (NT) st_glds16_nt(src, dst_); else st_glds16(src, dst_); };
    auto stage = [&](int t, char* dst) {
        const bool pre = t < n_pre;
        const int kt0 = (pre ? t : t - n_pre) * KV_TILE, row0 = pre ? pb : sk, Lc = pre ? Lp : L;
        const uint16_t* kb_ = pre ? kpbase : kbase;
        const uint16_t* vb_ = pre ? vpbase : vbase;
        const int64_t ldk_ = pre ? ldkp : ldk, ldv_ = pre ? ldvp : ldv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // K: instruction = 4 rows x 16 chunks
            const int inst = wave * 4 + j;
            const int row = inst * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (row & 15);
            int key = kt0 + row; key = key < Lc ? key : Lc - 1;
            cp16(kb_ + (int64_t)(row0 + key) * ldk_ + c * 8, dst + inst * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // V: instruction = 2 sub-tiles of [8 keys][32 d]
            const int inst = wave * 4 + j;
            const int u = 2 * inst + (lane >> 5), slot = lane & 31;
            int key = kt0 + (u >> 2) * 8 + (slot >> 2); key = key < Lc ? key : Lc - 1;
            cp16(vb_ + (int64_t)(row0 + key) * ldv_ + (u & 3) * 32 + (slot & 3) * 8, dst + F2_V_OFF + inst * 1024);
        }
    };
    // Full tiles of the own keys take the cheap path: source = wave-uniform tile base + a lane-constant 32-bit offset per copy
    // (the general path spends ~16 integer VALU instructions per copy on clamping and 64-bit address arithmetic, ~130 per tile,
    // and plain VALU work does not hide under MFMAs on this part — tools/probes/mfma_valu_overlap.hip)
    uint32_t koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int row = inst * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        koff[j] = (uint32_t)(row * (int)ldk + c * 8) * 2u;
        const int u = 2 * inst + (lane >> 5), slot = lane & 31;
        voff[j] = (uint32_t)(((u >> 2) * 8 + (slot >> 2)) * (int)ldv + (u & 3) * 32 + (slot & 3) * 8) * 2u;
    }
    auto stage_any = [&](int t, char* dst) {
        const bool pre = t < n_pre;
        const int kt0 = (pre ? t : t - n_pre) * KV_TILE;
        if (pre || kt0 + KV_TILE > L) { stage(t, dst); return; }
        const char* kt = reinterpret_cast<const char*>(kbase + (int64_t)(sk + kt0) * ldk);
        const char* vt = reinterpret_cast<const char*>(vbase + (int64_t)(sk + kt0) * ldv);
#pragma unroll
        for (int j = 0; j < 4; ++j) cp16(kt + koff[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) cp16(vt + voff[j], dst + F2_V_OFF + (wave * 4 + j) * 1024);
    };
    if (n_tiles > 0) stage_any(0, smem);
    // Q after the first tile's copies are on their way: the two latencies overlap (a decode item is ~5 us of prologue + 3.3 us per tile)
    bf16x8 qf[8];
    {
        const int64_t R = s0 + (q_ok ? q_idx : 0);
        const uint16_t* qp = qgroup > 0 ? q + (R / qgroup) * ldq + ((int64_t)h * qgroup + R % qgroup) * D + half * 8
                                        : q + R * ldq + (int64_t)h * D + half * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint4 r = q_ok ? *reinterpret_cast<const uint4*>(qp + s * 16) : make_uint4(0, 0, 0, 0);
            qf[s] = *reinterpret_cast<bf16x8*>(&r);
        }
    }
#ifdef FWD_SKEW
    {   // experiment: start the two co-resident waves of a SIMD half a tile period apart (attention)
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid >> FWD_SKEW_BIT) & 1) __builtin_amdgcn_s_sleep(FWD_SKEW);
    }
#endif

    // lane-constant LDS offsets
    const int k_row_off = qc * 256, k_swz = qc & 15;                       // rows kb*32+qc: (row & 15) == (qc & 15)
    const int v_lane_off = (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;

    for (int t = 0; t < n_tiles; ++t) {
        const bool pre = t < n_pre;
        const int kt0 = (pre ? t : t - n_pre) * KV_TILE, Lc = pre ? Lp : L;
        const bool causal_t = CAUSAL && !pre;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < n_tiles) stage_any(t + 1, smem + ((t + 1) & 1) * F2_STAGE);
        if (causal_t && kt0 > q_base + wave * 32 + 31) continue;        // tile entirely above this wave's diagonal
        if (q_base + wave * 32 >= Lq) continue;                         // decode: most waves of a tile hold no query row at all
        const char* ks = smem + (t & 1) * F2_STAGE;
        const uint32_t vaddr = smem_lds + (t & 1) * F2_STAGE + F2_V_OFF + v_lane_off;
        uint2 va[8], vb[8];
        tr_issue8(va, vaddr);                               // k-slot 0 of V^T: lands while S^T and the softmax run

        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            const char* kp = ks + kb * 32 * 256 + k_row_off;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + (((2 * s + half) ^ k_swz) << 4));
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
        // mask only where the tile meets the diagonal or the end of the keys (wave-uniform test)
        if ((kt0 + KV_TILE > Lc) || (causal_t && kt0 + KV_TILE - 1 > q_base + wave * 32)) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const bool ok = (key < Lc) && (!causal_t || key <= q_idx);
                    sacc[kb][r] = ok ? sacc[kb][r] : -INFINITY;
                }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
        mx = st_half_max(mx) * scale_log2;                  // scale_log2 > 0: max commutes with the scaling
        const float m_new = fmaxf(m_i, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);          // m_i = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], scale_log2, -m_use));
                sacc[kb][r] = p;
                rs += p;
            }
        rs = st_half_sum(rs);
        l_i = l_i * alpha + rs;
        m_i = m_new;
        if (!__all(alpha == 1.f)) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        }
        auto pv_step = [&](int ks2, uint2 (&vf)[8]) {
            const int kb = ks2 >> 1, rb = (ks2 & 1) * 8;
            uint4 pw;
            pw.x = st_pk_bf16(sacc[kb][rb + 0], sacc[kb][rb + 1]);
            pw.y = st_pk_bf16(sacc[kb][rb + 2], sacc[kb][rb + 3]);
            pw.z = st_pk_bf16(sacc[kb][rb + 4], sacc[kb][rb + 5]);
            pw.w = st_pk_bf16(sacc[kb][rb + 6], sacc[kb][rb + 7]);
            const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(vf[2 * b], vf[2 * b + 1]), pf, o[b], 0, 0, 0);
        };
        // V^T fragments ping-pong between two register sets: the reads of k-slot s+1 fly under the MFMAs of k-slot s
        tr_wait8(va);  tr_issue8(vb, vaddr + 4096);   pv_step(0, va);
        tr_wait8(vb);  tr_issue8(va, vaddr + 8192);   pv_step(1, vb);
        tr_wait8(va);  tr_issue8(vb, vaddr + 12288);  pv_step(2, va);
        tr_wait8(vb);                                 pv_step(3, vb);
    }
    if (!q_ok) return;
    const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;
    const int64_t orow = (o_beg ? o_beg[seq] : s0) + q_idx;
    uint16_t* op = out + orow * ldo + (int64_t)h * D;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            uint2 w;
            w.x = st_pk_bf16(o[b][4 * g + 0] * inv_l, o[b][4 * g + 1] * inv_l);
            w.y = st_pk_bf16(o[b][4 * g + 2] * inv_l, o[b][4 * g + 3] * inv_l);
            *reinterpret_cast<uint2*>(op + d) = w;
        }
    if (half == 0) lse[(int64_t)h * T + orow] = l_i > 0.f ? (m_i + log2f(l_i)) * LN2 : -INFINITY;
}


// =============================================================================================
// Decode attention, PERSISTENT (round 4).  The one-item-per-workgroup kernel above keeps one 32-KiB K/V tile in flight per workgroup
// (two workgroups per CU: 64 KiB per CU) and pays ~5 us of prologue per item; decode items are short (a 576-key chunk of a prompt for
// the 56 query rows of a rollout group, or a <= 512-key chunk of one sample's own cache for its 7 query rows), so the launch ran at
// ~3.5-4 TB/s with most of its time spent waiting for ONE tile (profiles/r03_decode_pmc.md).  Here ONE workgroup per CU walks a
// static list of items (w = blockIdx.x, += gridDim.x; w = item * heads + head) and streams the tiles of ALL its items through a
// single ring in the CU's whole LDS: 4 K/V slots of 32 KiB (3 tiles = 96 KiB in flight per CU while one is consumed) + 2 Q buffers
// of 16 KiB.  The producer side (all four waves issue their share of every LDS-DMA copy) runs up to 3 tiles and one item ahead of
// the consumer, so the first tiles — and the Q rows — of the next item are already landing while the current item is finished:
// no per-item prologue, no empty workgroups (items without keys are skipped by the walk, their lse set to -inf as st_attn_merge expects).
// Waits are counted: every wave keeps, for the units it has issued and not yet consumed, the number of copies per unit (8 per K/V
// tile, +4 when the unit also carries the item's Q rows) and waits with s_waitcnt vmcnt(copies issued AFTER the unit it is about to read).
// Same arithmetic as attn_fwd128_kernel<false> tile for tile (same S^T / softmax / PV sequence): bit-identical partials.
// Needs max_q <= 64 query rows per item (two row blocks); larger groups (G = 16 rollouts x 7 heads) stay on the kernel above.
// =============================================================================================
#define DEC_QBYTES 16384
// two shapes of the persistent kernel: <4 slots, 2 Q buffers, 1 workgroup per CU> (the first version: 3 tiles in flight per CU, ONE computing
// wave per tile step) and <2 slots, 1 Q buffer, 2 workgroups per CU> (80 KiB each = the CU's whole LDS: one tile in flight per workgroup as
// in the one-item-per-workgroup kernel, two computing waves per CU, no per-item prologue; the next item's Q may be staged once every wave has
// read the current item's Q into registers, i.e. from the item's second tile on)
#define DEC_LDS_BYTES_OF(SLOTS, QBUFS) ((SLOTS) * F2_STAGE + (QBUFS) * DEC_QBYTES)

template <int N> __device__ __forceinline__ void dec_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
}
__device__ __forceinline__ void dec_wait_vm_dyn(int n) {
    switch (n) {
        case 0: dec_wait_vm<0>(); break;    case 4: dec_wait_vm<4>(); break;    case 8: dec_wait_vm<8>(); break;
        case 12: dec_wait_vm<12>(); break;  case 16: dec_wait_vm<16>(); break;  case 20: dec_wait_vm<20>(); break;
        case 24: dec_wait_vm<24>(); break;  case 28: dec_wait_vm<28>(); break;  case 32: dec_wait_vm<32>(); break;
        case 36: dec_wait_vm<36>(); break;
        default: dec_wait_vm<0>(); break;
    }
}

struct DecItem {          // one (item, head) of the launch; every field is wave-uniform
    int seq, h, s0, Lq, sk, L, pb, Lp, n_pre, n_tiles, orow;
};
struct DecArgs {          // the launch arguments the helpers need (passed by value: stays in SGPRs)
    const uint16_t *q, *k, *v, *k_pre, *v_pre;
    int64_t ldq, ldk, ldv, ldkp, ldvp;
    const int32_t *q_beg, *q_end, *k_beg, *k_end, *o_beg, *pre_beg, *pre_end;
    float* lse;
    int qgroup, T, n_heads, W, G;
};

// (free functions, not lambdas: closures that capture other closures ended up in scratch memory, and a scratch access is a VMEM
// operation — it would sit in the same in-order queue as the counted LDS-DMA copies and drain it at every use)
__device__ __forceinline__ void dec_load_item(const DecArgs& a, int w, DecItem& it) {
    const int seq = __builtin_amdgcn_readfirstlane(w / a.n_heads);
    it.seq = seq; it.h = __builtin_amdgcn_readfirstlane(w - seq * a.n_heads);
    it.s0 = __builtin_amdgcn_readfirstlane(a.q_beg[seq]);
    it.Lq = __builtin_amdgcn_readfirstlane(a.q_end[seq]) - it.s0;
    it.sk = __builtin_amdgcn_readfirstlane(a.k_beg[seq]);
    it.L = max(0, __builtin_amdgcn_readfirstlane(a.k_end[seq]) - it.sk);
    it.pb = a.pre_beg ? __builtin_amdgcn_readfirstlane(a.pre_beg[seq]) : 0;
    it.Lp = a.pre_beg ? max(0, __builtin_amdgcn_readfirstlane(a.pre_end[seq]) - it.pb) : 0;
    it.n_pre = (it.Lp + KV_TILE - 1) / KV_TILE;
    it.n_tiles = it.Lq > 0 ? it.n_pre + (it.L + KV_TILE - 1) / KV_TILE : 0;
    it.orow = a.o_beg ? __builtin_amdgcn_readfirstlane(a.o_beg[seq]) : it.s0;
}
// next item WITH keys at or after work index w (stride G); items without keys get lse = -inf when `mark` (producer walk only)
__device__ __forceinline__ bool dec_next_item(const DecArgs& a, int& w, DecItem& it, bool mark) {
    while (w < a.W) {
        dec_load_item(a, w, it);
        w += a.G;
        if (it.n_tiles > 0) return true;
        if (mark && (int)threadIdx.x < it.Lq) a.lse[(int64_t)it.h * a.T + it.orow + threadIdx.x] = -INFINITY;
    }
    return false;
}
// LDS-DMA staging (the copy shapes of attn_fwd128_kernel: K = 4 rows x 16 chunks per instruction, V = 2 sub-tiles of [8 keys][32 d])
__device__ __forceinline__ void dec_stage_kv(const DecArgs& a, const DecItem& it, int t, char* dst, int wave, int lane, const uint32_t (&koff)[4],
                                             const uint32_t (&voff)[4]) {
    constexpr int D = 128;
    const bool pre = t < it.n_pre;
    const int kt0 = (pre ? t : t - it.n_pre) * KV_TILE;
    const int kvh = it.h;                                         // decode launches pass n_q == n_kv: the head IS the KV head
    if (!pre && kt0 + KV_TILE <= it.L) {                          // full tile of the own keys: uniform base + lane-constant offsets
        const char* kt = reinterpret_cast<const char*>(a.k + kvh * D + (int64_t)(it.sk + kt0) * a.ldk);
        const char* vt = reinterpret_cast<const char*>(a.v + kvh * D + (int64_t)(it.sk + kt0) * a.ldv);
#pragma unroll
        for (int j = 0; j < 4; ++j) st_glds16(kt + koff[j], dst + (wave * 4 + j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) st_glds16(vt + voff[j], dst + F2_V_OFF + (wave * 4 + j) * 1024);
        return;
    }
    const int row0 = pre ? it.pb : it.sk, Lc = pre ? it.Lp : it.L;
    const uint16_t* kb_ = (pre ? a.k_pre : a.k) + kvh * D;
    const uint16_t* vb_ = (pre ? a.v_pre : a.v) + kvh * D;
    const int64_t ldk_ = pre ? a.ldkp : a.ldk, ldv_ = pre ? a.ldvp : a.ldv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int row = inst * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);
        int key = kt0 + row; key = key < Lc ? key : Lc - 1;
        st_glds16(kb_ + (int64_t)(row0 + key) * ldk_ + c * 8, dst + inst * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int u = 2 * inst + (lane >> 5), slot = lane & 31;
        int key = kt0 + (u >> 2) * 8 + (slot >> 2); key = key < Lc ? key : Lc - 1;
        st_glds16(vb_ + (int64_t)(row0 + key) * ldv_ + (u & 3) * 32 + (slot & 3) * 8, dst + F2_V_OFF + inst * 1024);
    }
}
// the item's Q rows as a 64-row K-shaped image (chunk position = chunk ^ (row & 15)): rows beyond Lq repeat the last one
__device__ __forceinline__ void dec_stage_q(const DecArgs& a, const DecItem& it, char* dst, int wave, int lane) {
    constexpr int D = 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int row = inst * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (row & 15);
        const int64_t R = it.s0 + min(row, it.Lq - 1);
        const uint16_t* qp = a.qgroup > 0 ? a.q + (R / a.qgroup) * a.ldq + ((int64_t)it.h * a.qgroup + R % a.qgroup) * D
                                          : a.q + R * a.ldq + (int64_t)it.h * D;
        st_glds16(qp + c * 8, dst + inst * 1024);
    }
}

template <int DEC_SLOTS, int DEC_QBUFS, int DEC_MINB>
__global__ __launch_bounds__(256, DEC_MINB) void attn_decode128_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                               const uint16_t* __restrict__ k, int64_t ldk,
                                                               const uint16_t* __restrict__ v, int64_t ldv,
                                                               const int32_t* __restrict__ q_beg, const int32_t* __restrict__ q_end,
                                                               const int32_t* __restrict__ k_beg, const int32_t* __restrict__ k_end,
                                                               const int32_t* __restrict__ o_beg, int qgroup, int T, int n_items, int n_heads,
                                                               float scale_log2, uint16_t* __restrict__ out, int64_t ldo,
                                                               float* __restrict__ lse, const int32_t* __restrict__ pre_beg,
                                                               const int32_t* __restrict__ pre_end,
                                                               const uint16_t* __restrict__ k_pre, int64_t ldk_pre,
                                                               const uint16_t* __restrict__ v_pre, int64_t ldv_pre) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    char* const qbuf = dsm + DEC_SLOTS * F2_STAGE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qc = lane & 31, half = lane >> 5;
    DecArgs a;
    a.q = q; a.k = k; a.v = v; a.k_pre = k_pre ? k_pre : k; a.v_pre = v_pre ? v_pre : v;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldkp = k_pre ? ldk_pre : ldk; a.ldvp = v_pre ? ldv_pre : ldv;
    a.q_beg = q_beg; a.q_end = q_end; a.k_beg = k_beg; a.k_end = k_end; a.o_beg = o_beg; a.pre_beg = pre_beg; a.pre_end = pre_end;
    a.lse = lse; a.qgroup = qgroup; a.T = T; a.n_heads = n_heads; a.W = n_items * n_heads; a.G = gridDim.x;

    uint32_t koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int row = inst * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        koff[j] = (uint32_t)(row * (int)ldk + c * 8) * 2u;
        const int u = 2 * inst + (lane >> 5), slot = lane & 31;
        voff[j] = (uint32_t)(((u >> 2) * 8 + (slot >> 2)) * (int)ldv + (u & 3) * 32 + (slot & 3) * 8) * 2u;
    }

    // ---- producer / consumer state
    int p_w = blockIdx.x, c_w = blockIdx.x;
    DecItem pit, cit;
    bool p_open = dec_next_item(a, p_w, pit, true);
    bool c_open = dec_next_item(a, c_w, cit, false);
    int p_tile = 0, p_seq = 0, c_tile = 0, c_seq = 0;
    int produced = 0, consumed = 0;
    uint64_t fifo = 0;                                             // copies per unit (8 bits each), oldest unit in the low byte
    // one stream unit = one K/V tile (8 copies per wave) + the item's Q rows in front of its first tile (4 more); the producer
    // stays within the item after the consumer's (the Q double buffer) and within DEC_SLOTS - 1 units of it (the K/V ring)
#define DEC_PRODUCE_WHILE(LIMIT)                                                                                                     \
    while (produced - consumed < (LIMIT) && p_open &&                                                                               \
           (DEC_QBUFS == 2 ? p_seq <= c_seq + 1                                                                                      \
                           : (p_seq == c_seq || (p_seq == c_seq + 1 && (p_tile > 0 || c_tile >= 1))))) {                            \
        int cnt_ = 8;                                                                                                               \
        if (p_tile == 0) { dec_stage_q(a, pit, qbuf + (p_seq & (DEC_QBUFS - 1)) * DEC_QBYTES, wave, lane); cnt_ = 12; }             \
        dec_stage_kv(a, pit, p_tile, dsm + (produced & (DEC_SLOTS - 1)) * F2_STAGE, wave, lane, koff, voff);                        \
        fifo |= (uint64_t)cnt_ << (8 * (produced - consumed));                                                                      \
        ++produced;                                                                                                                 \
        if (++p_tile == pit.n_tiles) { p_open = dec_next_item(a, p_w, pit, true); p_tile = 0; ++p_seq; }                            \
    }
    DEC_PRODUCE_WHILE(DEC_SLOTS - 1)

    const int k_row_off = qc * 256, k_swz = qc & 15;
    const int v_lane_off = (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)dsm;
    bf16x8 qf[8];
    f32x16 o[4];
    float m_i = -INFINITY, l_i = 0.f;

    while (c_open) {
        // copies this wave issued AFTER the unit it is about to read may stay in flight
        int after = 0;
        for (int i = 1; i < produced - consumed; ++i) after += (int)((fifo >> (8 * i)) & 0xff);
        dec_wait_vm_dyn(after);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fifo >>= 8;
        const int slot = consumed & (DEC_SLOTS - 1);
        ++consumed;                                                // unit p may enter slot p & 3 once unit p - 4 has been read: p <= c + 3
        DEC_PRODUCE_WHILE(DEC_SLOTS - 1)
        const bool active = wave * 32 < cit.Lq;                    // decode: most waves of an item hold no query row at all
        const int q_idx = wave * 32 + qc;
        const bool q_ok = q_idx < cit.Lq;
        if (c_tile == 0 && active) {
            const char* qp = qbuf + (c_seq & (DEC_QBUFS - 1)) * DEC_QBYTES + (wave * 32 + qc) * 256;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) qf[s2] = *reinterpret_cast<const bf16x8*>(qp + (((2 * s2 + half) ^ k_swz) << 4));
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
            m_i = -INFINITY; l_i = 0.f;
        }
        if (active) {
            const bool pre = c_tile < cit.n_pre;
            const int kt0 = (pre ? c_tile : c_tile - cit.n_pre) * KV_TILE, Lc = pre ? cit.Lp : cit.L;
            const char* ks = dsm + slot * F2_STAGE;
            const uint32_t vaddr = smem_lds + slot * F2_STAGE + F2_V_OFF + v_lane_off;
            uint2 va[8], vb[8];
            tr_issue8(va, vaddr);
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
                const char* kp = ks + kb * 32 * 256 + k_row_off;
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + (((2 * s2 + half) ^ k_swz) << 4));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s2], sacc[kb], 0, 0, 0);
                }
            }
            if (kt0 + KV_TILE > Lc) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        sacc[kb][r] = key < Lc ? sacc[kb][r] : -INFINITY;
                    }
            }
            float mx = sacc[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
            mx = st_half_max(mx) * scale_log2;
            const float m_new = fmaxf(m_i, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], scale_log2, -m_use));
                    sacc[kb][r] = pr;
                    rs += pr;
                }
            rs = st_half_sum(rs);
            l_i = l_i * alpha + rs;
            m_i = m_new;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
#define DEC_PV_STEP(KS2, VF)                                                                                                         \
            {                                                                                                                       \
                constexpr int kb_ = (KS2) >> 1, rb_ = ((KS2) & 1) * 8;                                                              \
                uint4 pw;                                                                                                           \
                pw.x = st_pk_bf16(sacc[kb_][rb_ + 0], sacc[kb_][rb_ + 1]);                                                          \
                pw.y = st_pk_bf16(sacc[kb_][rb_ + 2], sacc[kb_][rb_ + 3]);                                                          \
                pw.z = st_pk_bf16(sacc[kb_][rb_ + 4], sacc[kb_][rb_ + 5]);                                                          \
                pw.w = st_pk_bf16(sacc[kb_][rb_ + 6], sacc[kb_][rb_ + 7]);                                                          \
                const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);                                                                  \
                _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                                       \
                    o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(VF[2 * b], VF[2 * b + 1]), pf, o[b], 0, 0, 0);           \
            }
            tr_wait8(va);  tr_issue8(vb, vaddr + 4096);   DEC_PV_STEP(0, va)
            tr_wait8(vb);  tr_issue8(va, vaddr + 8192);   DEC_PV_STEP(1, vb)
            tr_wait8(va);  tr_issue8(vb, vaddr + 12288);  DEC_PV_STEP(2, va)
            tr_wait8(vb);                                 DEC_PV_STEP(3, vb)
#undef DEC_PV_STEP
        }
        if (++c_tile == cit.n_tiles) {                            // item finished: normalise and store, then move on
            if (active && q_ok) {
                const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;
                const int64_t orow = cit.orow + q_idx;
                uint16_t* op = out + orow * ldo + (int64_t)cit.h * D;
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) {
                        const int d = b * 32 + 8 * g2 + 4 * half;
                        uint2 wv;
                        wv.x = st_pk_bf16(o[b][4 * g2 + 0] * inv_l, o[b][4 * g2 + 1] * inv_l);
                        wv.y = st_pk_bf16(o[b][4 * g2 + 2] * inv_l, o[b][4 * g2 + 3] * inv_l);
                        *reinterpret_cast<uint2*>(op + d) = wv;
                    }
                if (half == 0) lse[(int64_t)cit.h * T + orow] = l_i > 0.f ? (m_i + log2f(l_i)) * LN2 : -INFINITY;
            }
            c_open = dec_next_item(a, c_w, cit, false);
            c_tile = 0; ++c_seq;
        }
    }
#undef DEC_PRODUCE_WHILE
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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

// =============================================================================================
// D = 128 backward with asynchronous staging (same decomposition as the generic kernels below; LM path).
// Streamed tiles ([rows][128] bf16, 256-B rows) are needed BOTH as natural MFMA operands (ds_read_b128 of 8 contiguous d)
// AND transposed (ds_read_b64_tr_b16, contraction over the row index), so they use ONE dual-use image: 16-byte chunk c
// of row r sits at chunk position c ^ swz16(r), swz16(r) = (r&3)<<2 | (r>>2)&3.  The 16 rows of a ds_read_b128 lane group
// have 16 distinct (r & 15) -> 16 distinct slots; the 4 rows x 4 chunks a 32-lane half of a transpose read touches differ
// in the upper two chunk bits by (r&3) -> 16 distinct slots as well.  Filled by LDS-DMA with the permutation applied to
// the source column, double/triple buffered, one barrier per tile.
// =============================================================================================
__device__ __forceinline__ int swz16(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int ROWS>
__device__ __forceinline__ void stage_dual(const uint16_t* __restrict__ base, int64_t ld, int row0, int L, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < ROWS / 16; ++j) {                   // ROWS/4 one-KiB instructions over 4 waves
        const int inst = wave * (ROWS / 16) + j;
        const int row = inst * 4 + (lane >> 4);
        const int c = (lane & 15) ^ swz16(row);
        int g = row0 + row; g = g < L ? g : L - 1;
        st_glds16(base + (int64_t)g * ld + c * 8, dst + inst * 1024);
    }
}

// Full tiles: copy `inst` (rows 4*inst .. 4*inst+3) reads from the wave-uniform address of its first row + a lane part that needs
// two lane-constant registers and two VALU instructions: row r0 = lane >> 4 of the four, 16-byte chunk (lane & 15) ^ swz16(row) =
// cx ^ (inst & 3) with cx = (lane & 15) ^ (r0 << 2).  (The general path below clamps rows and builds 64-bit addresses: ~16 integer VALU
// instructions per copy, ~130 per tile — and plain VALU work does not hide under MFMAs: tools/probes/mfma_valu_overlap.hip.)  The
// empty asm keeps hipcc from hoisting the per-copy offsets out of the tile loop into registers the kernels do not have.
struct StageLane { uint32_t cx4, r0ld; };
__device__ __forceinline__ StageLane stage_lane(int64_t ld, int lane) {
    const int r0 = lane >> 4;
    return {(uint32_t)(((lane & 15) ^ (r0 << 2)) << 4), (uint32_t)(r0 * (int)ld) * 2u};
}
template <int ROWS>
__device__ __forceinline__ void stage_dual_fast(const uint16_t* __restrict__ tile_base, int64_t ld, uint32_t cx4, uint32_t r0ld, char* dst, int wave) {
    asm volatile("" : "+v"(cx4));
#pragma unroll
    for (int j = 0; j < ROWS / 16; ++j) {
        const int inst = wave * (ROWS / 16) + j;
        const char* tb = reinterpret_cast<const char*>(tile_base + (int64_t)(inst * 4) * ld);
        st_glds16(tb + ((cx4 ^ (uint32_t)((inst & 3) << 4)) + r0ld), dst + inst * 1024);
    }
}

// lane-constant byte offsets of the transpose reads inside a dual-use image: entry [2*b + j] = d-block b (32 d), key/row
// group j (rows +0..3 / +8..11 of a 16-row k-slot; the +8 rows and the k-slot base are immediates at the call site)
__device__ __forceinline__ void tr_dual_offsets(uint32_t (&a)[8], int lane) {
    const int s = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = ((s >> 2) << 2) | ((h + 2 * j) & 3);
            const int c = 4 * b + 2 * g + ((s & 3) >> 1);
            a[2 * b + j] = (uint32_t)((4 * h + (s >> 2)) * 256 + ((c ^ f) << 4) + (s & 1) * 8);
        }
}
template <int OFF>
__device__ __forceinline__ void tr_issue8_dual(uint2 (&f)[8], const uint32_t (&a)[8], uint32_t base) {
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%16\n\t"
                 "ds_read_b64_tr_b16 %1, %9 offset:%17\n\t"
                 "ds_read_b64_tr_b16 %2, %10 offset:%16\n\t"
                 "ds_read_b64_tr_b16 %3, %11 offset:%17\n\t"
                 "ds_read_b64_tr_b16 %4, %12 offset:%16\n\t"
                 "ds_read_b64_tr_b16 %5, %13 offset:%17\n\t"
                 "ds_read_b64_tr_b16 %6, %14 offset:%16\n\t"
                 "ds_read_b64_tr_b16 %7, %15 offset:%17"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
                 : "v"(a[0] + base), "v"(a[1] + base), "v"(a[2] + base), "v"(a[3] + base), "v"(a[4] + base), "v"(a[5] + base),
                   "v"(a[6] + base), "v"(a[7] + base), "n"(OFF), "n"(OFF + 2048)
                 : "memory");
}

// Order of an S / dP loop: NPAIR (fragment read pair, MFMA pair) steps with the reads running AHEAD steps in front of their MFMAs
// (hipcc otherwise issues a step's two ds_read_b128 right in front of its MFMAs and waits lgkmcnt(0): the LDS latency is exposed
// once per step, eight times per 32 x 32 block).  RD = LDS reads per step, MM = MFMAs per step.
template <int I, int NPAIR, int AHEAD, int RD, int MM>
__device__ __forceinline__ void pin_sloop() {
    if constexpr (I == 0) __builtin_amdgcn_sched_group_barrier(0x100, RD * AHEAD, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MM, 0);
    if constexpr (I + AHEAD < NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
    if constexpr (I + 1 < NPAIR) pin_sloop<I + 1, NPAIR, AHEAD, RD, MM>();
}

#ifndef BWD_FAST_STAGE
#define BWD_FAST_STAGE 1
#endif
#ifndef SL_AHEAD
#define SL_AHEAD 2
#endif
#ifndef KV0_AHEAD
#define KV0_AHEAD 1
#endif
#ifndef KV1_AHEAD
#define KV1_AHEAD 4
#endif
#define B2_KV_STAGE 32768          // dq kernel: K image 16 KiB + V image 16 KiB
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd128_dq_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                               const uint16_t* __restrict__ k, int64_t ldk,
                                                               const uint16_t* __restrict__ v, int64_t ldv,
                                                               const uint16_t* __restrict__ dout, int64_t lddo,
                                                               const uint16_t* __restrict__ out, int64_t ldo,
                                                               const float* __restrict__ lse, float* __restrict__ delta,
                                                               float* __restrict__ lse2_out,
                                                               const int32_t* __restrict__ seg_b, const int32_t* __restrict__ seg_e,
                                                               const int32_t* __restrict__ pre_b, const int32_t* __restrict__ pre_e,
                                                               int T, int n_q, int n_kv,
                                                               float scale, uint16_t* __restrict__ dq, int64_t lddq) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(1024))) char smem[2 * B2_KV_STAGE];
    const int seq = blockIdx.z, h = blockIdx.y;
    const int s0 = seg_b[seq], L = seg_e[seq] - s0;
    const int q_base = (CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * Q_TILE;
    if (q_base >= L) return;
    const int kvh = h / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qc = lane & 31, half = lane >> 5;
    const int q_idx = q_base + wave * 32 + qc;
    const bool q_ok = q_idx < L;
    const float scale_log2 = scale * LOG2E;

    // Q, dO fragments; delta = rowsum(dO o O) is formed here (each lane holds half of its row's d) and published for the
    // dK kernel that follows on the stream
    bf16x8 qf[8], dof[8];
    float dlt = 0.f;
    {
        const int64_t row = s0 + (q_ok ? q_idx : 0);
        const uint16_t* qp = q + row * ldq + (int64_t)h * D + half * 8;
        const uint16_t* dp = dout + row * lddo + (int64_t)h * D + half * 8;
        const uint16_t* op = out + row * ldo + (int64_t)h * D + half * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint4 a = q_ok ? *reinterpret_cast<const uint4*>(qp + s * 16) : make_uint4(0, 0, 0, 0);
            uint4 b = q_ok ? *reinterpret_cast<const uint4*>(dp + s * 16) : make_uint4(0, 0, 0, 0);
            uint4 c = q_ok ? *reinterpret_cast<const uint4*>(op + s * 16) : make_uint4(0, 0, 0, 0);
            qf[s] = *reinterpret_cast<bf16x8*>(&a);
            dof[s] = *reinterpret_cast<bf16x8*>(&b);
            float x[8], y[8];
            unpack8(b, x); unpack8(c, y);
#pragma unroll
            for (int j = 0; j < 8; ++j) dlt = fmaf(x[j], y[j], dlt);
        }
    }
    dlt = st_half_sum(dlt);
    const float lse2 = q_ok ? lse[(int64_t)h * T + s0 + q_idx] * LOG2E : 0.f;
    if (q_ok && half == 0) { delta[(int64_t)h * T + s0 + q_idx] = dlt; lse2_out[(int64_t)h * T + s0 + q_idx] = lse2; }
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // keys: shared prefix rows [pre_b, pre_e) (fully visible) then the own rows (causal) — see attn_fwd128_kernel
    const int pb = pre_b ? pre_b[seq] : 0, Lp = pre_b ? pre_e[seq] - pb : 0;
    const int n_pre = (Lp + KV_TILE - 1) / KV_TILE;
    const int kv_end = CAUSAL ? min(L, q_base + Q_TILE) : L;
    const int n_tiles = n_pre + (kv_end + KV_TILE - 1) / KV_TILE;
    const StageLane sl_k = stage_lane(ldk, lane), sl_v = stage_lane(ldv, lane);
    auto stage = [&](int t, char* dst) {
        const bool pre = t < n_pre;
        const int kt0 = (pre ? t : t - n_pre) * KV_TILE, row0 = pre ? pb : s0, Lc = pre ? Lp : L;
        if (BWD_FAST_STAGE && kt0 + KV_TILE <= Lc) {
            stage_dual_fast<KV_TILE>(k + (int64_t)(row0 + kt0) * ldk + kvh * D, ldk, sl_k.cx4, sl_k.r0ld, dst, wave);
            stage_dual_fast<KV_TILE>(v + (int64_t)(row0 + kt0) * ldv + kvh * D, ldv, sl_k.cx4, sl_v.r0ld, dst + 16384, wave);
            return;
        }
        stage_dual<KV_TILE>(k + (int64_t)row0 * ldk + kvh * D, ldk, kt0, Lc, dst, wave, lane);
        stage_dual<KV_TILE>(v + (int64_t)row0 * ldv + kvh * D, ldv, kt0, Lc, dst + 16384, wave, lane);
    };
    if (n_tiles > 0) stage(0, smem);

    const int nat_row = qc * 256, nat_swz = swz16(qc);       // natural reads: rows kb*32 + qc
    uint32_t tro[8];
    tr_dual_offsets(tro, lane);
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;

    for (int t = 0; t < n_tiles; ++t) {
        const bool pre = t < n_pre;
        const int kt0 = (pre ? t : t - n_pre) * KV_TILE, Lc = pre ? Lp : L;
        const bool causal_t = CAUSAL && !pre;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < n_tiles) stage(t + 1, smem + ((t + 1) & 1) * B2_KV_STAGE);
        if (causal_t && kt0 > q_base + wave * 32 + 31) continue;
        const char* ks = smem + (t & 1) * B2_KV_STAGE;
        const char* vs = ks + 16384;
        const uint32_t kaddr = smem_lds + (t & 1) * B2_KV_STAGE;
        const bool need_mask = (kt0 + KV_TILE > Lc) || (causal_t && kt0 + KV_TILE - 1 > q_base + wave * 32);

        uint2 ta[8], tb[8];
        tr_issue8_dual<0>(ta, tro, kaddr);                               // K^T of keys 0..15
        auto block = [&](int kb, uint2 (&t0)[8], uint2 (&t1)[8], auto issue_mid, auto issue_end) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; pacc[r] = 0.f; }
            const char* kp = ks + kb * 32 * 256 + nat_row;
            const char* vp = vs + kb * 32 * 256 + nat_row;
            bf16x8 kfr[8], vfr[8];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int off = ((2 * s + half) ^ nat_swz) << 4;
                kfr[s] = *reinterpret_cast<const bf16x8*>(kp + off);
                vfr[s] = *reinterpret_cast<const bf16x8*>(vp + off);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[s], qf[s], sacc, 0, 0, 0);     // S^T  = K Q^T
                pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[s], dof[s], pacc, 0, 0, 0);    // dP^T = V dO^T
            }
            pin_sloop<0, 8, SL_AHEAD, 2, 2>();
            __builtin_amdgcn_sched_barrier(0);
            // P first (16 independent fma + exp chains), then ONE wave-uniform branch for the tiles that need a mask: with the test
            // inside the element loop hipcc wraps every element in its own exec-mask region and the exp latency is exposed 16 times
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = __builtin_amdgcn_exp2f(fmaf(sacc[r], scale_log2, -lse2));
            if (need_mask) {
                const int key0 = kt0 + kb * 32 + 4 * half;
                const int lim = causal_t ? min(Lc - 1, q_idx) : Lc - 1;                        // visible keys: key <= lim
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = (key0 + (r & 3) + 8 * (r >> 2) <= lim) ? sacc[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = sacc[r] * (pacc[r] - dlt);                  // dS^T / scale (the factor goes on dQ at the end)
            auto dq_step = [&](int rb, uint2 (&tf)[8]) {
                uint4 pw;
                pw.x = st_pk_bf16(sacc[rb + 0], sacc[rb + 1]);
                pw.y = st_pk_bf16(sacc[rb + 2], sacc[rb + 3]);
                pw.z = st_pk_bf16(sacc[rb + 4], sacc[rb + 5]);
                pw.w = st_pk_bf16(sacc[rb + 6], sacc[rb + 7]);
                const bf16x8 df = *reinterpret_cast<bf16x8*>(&pw);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tf[2 * b], tf[2 * b + 1]), df, acc[b], 0, 0, 0);   // dQ^T += K^T dS^T
            };
            tr_wait8(t0); issue_mid(); dq_step(0, t0);
            tr_wait8(t1); issue_end(); dq_step(8, t1);
        };
        block(0, ta, tb, [&] { tr_issue8_dual<4096>(tb, tro, kaddr); }, [&] { tr_issue8_dual<8192>(ta, tro, kaddr); });
        block(1, ta, tb, [&] { tr_issue8_dual<12288>(tb, tro, kaddr); }, [&] {});
    }
    if (!q_ok) return;
    uint16_t* op = dq + (int64_t)(s0 + q_idx) * lddq + (int64_t)h * D;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            uint2 w;
            w.x = st_pk_bf16(acc[b][4 * g + 0] * scale, acc[b][4 * g + 1] * scale);
            w.y = st_pk_bf16(acc[b][4 * g + 2] * scale, acc[b][4 * g + 3] * scale);
            *reinterpret_cast<uint2*>(op + d) = w;
        }
}

// dK / dV: workgroup = 128 keys x ONE query head (wave = 32 keys, K/V fragments register resident); the head's 32-row Q/dO
// tiles stream through a 3-slot LDS ring (dual-use images + the rows' lse/delta), two tiles in flight.  Holding dK^T and dV^T
// accumulators together needs > 256 registers per lane (one wave per SIMD, nothing to overlap the softmax VALU work with),
// so the work is split into two launches that each fit two waves per SIMD:
//   MODE 0: S, dP, dS -> dK^T += Q^T dS      MODE 1: S, P -> dV^T += dO^T P      (S is recomputed: 5 matmuls instead of 4)
// Each (query head, key block) writes a bf16 partial; attn_bwd128_reduce_kernel sums the query heads of a KV group in fp32
// in a fixed order (deterministic, no atomics).
template <int N> __device__ __forceinline__ void st_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

// QR = q rows per ring slot (32 or 64), NST = ring slots.  Slot = Q image (QR x 256 B) + dO image + QR lse + QR delta floats.
template <bool CAUSAL, int MODE, int QR, int NST>
__global__ __launch_bounds__(256, 2) void attn_bwd128_kv_kernel(const uint16_t* __restrict__ q, int64_t ldq,
                                                               const uint16_t* __restrict__ k, int64_t ldk,
                                                               const uint16_t* __restrict__ v, int64_t ldv,
                                                               const uint16_t* __restrict__ dout, int64_t lddo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const int32_t* __restrict__ seg_b, const int32_t* __restrict__ seg_e,
                                                               const int32_t* __restrict__ dep_e, int T, int n_q, int n_kv,
                                                               float scale, uint16_t* __restrict__ part) {
    constexpr int D = 128;
    constexpr int IMG = QR * 256, SLOT = 2 * IMG + 1024;
    constexpr int PER = 2 * (QR / 16), EXTRA = QR / 32;      // LDS-DMA instructions per wave per slot (+ wave 0: lse/delta)
    __shared__ __attribute__((aligned(1024))) char smem[NST * SLOT];
    const int seq = blockIdx.z, hq = blockIdx.y;
    // keys = rows [seg_b, seg_e) (Lk of them); queries that see them = the own rows (causal) followed, for a shared prefix
    // segment, by its dependents [seg_e, dep_e): every response row of the group, all of which see every prefix key.
    const int s0 = seg_b[seq], Lk = seg_e[seq] - s0;
    const int L = (dep_e ? dep_e[seq] : seg_e[seq]) - s0;          // length of the query stream
    const int k_base = blockIdx.x * 128;
    if (k_base >= Lk) return;
    const int kvh = hq / (n_q / n_kv);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kc = lane & 31, half = lane >> 5;
    const int key_idx = k_base + wave * 32 + kc;
    const bool k_ok = key_idx < Lk;
    const float scale_log2 = scale * LOG2E;

    bf16x8 kf[8], vf[MODE == 0 ? 8 : 1];      // MFMA B operands: n = key, k = 8 contiguous d
    {
        const int64_t row = s0 + (k_ok ? key_idx : 0);
        const uint16_t* kp = k + row * ldk + (int64_t)kvh * D + half * 8;
        const uint16_t* vp = v + row * ldv + (int64_t)kvh * D + half * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            uint4 a = k_ok ? *reinterpret_cast<const uint4*>(kp + s * 16) : make_uint4(0, 0, 0, 0);
            kf[s] = *reinterpret_cast<bf16x8*>(&a);
            if constexpr (MODE == 0) {
                uint4 b = k_ok ? *reinterpret_cast<const uint4*>(vp + s * 16) : make_uint4(0, 0, 0, 0);
                vf[s] = *reinterpret_cast<bf16x8*>(&b);
            }
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int q_start = CAUSAL ? (k_base / QR) * QR : 0;
    const int n_stage = (L - q_start + QR - 1) / QR;
    const uint16_t* qbase = q + (int64_t)s0 * ldq + hq * D;
    const uint16_t* dobase = dout + (int64_t)s0 * lddo + hq * D;
    const StageLane sl_q = stage_lane(ldq, lane), sl_do = stage_lane(lddo, lane);
    auto stage = [&](int i, char* dst) {
        const int qt0 = q_start + i * QR;
        if (BWD_FAST_STAGE && qt0 + QR <= L) {
            stage_dual_fast<QR>(qbase + (int64_t)qt0 * ldq, ldq, sl_q.cx4, sl_q.r0ld, dst, wave);
            stage_dual_fast<QR>(dobase + (int64_t)qt0 * lddo, lddo, sl_q.cx4, sl_do.r0ld, dst + IMG, wave);
        } else {
            stage_dual<QR>(qbase, ldq, qt0, L, dst, wave, lane);
            stage_dual<QR>(dobase, lddo, qt0, L, dst + IMG, wave, lane);
        }
        if (wave == 0) {                                   // 4-byte DMA: QR lse rows then QR delta rows
#pragma unroll
            for (int e = 0; e < EXTRA; ++e) {
                const int idx = e * 64 + lane;             // < 2*QR
                int qi = qt0 + (idx % QR); qi = qi < L ? qi : L - 1;
                const float* src = (idx < QR ? lse : delta) + (int64_t)hq * T + s0 + qi;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + 2 * IMG + e * 256), 4, 0, 0);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) if (i < n_stage) stage(i, smem + i * SLOT);

    const int nat_row = kc * 256, nat_swz = swz16(kc);       // natural reads: rows = q_local = lane & 31
    uint32_t tro[8];
    tr_dual_offsets(tro, lane);
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;
    constexpr int TR_IMG = MODE == 0 ? 0 : IMG;              // transposed operand: Q^T (dK) or dO^T (dV)

    int slot = 0;
    for (int i = 0; i < n_stage; ++i) {
        // tile i must have landed; with 3 slots tile i+1 (this wave's own copies) may stay in flight
        if (NST == 2 || i + 1 >= n_stage) st_wait_vmcnt<0>();
        else if (wave == 0) st_wait_vmcnt<PER + EXTRA>();
        else st_wait_vmcnt<PER>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + NST - 1 < n_stage) { int ns = slot + NST - 1; ns = ns >= NST ? ns - NST : ns; stage(i + NST - 1, smem + ns * SLOT); }
        const char* slot_p = smem + slot * SLOT;
        const uint32_t slot_a = smem_lds + slot * SLOT;
        slot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
        for (int sub = 0; sub < QR / 32; ++sub) {
            const int qt0 = q_start + i * QR + sub * 32;
            if (qt0 >= L) continue;
            if (CAUSAL && qt0 + 31 < k_base + wave * 32) continue;           // these 32 q rows are before this wave's keys
            const char* qs = slot_p + sub * 8192;
            const char* dos = qs + IMG;
            const uint32_t qaddr = slot_a + sub * 8192;
            const float* s_lse = reinterpret_cast<const float*>(slot_p + 2 * IMG) + sub * 32;
            const float* s_dlt = s_lse + QR;

            uint2 ta[8], tb[8];
            tr_issue8_dual<TR_IMG>(ta, tro, qaddr);            // transposed operand, q rows 0..15
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; if constexpr (MODE == 0) pacc[r] = 0.f; }
            bf16x8 qfr[8], dofr[MODE == 0 ? 8 : 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int off = nat_row + (((2 * s + half) ^ nat_swz) << 4);
                qfr[s] = *reinterpret_cast<const bf16x8*>(qs + off);
                if constexpr (MODE == 0) dofr[s] = *reinterpret_cast<const bf16x8*>(dos + off);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[s], kf[s], sacc, 0, 0, 0);     // S  = Q K^T   (rows q, cols key)
                if constexpr (MODE == 0) pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr[s], vf[s], pacc, 0, 0, 0);      // dP = dO V^T
            }
            if constexpr (MODE == 0) pin_sloop<0, 8, KV0_AHEAD, 2, 2>(); else pin_sloop<0, 8, KV1_AHEAD, 1, 1>();
            __builtin_amdgcn_sched_barrier(0);
            const bool need_mask = (qt0 + 32 > L) || (k_base + wave * 32 + 32 > Lk) || (CAUSAL && qt0 < k_base + wave * 32 + 31);
            // P for all 16 elements first (independent fma + exp chains; the staged lse is already in log2 units), then ONE wave-uniform
            // branch for the sub-tiles that need a mask, then dS / the bf16 packing
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(s_lse + 8 * g + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) sacc[4 * g + j] = __builtin_amdgcn_exp2f(fmaf(sacc[4 * g + j], scale_log2, -l4[j]));
            }
            if (need_mask) {                                   // visible: lo <= qi < L with lo = the key's own row (causal; dependents lie behind Lk)
                const int lo = k_ok ? (CAUSAL ? key_idx : 0) : 0x3fffffff;
                const int q0 = qt0 + 4 * half - lo;
                const uint32_t span = (uint32_t)max(L - lo, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = ((uint32_t)(q0 + 8 * (r >> 2) + (r & 3)) < span) ? sacc[r] : 0.f;
            }
            uint32_t pk[8];                                    // packed bf16 P (dV) or dS / scale (dK), rows 2i, 2i+1
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (MODE == 0) {
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(s_dlt + 8 * g + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j) sacc[4 * g + j] *= pacc[4 * g + j] - d4[j];
                }
                pk[2 * g] = st_pk_bf16(sacc[4 * g], sacc[4 * g + 1]);
                pk[2 * g + 1] = st_pk_bf16(sacc[4 * g + 2], sacc[4 * g + 3]);
            }
            auto kv_step = [&](int ks2, uint2 (&tf)[8]) {
                uint4 pw;
                pw.x = pk[4 * ks2]; pw.y = pk[4 * ks2 + 1]; pw.z = pk[4 * ks2 + 2]; pw.w = pk[4 * ks2 + 3];
                const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tf[2 * b], tf[2 * b + 1]), pf, acc[b], 0, 0, 0);
            };
            tr_wait8(ta);
            tr_issue8_dual<TR_IMG + 4096>(tb, tro, qaddr);     // q rows 16..31
            kv_step(0, ta);
            tr_wait8(tb);
            kv_step(1, tb);
        }
    }
    if (!k_ok) return;
    uint16_t* pp = part + ((int64_t)hq * T + s0 + key_idx) * D;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            const float f = MODE == 0 ? scale : 1.f;           // dS was kept without the softmax scale
            uint2 w;
            w.x = st_pk_bf16(acc[b][4 * g + 0] * f, acc[b][4 * g + 1] * f);
            w.y = st_pk_bf16(acc[b][4 * g + 2] * f, acc[b][4 * g + 3] * f);
            *reinterpret_cast<uint2*>(pp + d) = w;
        }
}

// dk[t][kvh*128 + d] = sum over the group's query heads of the bf16 partials (fp32 accumulate, fixed order); same for dv.
__global__ void attn_bwd128_reduce_kernel(const uint16_t* __restrict__ pk, const uint16_t* __restrict__ pv,
                                          const int32_t* __restrict__ t_end_ptr, int t_end_val, int T, int n_kv, int group, uint16_t* __restrict__ dk, int64_t lddk, uint16_t* __restrict__ dv,
                                          int64_t lddv) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte chunk of one (tensor, kv head, token)
    const int64_t per = (int64_t)n_kv * T * 16;
    if (idx >= 2 * per) return;
    const bool is_v = idx >= per;
    const int64_t i = is_v ? idx - per : idx;
    const int c = (int)(i & 15);
    const int64_t t = (i >> 4) % T;
    if (t >= (t_end_ptr ? *t_end_ptr : t_end_val)) return; // rows after the last sequence (padding) hold no partials: left untouched
    const int kvh = (int)((i >> 4) / T);
    const uint16_t* src = (is_v ? pv : pk) + (((int64_t)kvh * group) * T + t) * 128 + c * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < group; ++g) {
        float x[8];
        unpack8(*reinterpret_cast<const uint4*>(src + (int64_t)g * T * 128), x);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += x[j];
    }
    uint4 o;
    o.x = st_pk_bf16(a[0], a[1]); o.y = st_pk_bf16(a[2], a[3]); o.z = st_pk_bf16(a[4], a[5]); o.w = st_pk_bf16(a[6], a[7]);
    uint16_t* dst = (is_v ? dv + t * lddv : dk + t * lddk) + kvh * 128 + c * 8;
    *reinterpret_cast<uint4*>(dst) = o;
}


// ViT windows (attention_win.hip): D = 80, bidirectional, n_q == n_kv, every sequence <= 64 tokens.  ST_VIT_WIN=0 keeps the generic kernels.
int st_attn_win80_fwd_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                             const int32_t* cu, int n_seq, int T, int n_q, float scale, uint16_t* out, int64_t ldo, float* lse,
                             hipStream_t s);
int st_attn_win80_bwd_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                             const uint16_t* dout, int64_t lddo, const float* lse, const int32_t* cu, int n_seq, int T, int n_q,
                             float scale, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv,
                             float* delta, hipStream_t s);
// (plain functions, not lambdas: with a second lambda-initialised static in this file — one outside, one inside extern "C" — hipcc
// initialised g_decode_attn_persistent below with THIS initialiser's value, and every decode launch took the persistent kernel)
static bool vit_win_from_env() { const char* e = getenv("ST_VIT_WIN"); return !(e && e[0] == '0'); }
static const bool g_vit_win = vit_win_from_env();

// K/V copies of the rollout's decode partials with the non-temporal policy (round 6): every byte is read by ONE workgroup, once per decode
// iteration; same-box A/B: 10.27 -> 10.19 ms per 512-row iteration at 200-token contexts, 11.68 -> 11.45 at 700, 4.23 -> 4.18 at 64 rows
// (profiles/r06_notes.md §2).  Results unchanged (a cache policy).  ST_DECODE_ATTN_NT=0 restores the default policy.
static int decode_attn_nt_from_env() { const char* e = getenv("ST_DECODE_ATTN_NT"); return e ? atoi(e) : 1; }
int g_decode_attn_nt = decode_attn_nt_from_env();

extern "C" {

static int attn_fwd_launch(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                           const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end,
                           const int32_t* o_beg, int qgroup, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                           st_bf16* out, int64_t ldo, float* lse, int max_q, int klass, st_stream_t stream,
                           const int32_t* pre_beg = nullptr, const int32_t* pre_end = nullptr, const st_bf16* k_pre = nullptr,
                           int64_t ldk_pre = 0, const st_bf16* v_pre = nullptr, int64_t ldv_pre = 0) {
    if ((k_pre == nullptr) != (v_pre == nullptr) || (k_pre && (!pre_beg || (ldk_pre & 7) || (ldv_pre & 7)))) return ST_EINVAL;
    if (pre_beg && D != 128) return ST_EINVAL;             // shared-prefix ranges exist for the LM head dim only
    if (!q || !k || !v || !q_beg || !q_end || !k_beg || !k_end || !out || !lse || n_seq <= 0 || T <= 0 || n_q <= 0 || n_kv <= 0 ||
        (n_q % n_kv) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || max_q <= 0)
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(st_cdiv(max_q, Q_TILE), n_q, n_seq);
    const float sl2 = scale * LOG2E;
    StProfScope ps(klass, s, 0.0);
#define ST_FWD(DD, CC) hipLaunchKernelGGL((attn_fwd_kernel<DD, CC>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, qgroup, T, n_q, n_kv, sl2, out, ldo, lse)
#define ST_FWD2(CC) hipLaunchKernelGGL((attn_fwd128_kernel<CC>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, qgroup, T, n_q, n_kv, sl2, out, ldo, lse, pre_beg, pre_end, k_pre, ldk_pre, v_pre, ldv_pre)
    if (D == 128 && !causal && qgroup > 0 && g_decode_attn_nt)     // the rollout's decode partials: K/V streamed once per workgroup
        hipLaunchKernelGGL((attn_fwd128_kernel<false, true>), grid, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, qgroup, T, n_q, n_kv, sl2, out, ldo, lse, pre_beg, pre_end, k_pre, ldk_pre, v_pre, ldv_pre);
    else if (D == 128) { if (causal) ST_FWD2(true); else ST_FWD2(false); }
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
    if (g_vit_win && D == 80 && !causal && n_q == n_kv && max_seqlen > 0 && max_seqlen <= 64) {
        if (!q || !k || !v || !out || !lse || n_seq <= 0 || T <= 0 || n_q <= 0 || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return ST_EINVAL;
        StProfScope ps(ST_K_VIT_WIN, (hipStream_t)stream, 0.0);
        return st_attn_win80_fwd_launch(q, ldq, k, ldk, v, ldv, cu_seqlens, n_seq, T, n_q, scale, out, ldo, lse, (hipStream_t)stream);
    }
    return attn_fwd_launch(q, ldq, k, ldk, v, ldv, cu_seqlens, cu_seqlens + 1, cu_seqlens, cu_seqlens + 1, nullptr, 0, n_seq, T, n_q, n_kv, D,
                           scale, causal, out, ldo, lse, max_seqlen, D == 128 ? ST_K_ATTN_FWD : ST_K_VIT_ATTN, stream);
}

// default 0: measured on MI355X the persistent kernel is SLOWER than one workgroup per item (per-layer launch at 512 rows / 448-token
// generated contexts: 203 vs 144 us; 344 rows / 256: 109 vs 75 us; decode iteration 11.9 vs 10.8 ms at 512 rows — profiles/r04_notes.md):
// with one workgroup per CU a single wave computes every tile of the CU (decode items have 7 or 56 query rows) and also issues its share
// of the copies, ~2 us per tile, where two co-resident workgroups run two such waves side by side.  Kept selectable (ST_DECODE_ATTN=persistent,
// st_decode_attn_select) and bit-identical (tests/test_gpu_kernels.py).
static int decode_attn_from_env() { const char* e = getenv("ST_DECODE_ATTN"); return (e && e[0] == 'p') ? ((e[1] == '2') ? 2 : 1) : 0; }      // "persistent" / "p2"
static int g_decode_attn_persistent = decode_attn_from_env();
int st_decode_attn_select(int persistent) {          // 0 = one workgroup per item, 1 = persistent, one workgroup per CU, 2 = persistent, two per CU
    if (persistent < 0 || persistent > 2) return ST_EINVAL;
    g_decode_attn_persistent = persistent;
    return 0;
}
int64_t st_decode_attn_selected(void) { return g_decode_attn_persistent; }

int st_attn_fwd_ranges(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                       const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end,
                       const int32_t* o_beg, int q_group, int n_seq, int T_out, int n_q, int n_kv, int D, float scale, st_bf16* out,
                       int64_t ldo, float* lse, int max_q, const int32_t* pre_beg, const int32_t* pre_end, const st_bf16* k_pre,
                       int64_t ldk_pre, const st_bf16* v_pre, int64_t ldv_pre, st_stream_t stream) {
    // decode launches (n_q == n_kv heads, items of <= 64 query rows) may take the persistent one-workgroup-per-CU kernel (see the switch above)
    if (g_decode_attn_persistent && D == 128 && n_q == n_kv && max_q <= 64 && q && k && v && q_beg && q_end && k_beg && k_end && out && lse && n_seq > 0 && T_out > 0 &&
        n_q > 0 && !(ldq & 7) && !(ldk & 7) && !(ldv & 7) && !(ldo & 3) && max_q > 0 && (k_pre == nullptr) == (v_pre == nullptr) &&
        (!k_pre || (pre_beg && !(ldk_pre & 7) && !(ldv_pre & 7)))) {
        hipStream_t s = (hipStream_t)stream;
        const int64_t W = (int64_t)n_seq * n_q;
        StProfScope ps(ST_K_DECODE_ATTN, s, 0.0);
#define DEC_GO(SLOTS, QBUFS, MINB)                                                                                                    \
        do {                                                                                                                          \
            auto kern = attn_decode128_kernel<SLOTS, QBUFS, MINB>;                                                                    \
            static bool configured = false;                                                                                           \
            if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LDS_BYTES_OF(SLOTS, QBUFS)); configured = true; } \
            const int64_t slots_ = (int64_t)st_num_cus() * (MINB);                                                                     \
            hipLaunchKernelGGL(kern, dim3((int)(W < slots_ ? W : slots_)), dim3(256), DEC_LDS_BYTES_OF(SLOTS, QBUFS), s, q, ldq, k, ldk, v, ldv, q_beg, q_end, \
                               k_beg, k_end, o_beg, q_group, T_out, n_seq, n_q, scale * LOG2E, out, ldo, lse, pre_beg, pre_end, k_pre, ldk_pre, v_pre, ldv_pre); \
        } while (0)
        if (g_decode_attn_persistent == 2) DEC_GO(2, 1, 2); else DEC_GO(4, 2, 1);
#undef DEC_GO
        ST_CHECK_LAUNCH();
        return 0;
    }
    return attn_fwd_launch(q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, q_group, n_seq, T_out, n_q, n_kv, D, scale, 0, out, ldo,
                           lse, max_q, ST_K_DECODE_ATTN, stream, pre_beg, pre_end, k_pre, ldk_pre, v_pre, ldv_pre);
}

int64_t st_attn_bwd_workspace_bytes(int T, int n_q, int D) {
    return D == 128 ? (int64_t)2 * n_q * T * 128 * (int64_t)sizeof(uint16_t) + (int64_t)n_q * T * (int64_t)sizeof(float) : 0;   // dK / dV partials + lse in log2 units
}

// D = 128 backward over segments: dQ (+ delta), per-head dK and dV partials, group reduce
static int attn_bwd128_launch(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                              const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                              const int32_t* seg_b, const int32_t* seg_e, const int32_t* pre_b, const int32_t* pre_e,
                              const int32_t* dep_e, const int32_t* t_end_ptr, int t_end_val, int n_seg, int T, int n_q, int n_kv,
                              float scale, int causal, st_bf16* dq, int64_t lddq, st_bf16* dk, int64_t lddk, st_bf16* dv, int64_t lddv,
                              float* delta, void* workspace, int64_t workspace_bytes, int max_seg, hipStream_t s) {
    if (!workspace || workspace_bytes < st_attn_bwd_workspace_bytes(T, n_q, 128)) return ST_EINVAL;
    uint16_t* part_k = (uint16_t*)workspace;
    uint16_t* part_v = part_k + (int64_t)n_q * T * 128;
    float* lse2 = (float*)(part_v + (int64_t)n_q * T * 128);       // written by the dQ kernel (with delta), read by the dK / dV kernels
    const dim3 gq(st_cdiv(max_seg, Q_TILE), n_q, n_seg), gkv(st_cdiv(max_seg, 128), n_q, n_seg);
#define ST_BWD2(CC)                                                                                                            \
    hipLaunchKernelGGL((attn_bwd128_dq_kernel<CC>), gq, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, out, ldo, lse,     \
                       delta, lse2, seg_b, seg_e, pre_b, pre_e, T, n_q, n_kv, scale, dq, lddq);                                      \
    hipLaunchKernelGGL((attn_bwd128_kv_kernel<CC, 0, 64, 2>), gkv, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse2,   \
                       delta, seg_b, seg_e, dep_e, T, n_q, n_kv, scale, part_k);                                               \
    hipLaunchKernelGGL((attn_bwd128_kv_kernel<CC, 1, 64, 2>), gkv, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse2,   \
                       delta, seg_b, seg_e, dep_e, T, n_q, n_kv, scale, part_v);                                               \
    hipLaunchKernelGGL(attn_bwd128_reduce_kernel, dim3(st_cdiv((int64_t)2 * n_kv * T * 16, 256)), dim3(256), 0, s, part_k,     \
                       part_v, t_end_ptr, t_end_val, T, n_kv, n_q / n_kv, dk, lddk, dv, lddv)
    if (causal) { ST_BWD2(true); } else { ST_BWD2(false); }
#undef ST_BWD2
    ST_CHECK_LAUNCH();
    return 0;
}

int st_attn_bwd(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                const int32_t* cu_seqlens, int n_seq, int T, int n_q, int n_kv, int D, float scale, int causal,
                st_bf16* dq, int64_t lddq, st_bf16* dk, int64_t lddk, st_bf16* dv, int64_t lddv, float* delta,
                void* workspace, int64_t workspace_bytes, int max_seqlen, st_stream_t stream) {
    if (!q || !k || !v || !out || !dout || !lse || !cu_seqlens || !dq || !dk || !dv || !delta || n_seq <= 0 || T <= 0 ||
        n_q <= 0 || n_kv <= 0 || (n_q % n_kv) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) ||
        (lddk & 3) || (lddv & 3) || max_seqlen <= 0 || (D != 128 && D != 80))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vit_win = g_vit_win && D == 80 && !causal && n_q == n_kv && max_seqlen <= 64;
    StProfScope ps(D == 128 ? ST_K_ATTN_BWD : vit_win ? ST_K_VIT_WIN : ST_K_VIT_ATTN, s, 0.0);
    if (D == 128)
        return attn_bwd128_launch(q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, lse, cu_seqlens, cu_seqlens + 1, nullptr, nullptr, nullptr,
                                  cu_seqlens + n_seq, 0, n_seq, T, n_q, n_kv, scale, causal, dq, lddq, dk, lddk, dv, lddv, delta, workspace,
                                  workspace_bytes, max_seqlen, s);
    if (vit_win)
        return st_attn_win80_bwd_launch(q, ldq, k, ldk, v, ldv, dout, lddo, lse, cu_seqlens, n_seq, T, n_q, scale, dq, lddq, dk, lddk, dv, lddv,
                                        delta, s);
    const dim3 gq(st_cdiv(max_seqlen, Q_TILE), n_q, n_seq), gk(st_cdiv(max_seqlen, 128), n_kv, n_seq);
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(st_cdiv((int64_t)T * n_q, 256)), dim3(256), 0, s, out, ldo, dout, lddo, T, n_q, D, delta);
#define ST_BWD(DD, CC)                                                                                                         \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DD, CC>), gq, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse, delta,       \
                       cu_seqlens, T, n_q, n_kv, scale, dq, lddq);                                                             \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DD, CC>), gk, dim3(256), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse, delta,      \
                       cu_seqlens, T, n_q, n_kv, scale, dk, lddk, dv, lddv)
    if (causal) { ST_BWD(80, true); } else { ST_BWD(80, false); }
#undef ST_BWD
    ST_CHECK_LAUNCH();
    return 0;
}

/* ---- shared-prefix ("segment") attention of the packed GRPO micro-batch, D = 128 -------------------------------------------
 * Segment s owns rows [seg_b[s], seg_e[s]) of q/k/v.  Its queries see the prefix rows [pre_b[s], pre_e[s]) entirely (the prompt
 * that the rollouts of one group share; empty range for a prompt or a stand-alone sequence) and then their own rows causally.
 * Backward additionally needs dep_e[s] >= seg_e[s]: rows [seg_e[s], dep_e[s]) are the queries OUTSIDE the segment that see all of
 * its keys (the group's response rows, packed right behind their prompt); dep_e[s] == seg_e[s] for segments nobody depends on. */
int st_attn_fwd_seg(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                    const int32_t* seg_b, const int32_t* seg_e, const int32_t* pre_b, const int32_t* pre_e, const st_bf16* k_pre,
                    int64_t ldk_pre, const st_bf16* v_pre, int64_t ldv_pre, int n_seg, int T, int n_q,
                    int n_kv, int D, float scale, st_bf16* out, int64_t ldo, float* lse, int max_seg, st_stream_t stream) {
    if (D != 128 || !pre_b || !pre_e) return ST_EINVAL;
    return attn_fwd_launch(q, ldq, k, ldk, v, ldv, seg_b, seg_e, seg_b, seg_e, nullptr, 0, n_seg, T, n_q, n_kv, D, scale, 1, out, ldo, lse,
                           max_seg, ST_K_ATTN_FWD, stream, pre_b, pre_e, k_pre, ldk_pre, v_pre, ldv_pre);
}

int st_attn_bwd_seg(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                    const st_bf16* out, int64_t ldo, const st_bf16* dout, int64_t lddo, const float* lse,
                    const int32_t* seg_b, const int32_t* seg_e, const int32_t* pre_b, const int32_t* pre_e, const int32_t* dep_e,
                    int n_seg, int T, int T_valid, int n_q, int n_kv, int D, float scale, st_bf16* dq, int64_t lddq, st_bf16* dk,
                    int64_t lddk, st_bf16* dv, int64_t lddv, float* delta, void* workspace, int64_t workspace_bytes, int max_seg,
                    st_stream_t stream) {
    if (!q || !k || !v || !out || !dout || !lse || !seg_b || !seg_e || !pre_b || !pre_e || !dep_e || !dq || !dk || !dv || !delta ||
        n_seg <= 0 || T <= 0 || T_valid < 0 || T_valid > T || n_q <= 0 || n_kv <= 0 || (n_q % n_kv) || (ldq & 7) || (ldk & 7) ||
        (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3) || max_seg <= 0 || D != 128)
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_ATTN_BWD, s, 0.0);
    return attn_bwd128_launch(q, ldq, k, ldk, v, ldv, out, ldo, dout, lddo, lse, seg_b, seg_e, pre_b, pre_e, dep_e, nullptr, T_valid, n_seg,
                              T, n_q, n_kv, scale, 1, dq, lddq, dk, lddk, dv, lddv, delta, workspace, workspace_bytes, max_seg, s);
}

}  // extern "C"
