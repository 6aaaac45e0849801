// attention_win.hip — ViT window attention (round 5): bidirectional sequences of <= 64 tokens, head dim 80, n_q == n_kv.
// The reference runs these through flash_attn_varlen over cu_window_seqlens (HF modeling_qwen2_5_vl.py:225-291 via
// /root/reference/verl/workers/actor/dp_actor.py:118-124); 28 of the ViT's 32 layers are windowed.
//
// The generic D = 80 kernels (attention.hip) give a 64-token window a 128-row workgroup (two of four waves idle but computing),
// stage K / V through registers with 2-byte transposing LDS stores behind two barriers per tile, and split the backward into three
// launches (delta, dQ, dK/dV) that each re-stage the same 10-KiB tiles.  A window is ONE tile: here a workgroup of two waves owns one
// (window, head) pair, wave w the rows 32w..32w+31, and every byte of q / k / v / dO crosses the memory system ONCE per workgroup:
//   * each (<= 64) x 80 operand tile is copied by LDS-DMA into ONE dual-use image of [8 tokens][32 d] sub-tiles (512 B, 64-B rows;
//     sub-tile = token group * 3 + d block, 12 KiB per tile): fragments contracted over d are ds_read_b128 of 8 contiguous d of a
//     token row, fragments contracted over the TOKEN index (V^T for P V; K^T, Q^T, dO^T in the backward) are ds_read_b64_tr_b16 of the
//     same bytes — the image the D = 128 forward uses for V.  d block 2 holds d 64..79 twice (its upper half only feeds accumulator
//     rows d >= 80, which are never stored).  Four consecutive lanes copy one 64-byte piece: 192 requests per tile.  (A first version
//     read the d-contracted fragments straight from global memory, 16 bytes per lane with the lanes of an instruction in 32 different
//     rows: the same bytes, but 2 700 requests per workgroup — 211 G requests/s over the launch, the L2s' request rate, at 3.4 TB/s
//     with FETCH_SIZE already at the algorithmic 686 MB; profiles/r05_notes.md §6.)
//   * results leave the same way: the transposed accumulators (lane = token) go through LDS once and are stored as 16-byte chunks of
//     whole 160-byte rows, consecutive lanes consecutive chunks (8-byte stores from the accumulator layout: 1 280 write requests);
//   * forward: S^T = K Q^T, one-tile softmax in registers (lane = query column), O^T = V^T P^T — same MFMA order and the same
//     scalar arithmetic as attn_fwd_kernel<80, false>, so the results are bit-identical to it;
//   * backward, ONE launch: wave w computes S^T / dP^T for its query block, delta = sum_keys P dP in registers (equal to the
//     sum_d dO O of the generic path up to rounding; O is never read), dQ^T = K^T dS^T; then S / dP for its key block in the
//     untransposed layout (lane = key column, lse / delta per query row through 512 B of LDS) and dV^T = dO^T P, dK^T = Q^T dS.
// Roofline: the launch is HBM/latency-bound, not MFMA-bound (a 4-image pass reads 41 MB of q/k/v and computes 1.8 GF); it is
// reported in the ViT attention class of bench.py with the full-attention layers.
#include "common.h"

#define W_LOG2E 1.4426950408889634f
#define W_LN2 0.6931471805599453f
#define W_D 80
#define W_KS 5                  // 16-wide k-steps over d
#define W_IMG 12288             // bytes of one dual-use [64][80] image
#define W_OUT 5120              // bytes of one wave's 32 x 80 output rows on their way to global memory

// copy the (<= 64) x 80 tile whose first row is `src` (row pitch ld elements) into a transposable image; rows >= L repeat row L-1
// (finite data under zero probabilities).  NW waves share the 12 one-KiB copies.
template <int NW>
__device__ __forceinline__ void win_stage_tr(const uint16_t* __restrict__ src, int64_t ld, int L, char* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 12 / NW; ++j) {
        const int inst = wave * (12 / NW) + j;
        const int u = 2 * inst + (lane >> 5), slot = lane & 31;      // sub-tile u = token group * 3 + d block
        const int tg = u / 3, db = u - 3 * tg;
        int row = tg * 8 + (slot >> 2); row = row < L ? row : L - 1;
        int d = db * 32 + (slot & 3) * 8; d = d < W_D ? d : d - 16;
        st_glds16(src + (int64_t)row * ld + d, dst + inst * 1024);
    }
}

// the six transpose reads of one 16-token k-slot (3 d blocks x 2 token groups) as one asm statement, results valid after win_tr_wait
__device__ __forceinline__ void win_tr_issue(uint2 (&f)[6], uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %6\n\t"
                 "ds_read_b64_tr_b16 %1, %6 offset:1536\n\t"
                 "ds_read_b64_tr_b16 %2, %6 offset:512\n\t"
                 "ds_read_b64_tr_b16 %3, %6 offset:2048\n\t"
                 "ds_read_b64_tr_b16 %4, %6 offset:1024\n\t"
                 "ds_read_b64_tr_b16 %5, %6 offset:2560"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5])
                 : "v"(addr)
                 : "memory");
}
__device__ __forceinline__ void win_tr_wait(uint2 (&f)[6]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : : "memory");
}
__device__ __forceinline__ bf16x8 win_frag(const uint2& lo, const uint2& hi) {
    uint4 w; w.x = lo.x; w.y = lo.y; w.z = hi.x; w.w = hi.y;
    return *reinterpret_cast<bf16x8*>(&w);
}
// rows rb..rb+7 of a 32x32 accumulator block as the bf16 B operand of the next MFMA (k index = those rows)
__device__ __forceinline__ bf16x8 win_pack8(const f32x16& a, int rb) {
    uint4 w;
    w.x = f2bf2(a[rb + 0], a[rb + 1]); w.y = f2bf2(a[rb + 2], a[rb + 3]);
    w.z = f2bf2(a[rb + 4], a[rb + 5]); w.w = f2bf2(a[rb + 6], a[rb + 7]);
    return *reinterpret_cast<bf16x8*>(&w);
}
// the d-contracted fragment set of token row `row` of an image: d = 16 s + 8 half .. +7 (an MFMA A / B operand per k-step s)
__device__ __forceinline__ void win_lds_row(const char* img, int row, int half, bf16x8 (&f)[W_KS]) {
    const char* p = img + (row >> 3) * 1536 + (row & 7) * 64 + half * 16;
#pragma unroll
    for (int s = 0; s < W_KS; ++s) f[s] = *reinterpret_cast<const bf16x8*>(p + (s >> 1) * 512 + (s & 1) * 32);
}
// 32 x 80 results of a wave (accumulator layout: lane = token row qc, register (b, g, j) = d 32 b + 8 g + 4 half + j), scaled by `mul`,
// rounded to bf16, through `stage` (W_OUT bytes, this wave's) to rows [row0, row0 + n_rows) of dst as whole 16-byte chunks
__device__ __forceinline__ void win_store_rows(uint16_t* __restrict__ dst, int64_t ld, int n_rows, char* stage, const f32x16 (&o)[3],
                                               int lane, float mul) {
    const int qc = lane & 31, half = lane >> 5;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = b * 32 + 8 * g + 4 * half;
            if (d < W_D) {
                uint2 w;
                w.x = f2bf2(o[b][4 * g + 0] * mul, o[b][4 * g + 1] * mul);
                w.y = f2bf2(o[b][4 * g + 2] * mul, o[b][4 * g + 3] * mul);
                *reinterpret_cast<uint2*>(stage + qc * 160 + d * 2) = w;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own writes: no barrier
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int id = j * 64 + lane, row = id / 10, c = id - 10 * row;
        const uint4 x = *reinterpret_cast<const uint4*>(stage + id * 16);
        if (row < n_rows) *reinterpret_cast<uint4*>(dst + (int64_t)row * ld + c * 8) = x;
    }
}

// Workgroup -> (window, head), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, and the 16 heads of a window read
// interleaved 160-byte pieces of the same (T, 3 x 1280) rows — most 128-byte lines belong to two neighbouring heads.  Workgroups
// id = 8 j + x with the same x run on one XCD: they walk the heads of window 8 (j / n_q) + x in turn, so a window's lines are fetched
// into ONE L2 and used completely there (window-major ids fetched 1.5x the bytes).
#define WIN_ITEM(SEQ, HEAD)                                                                       \
    const int wi_slot_ = blockIdx.x >> 3;                                                         \
    const int HEAD = wi_slot_ % n_q, SEQ = (wi_slot_ / n_q) * 8 + (blockIdx.x & 7);               \
    if (SEQ >= n_seq) return

__global__ __launch_bounds__(128, 2) void attn_win80_fwd_kernel(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k,
                                                            int64_t ldk, const uint16_t* __restrict__ v, int64_t ldv,
                                                            const int32_t* __restrict__ cu, int n_seq, int T, int n_q, float scale_log2,
                                                            uint16_t* __restrict__ out, int64_t ldo, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * W_IMG];
    char* qi = smem;                       // Q image; after the scores: the two waves' output rows
    char* ki = smem + W_IMG;
    char* vi = smem + 2 * W_IMG;
    WIN_ITEM(seq, h);
    const int s0 = cu[seq], L = cu[seq + 1] - s0;
    if (L <= 0) return;
    if (L > 64) __builtin_trap();                           // the caller promised max_seqlen <= 64: fail loudly, not with partial windows
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qc = lane & 31, half = lane >> 5;
    const int col = h * W_D;
    win_stage_tr<2>(q + (int64_t)s0 * ldq + col, ldq, L, qi, wave, lane);
    win_stage_tr<2>(k + (int64_t)s0 * ldk + col, ldk, L, ki, wave, lane);
    win_stage_tr<2>(v + (int64_t)s0 * ldv + col, ldv, L, vi, wave, lane);
    const bool live = wave * 32 < L;                       // a window of <= 32 tokens is one wave's work
    const int nkb = L > 32 ? 2 : 1;
    const int q_idx = wave * 32 + qc;
    const bool q_ok = q_idx < L;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16 sacc[2];
    float l_i = 0.f, m_i = -INFINITY;
    if (live) {
        bf16x8 qf[W_KS];
        win_lds_row(qi, q_idx, half, qf);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
            if (kb < nkb) {
                bf16x8 kf[W_KS];
                win_lds_row(ki, kb * 32 + qc, half, kf);
#pragma unroll
                for (int s = 0; s < W_KS; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sacc[kb], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // both waves are done with the Q and K images
    asm volatile("" ::: "memory");
    if (!live) return;
    {
        // accumulator row r <-> key (r & 3) + 8 (r >> 2) + 4 half of the block
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float sv = key < L ? sacc[kb][r] * scale_log2 : -INFINITY;
                sacc[kb][r] = sv;
                mx = fmaxf(mx, sv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_use = (mx == -INFINITY) ? 0.f : mx;
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
        l_i = rs;
        m_i = mx;
    }
    f32x16 o[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    const uint32_t vaddr = (uint32_t)(uintptr_t)vi + (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
#pragma unroll
    for (int ks2 = 0; ks2 < 4; ++ks2) {                     // k-slot = 16 keys = token groups 2 ks2, 2 ks2 + 1
        if (ks2 >= 2 * nkb) break;
        uint2 vf[6];
        win_tr_issue(vf, vaddr + ks2 * 3072);
        const bf16x8 pf = win_pack8(sacc[ks2 >> 1], (ks2 & 1) * 8);
        win_tr_wait(vf);
#pragma unroll
        for (int b = 0; b < 3; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win_frag(vf[2 * b], vf[2 * b + 1]), pf, o[b], 0, 0, 0);
    }
    const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;        // rows >= L: finite values of repeated rows, never stored
    win_store_rows(out + (int64_t)(s0 + wave * 32) * ldo + col, ldo, L - wave * 32, qi + wave * W_OUT, o, lane, inv_l);
    if (q_ok && half == 0) lse[(int64_t)h * T + s0 + q_idx] = l_i > 0.f ? (m_i + log2f(l_i)) * W_LN2 : -INFINITY;
}

// One launch for dQ, dK, dV of a window.  LDS: the Q, K, V, dO images + lse (log2 units) and delta of the 64 query rows.
__global__ __launch_bounds__(128, 2) void attn_win80_bwd_kernel(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k,
                                                            int64_t ldk, const uint16_t* __restrict__ v, int64_t ldv,
                                                            const uint16_t* __restrict__ dout, int64_t lddo,
                                                            const float* __restrict__ lse, const int32_t* __restrict__ cu, int n_seq, int T,
                                                            int n_q, float scale, uint16_t* __restrict__ dq, int64_t lddq,
                                                            uint16_t* __restrict__ dk, int64_t lddk, uint16_t* __restrict__ dv,
                                                            int64_t lddv, float* __restrict__ delta) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * W_IMG + 512];
    char* ki = smem;                       // K: rows (S^T, S) and K^T (dQ^T = K^T dS^T); after part A: the waves' output rows
    char* vi = smem + W_IMG;               // V: rows (dP^T, dP)
    char* qi = smem + 2 * W_IMG;           // Q: rows and Q^T (dK^T = Q^T dS)
    char* doi = smem + 3 * W_IMG;          // dO: rows and dO^T (dV^T = dO^T P)
    float* s_lse = reinterpret_cast<float*>(smem + 4 * W_IMG);
    float* s_dlt = s_lse + 64;
    WIN_ITEM(seq, h);
    const int s0 = cu[seq], L = cu[seq + 1] - s0;
    if (L <= 0) return;
    if (L > 64) __builtin_trap();                           // the caller promised max_seqlen <= 64: fail loudly, not with partial windows
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qc = lane & 31, half = lane >> 5;
    const int col = h * W_D;
    win_stage_tr<2>(k + (int64_t)s0 * ldk + col, ldk, L, ki, wave, lane);
    win_stage_tr<2>(v + (int64_t)s0 * ldv + col, ldv, L, vi, wave, lane);
    win_stage_tr<2>(q + (int64_t)s0 * ldq + col, ldq, L, qi, wave, lane);
    win_stage_tr<2>(dout + (int64_t)s0 * lddo + col, lddo, L, doi, wave, lane);
    const bool live = wave * 32 < L;
    const int nb = L > 32 ? 2 : 1;                          // 32-token blocks in use
    const int idx = wave * 32 + qc;                         // this lane's query row (part A) and key row (part B)
    const bool ok = idx < L;
    const float scale_log2 = scale * W_LOG2E;
    const uint32_t lane_off = (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem + lane_off;
    const float lse2 = (live && ok) ? lse[(int64_t)h * T + s0 + idx] * W_LOG2E : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---------------- part A: query block `wave` — P^T, dP^T (lane = query column), delta, dS^T
    f32x16 sacc[2], pacc[2];
    bf16x8 kf[W_KS], vf[W_KS];                              // part B's operands: this wave's key rows
    if (live) {
        bf16x8 qf[W_KS], dof[W_KS];
        win_lds_row(qi, idx, half, qf);
        win_lds_row(doi, idx, half, dof);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[kb][r] = 0.f; pacc[kb][r] = 0.f; }
            if (kb < nb) {
                bf16x8 kr[W_KS], vr[W_KS];
                win_lds_row(ki, kb * 32 + qc, half, kr);
                win_lds_row(vi, kb * 32 + qc, half, vr);
#pragma unroll
                for (int s = 0; s < W_KS; ++s) {
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[s], qf[s], sacc[kb], 0, 0, 0);      // S^T  = K Q^T
                    pacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr[s], dof[s], pacc[kb], 0, 0, 0);     // dP^T = V dO^T
                }
            }
        }
        float dlt = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float p = (ok && key < L) ? exp2f(sacc[kb][r] * scale_log2 - lse2) : 0.f;
                sacc[kb][r] = p;
                dlt += p * pacc[kb][r];
            }
        dlt += __shfl_xor(dlt, 32, 64);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = sacc[kb][r] * (pacc[kb][r] - dlt) * scale;         // dS^T
        if (half == 0) {
            s_lse[idx] = lse2;
            s_dlt[idx] = dlt;
            if (ok) delta[(int64_t)h * T + s0 + idx] = dlt;
        }
        win_lds_row(ki, idx, half, kf);
        win_lds_row(vi, idx, half, vf);
    }
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    if (live) {
#pragma unroll
        for (int ks2 = 0; ks2 < 4; ++ks2) {
            if (ks2 >= 2 * nb) break;
            uint2 tf[6];
            win_tr_issue(tf, lds0 + ks2 * 3072);
            const bf16x8 df = win_pack8(sacc[ks2 >> 1], (ks2 & 1) * 8);
            win_tr_wait(tf);
#pragma unroll
            for (int b = 0; b < 3; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win_frag(tf[2 * b], tf[2 * b + 1]), df, acc[b], 0, 0, 0);   // dQ^T += K^T dS^T
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // lse / delta of all rows visible; the K and V images are free
    asm volatile("" ::: "memory");
    if (!live) return;
    char* stage = ki + wave * W_OUT;
    win_store_rows(dq + (int64_t)(s0 + wave * 32) * lddq + col, lddq, L - wave * 32, stage, acc, lane, 1.f);

    // ---------------- part B: key block `wave` — P, dS (lane = key column), dV^T, dK^T
    f32x16 dka[3], dva[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dka[b][r] = 0.f; dva[b][r] = 0.f; }
    for (int qb = 0; qb < nb; ++qb) {
        bf16x8 qa[W_KS], da[W_KS];
        win_lds_row(qi, qb * 32 + qc, half, qa);
        win_lds_row(doi, qb * 32 + qc, half, da);
        f32x16 sa, pa;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; pa[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < W_KS; ++s) {
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], kf[s], sa, 0, 0, 0);          // S  = Q K^T  (rows q, lane = key)
            pa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[s], vf[s], pa, 0, 0, 0);          // dP = dO V^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float p = (ok && qr < L) ? exp2f(sa[r] * scale_log2 - s_lse[qr]) : 0.f;
            sa[r] = p;
            pa[r] = p * (pa[r] - s_dlt[qr]) * scale;
        }
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            uint2 t1[6], t2[6];
            win_tr_issue(t1, lds0 + 3 * W_IMG + (qb * 2 + ks2) * 3072);        // dO^T
            win_tr_issue(t2, lds0 + 2 * W_IMG + (qb * 2 + ks2) * 3072);        // Q^T
            const bf16x8 pf = win_pack8(sa, ks2 * 8), df = win_pack8(pa, ks2 * 8);
            win_tr_wait(t1);
            win_tr_wait(t2);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                dva[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win_frag(t1[2 * b], t1[2 * b + 1]), pf, dva[b], 0, 0, 0);   // dV^T += dO^T P
                dka[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(win_frag(t2[2 * b], t2[2 * b + 1]), df, dka[b], 0, 0, 0);   // dK^T += Q^T dS
            }
        }
    }
    win_store_rows(dk + (int64_t)(s0 + wave * 32) * lddk + col, lddk, L - wave * 32, stage, dka, lane, 1.f);
    win_store_rows(dv + (int64_t)(s0 + wave * 32) * lddv + col, lddv, L - wave * 32, stage, dva, lane, 1.f);
}

// launchers used by st_attn_fwd / st_attn_bwd (attention.hip) when D == 80, bidirectional, n_q == n_kv and max_seqlen <= 64
int st_attn_win80_fwd_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                             const int32_t* cu, int n_seq, int T, int n_q, float scale, uint16_t* out, int64_t ldo, float* lse,
                             hipStream_t s) {
    hipLaunchKernelGGL(attn_win80_fwd_kernel, dim3(st_cdiv(n_seq, 8) * 8 * n_q), dim3(128), 0, s, q, ldq, k, ldk, v, ldv, cu, n_seq, T, n_q,
                       scale * W_LOG2E, out, ldo, lse);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_attn_win80_bwd_launch(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk, const uint16_t* v, int64_t ldv,
                             const uint16_t* dout, int64_t lddo, const float* lse, const int32_t* cu, int n_seq, int T, int n_q,
                             float scale, uint16_t* dq, int64_t lddq, uint16_t* dk, int64_t lddk, uint16_t* dv, int64_t lddv,
                             float* delta, hipStream_t s) {
    hipLaunchKernelGGL(attn_win80_bwd_kernel, dim3(st_cdiv(n_seq, 8) * 8 * n_q), dim3(128), 0, s, q, ldq, k, ldk, v, ldv, dout, lddo, lse, cu,
                       n_seq, T, n_q, scale, dq, lddq, dk, lddk, dv, lddv, delta);
    ST_CHECK_LAUNCH();
    return 0;
}
