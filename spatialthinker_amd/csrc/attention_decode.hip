// attention_decode.hip — decode attention over a sample's OWN generated keys, one WAVE per (sample, KV head) item (round 6).
//
// What it replaces: the generated-key partials of the rollout's decode step (the paged-attention decode of vLLM behind
// /root/reference verl/workers/rollout/vllm_rollout_spmd.py:115-188).  attn_fwd128_kernel<false> gave every such item a 256-thread
// workgroup with two 32-KiB K/V slots: 64 KiB of LDS = TWO workgroups per CU, each with ONE computing wave (an item has n_q / n_kv = 7
// query rows) and ~5 us of prologue per item in front of 2-8 tiles.  With 2048 live items per launch (512 rows x 4 KV heads) that is 8
// items per CU, two at a time: the launch is bound by workgroup lifetime, not by HBM (profiles/r06_notes.md: 34 us for 105 MB at 512 rows /
// 100-token contexts = 3.1 TB/s).  Here an item is ONE wave with its own ring of 32-key tiles (16 KiB each: K [32][256 B] + V 16
// sub-tiles of [8 keys][32 d]): 32 KiB of LDS per workgroup = five workgroups per CU, five tiles in flight and five computing waves per
// CU, no barriers at all (a wave waits for its own copies with a counted vmcnt), and the hardware's workgroup dispatcher balances the items.
// Same arithmetic per tile as attn_fwd128_kernel (S^T = K Q^T on MFMA 32x32x16, a lane owns one query column, P^T fed to the PV MFMA from
// the accumulator registers, V^T by ds_read_b64_tr_b16); the online softmax advances in steps of 32 keys instead of 64, so the partials
// equal the other kernel's up to fp32 rounding (not bit for bit).
// MEASURED (profiles/r06_notes.md §2): no faster than the workgroup kernel at 512 rows, 7 % slower per decode iteration at <= 64 rows; a third
// form with the tiles staged through REGISTERS (2-4 tiles = 32-64 KB in flight per wave, plain 16-byte loads, one LDS slot) was correct and
// 1-4 % slower still, i.e. neither workgroup lifetime nor bytes in flight bounds this launch: it runs at 4.4-4.9 TB/s of a ~6.3 TB/s copy rate
// whichever way the tiles travel.  Kept as the opt-in it is (ST_DECODE_ROWS=1) with its parity test; the register form was removed.
#include "common.h"

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define DW_KEYS 32
#define DW_TILE 16384
#define DW_V_OFF 8192

namespace {

// eight transpose reads of one 16-key k-slot as ONE asm statement (see attention.hip: the builtin form drags a vmcnt(0) in front of it)
__device__ __forceinline__ void dw_tr_issue8(uint2 (&f)[8], uint32_t addr) {
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
__device__ __forceinline__ void dw_tr_wait8(uint2 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                 :
                 : "memory");
}
__device__ __forceinline__ bf16x8 dw_tr_frag(const uint2& lo, const uint2& hi) {
    uint4 w; w.x = lo.x; w.y = lo.y; w.z = hi.x; w.w = hi.y;
    return *reinterpret_cast<bf16x8*>(&w);
}
template <int N> __device__ __forceinline__ void dw_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

}  // namespace

// grid (items, heads), 64 threads.  Item `seq`: query rows [q_beg, q_end) (at most 32; row R of the launch = sample R / qgroup, group-local
// head R % qgroup of KV head h, read straight from the (B, n_q*D) projection output), keys = rows [k_beg, k_end) of k / v (row pitch ldk /
// ldv, this head's 128 columns at h * 128).  An item without keys leaves lse = -inf (all st_attn_merge looks at) and touches nothing else.
template <int SLOTS>
__global__ __launch_bounds__(64, 2) void attn_decode_rows_kernel(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ k, int64_t ldk,
                                                             const uint16_t* __restrict__ v, int64_t ldv, const int32_t* __restrict__ q_beg,
                                                             const int32_t* __restrict__ q_end, const int32_t* __restrict__ k_beg,
                                                             const int32_t* __restrict__ k_end, const int32_t* __restrict__ o_beg, int qgroup,
                                                             int T, float scale_log2, uint16_t* __restrict__ out, int64_t ldo,
                                                             float* __restrict__ lse) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(1024))) char smem[SLOTS * DW_TILE];
    const int seq = blockIdx.x, h = blockIdx.y;
    const int lane = threadIdx.x;
    const int s0 = __builtin_amdgcn_readfirstlane(q_beg[seq]), Lq = __builtin_amdgcn_readfirstlane(q_end[seq]) - s0;
    const int sk = __builtin_amdgcn_readfirstlane(k_beg[seq]), L = __builtin_amdgcn_readfirstlane(k_end[seq]) - sk;
    const int orow0 = o_beg ? __builtin_amdgcn_readfirstlane(o_beg[seq]) : s0;
    if (Lq <= 0) return;
    if (L <= 0) {
        if (lane < Lq) lse[(int64_t)h * T + orow0 + lane] = -INFINITY;
        return;
    }
    const int qc = lane & 31, half = lane >> 5;
    const bool q_ok = qc < Lq;
    const int n_tiles = (L + DW_KEYS - 1) / DW_KEYS;
    const uint16_t* kbase = k + (int64_t)h * D;
    const uint16_t* vbase = v + (int64_t)h * D;

    // lane-constant source offsets of the 8 + 8 copies of a FULL tile (K: instruction j = rows 4j .. 4j+3 x 16 chunks, chunk position
    // c ^ (row & 15); V: instruction j = sub-tiles 2j, 2j+1 of [8 keys][32 d])
    uint32_t koff[8], voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = j * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        koff[j] = (uint32_t)(row * (int)ldk + c * 8) * 2u;
        const int u = 2 * j + (lane >> 5), sl = lane & 31;
        voff[j] = (uint32_t)(((u >> 2) * 8 + (sl >> 2)) * (int)ldv + (u & 3) * 32 + (sl & 3) * 8) * 2u;
    }
    auto stage = [&](int t, char* dst) {
        const int kt0 = t * DW_KEYS;
        if (kt0 + DW_KEYS <= L) {
            const char* kt = reinterpret_cast<const char*>(kbase + (int64_t)(sk + kt0) * ldk);
            const char* vt = reinterpret_cast<const char*>(vbase + (int64_t)(sk + kt0) * ldv);
#pragma unroll
            for (int j = 0; j < 8; ++j) st_glds16(kt + koff[j], dst + j * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) st_glds16(vt + voff[j], dst + DW_V_OFF + j * 1024);
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                       // the last, partial tile: keys past the end repeat the last one (masked below)
            const int row = j * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
            int key = kt0 + row; key = key < L ? key : L - 1;
            st_glds16(kbase + (int64_t)(sk + key) * ldk + c * 8, dst + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = 2 * j + (lane >> 5), sl = lane & 31;
            int key = kt0 + (u >> 2) * 8 + (sl >> 2); key = key < L ? key : L - 1;
            st_glds16(vbase + (int64_t)(sk + key) * ldv + (u & 3) * 32 + (sl & 3) * 8, dst + DW_V_OFF + j * 1024);
        }
    };
    // prologue: SLOTS - 1 tiles on their way, then Q (its latency overlaps theirs)
#pragma unroll
    for (int p = 0; p < SLOTS - 1; ++p)
        if (p < n_tiles) stage(p, smem + p * DW_TILE);
    bf16x8 qf[8];
    {
        const int64_t R = s0 + (q_ok ? qc : Lq - 1);        // rows past the item repeat its last row: finite values, never stored
        const uint16_t* qp = qgroup > 0 ? q + (R / qgroup) * ldq + ((int64_t)h * qgroup + R % qgroup) * D + half * 8
                                        : q + R * ldq + (int64_t)h * D + half * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 r = *reinterpret_cast<const uint4*>(qp + s * 16);
            qf[s] = *reinterpret_cast<const bf16x8*>(&r);
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    const int k_row_off = qc * 256, k_swz = qc & 15;
    const int v_lane_off = (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;

    int slot = 0, fill = (SLOTS - 1) % SLOTS;              // slot of tile t / slot the next staged tile goes to
    for (int t = 0; t < n_tiles; ++t) {
        // tile t + SLOTS - 1 into the slot tile t - 1 has left (its LDS reads were waited for before its last MFMAs)
        const int ahead = min(n_tiles - 1 - t, SLOTS - 1);  // tiles after t that are (or are about to be) in flight
        if (t + SLOTS - 1 < n_tiles) stage(t + SLOTS - 1, smem + fill * DW_TILE);
        fill = fill + 1 == SLOTS ? 0 : fill + 1;
        // tile t has landed when at most the copies of the `ahead` younger tiles are outstanding (16 per tile, in issue order)
        if constexpr (SLOTS == 2) { if (ahead >= 1) dw_wait_vm<16>(); else dw_wait_vm<0>(); }
        else if constexpr (SLOTS == 3) { if (ahead >= 2) dw_wait_vm<32>(); else if (ahead == 1) dw_wait_vm<16>(); else dw_wait_vm<0>(); }
        else { if (ahead >= 3) dw_wait_vm<48>(); else if (ahead == 2) dw_wait_vm<32>(); else if (ahead == 1) dw_wait_vm<16>(); else dw_wait_vm<0>(); }
        const int kt0 = t * DW_KEYS;
        const char* ks = smem + slot * DW_TILE;
        const uint32_t vaddr = smem_lds + slot * DW_TILE + DW_V_OFF + v_lane_off;
        uint2 va[8], vb[8];
        dw_tr_issue8(va, vaddr);                            // k-slot 0 of V^T lands while S^T and the softmax run
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        const char* kp = ks + k_row_off;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + (((2 * s + half) ^ k_swz) << 4));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
        }
        if (kt0 + DW_KEYS > L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                sacc[r] = key < L ? sacc[r] : -INFINITY;
            }
        }
        float mx = sacc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
        mx = st_half_max(mx) * scale_log2;
        const float m_new = fmaxf(m_i, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], scale_log2, -m_use));
            sacc[r] = p;
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
        auto pv_step = [&](int rb, uint2 (&vf)[8]) {
            uint4 pw;
            pw.x = st_pk_bf16(sacc[rb + 0], sacc[rb + 1]);
            pw.y = st_pk_bf16(sacc[rb + 2], sacc[rb + 3]);
            pw.z = st_pk_bf16(sacc[rb + 4], sacc[rb + 5]);
            pw.w = st_pk_bf16(sacc[rb + 6], sacc[rb + 7]);
            const bf16x8 pf = *reinterpret_cast<bf16x8*>(&pw);
#pragma unroll
            for (int b = 0; b < 4; ++b)
                o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dw_tr_frag(vf[2 * b], vf[2 * b + 1]), pf, o[b], 0, 0, 0);
        };
        dw_tr_wait8(va);  dw_tr_issue8(vb, vaddr + 4096);  pv_step(0, va);
        dw_tr_wait8(vb);                                   pv_step(8, vb);
        slot = slot + 1 == SLOTS ? 0 : slot + 1;
    }
    if (!q_ok) return;
    const float inv_l = l_i > 0.f ? 1.f / l_i : 0.f;
    const int64_t orow = orow0 + qc;
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

extern "C" int st_attn_decode_rows(const st_bf16* q, int64_t ldq, const st_bf16* k, int64_t ldk, const st_bf16* v, int64_t ldv,
                                   const int32_t* q_beg, const int32_t* q_end, const int32_t* k_beg, const int32_t* k_end, const int32_t* o_beg,
                                   int q_group, int n_items, int T_out, int n_heads, int D, float scale, st_bf16* out, int64_t ldo, float* lse,
                                   int max_q, int slots, st_stream_t stream) {
    if (!q || !k || !v || !q_beg || !q_end || !k_beg || !k_end || !out || !lse || n_items <= 0 || T_out <= 0 || n_heads <= 0 || D != 128 ||
        (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || max_q <= 0 || max_q > 32 || ldk >= (1 << 24) || ldv >= (1 << 24))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_DECODE_ATTN, s, 0.0);
    const dim3 grid(n_items, n_heads);
    const float sl2 = scale * LOG2E;
#define DW_GO(S) hipLaunchKernelGGL((attn_decode_rows_kernel<S>), grid, dim3(64), 0, s, q, ldq, k, ldk, v, ldv, q_beg, q_end, k_beg, k_end, o_beg, q_group, T_out, sl2, out, ldo, lse)
    if (slots <= 2) DW_GO(2); else if (slots == 3) DW_GO(3); else DW_GO(4);
#undef DW_GO
    ST_CHECK_LAUNCH();
    return 0;
}
