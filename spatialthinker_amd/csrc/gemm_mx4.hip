// gemm_mx4.hip — the block-scaled fp8 GEMM (gemm_fp8.hip: e4m3 elements, one e8m0 scale per 32 consecutive k) on the FOUR-wave
// hand-scheduled tile of gemm_asm4.hip:   C[M,N] = dequant(Aq)[M,K] * dequant(Bq)[N,K]^T  (bf16 (+bias)(+residual), SwiGLU, fp8 or fp32 results).
//
// Why a second fp8 kernel: the 8-wave fp8 tile of gemm_fp8.hip ran 1.48 PF/s in situ next to 1.38 PF/s of the bf16 4-wave tile — the
// fp8 mode paid for its quantiser passes and gained nothing (profiles/r03_notes.md).  The 4-wave tile's K loop carries over unchanged
// when k is counted in BYTES: a 256 x 256 tile, K-tile = 128 fp8 = the same 128-byte operand rows, the same LDS image ([row][128 B],
// 16-byte chunk position c ^ (row & 7), applied on the LDS-DMA source address), the same 16 LDS-DMA copies and 32 ds_read_b128 per wave
// and K-tile — and every byte carries twice the flops.
//   * MFMA: v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, 64 cycles): a K-tile is TWO k-steps, so the fragments double-buffer
//     exactly like the bf16 tile's (the K = 128 instruction needs all 128 fragment registers of a K-tile at once: nothing could be
//     prefetched).  One wave = 128 x 128 = 4 x 4 MFMA tiles of 32 x 32, 256 accumulator registers in the AGPR half, 16 MFMAs per k-step.
//   * operand layout (tools/probes/mx_layout_probe32.hip, on MI355X): lane (r = lane & 31, g = lane >> 5) holds k = 16g .. 16g+15 of
//     its row in bytes 0..15 and k = 32+16g .. 32+16g+15 in bytes 16..31; the scale of MX block b of the k-step comes from lane r + 32b.
//     So for k-step s a lane reads the 16-byte chunks 4s+g and 4s+2+g of its row and supplies the scale of block 2s+g.
//   * scales: S[kt][row] = one dword with the 4 block scales of the row inside K-tile kt (the quantiser's format).  A tile's 2 x 256
//     dwords travel with the operands (two 256-byte LDS-DMA copies per wave and K-tile); a lane reads its rows' dwords once per k-step
//     and shifts them right by 8g: byte 0 is then the scale of block g (k-step 0) and byte 2 that of block 2+g (k-step 1), selected by
//     the instruction's op_sel.
//   * schedule per K-tile and wave (32 MFMA slots of 64 cycles; every other instruction sits behind one of them) = compile-time data
//     (Q4Sched below; default schedule 6):
//       0..7    the 16 fragment reads + 8 scale reads of k-step 1 (this tile's slot), three per slot
//       11      lgkmcnt(0) + barrier #1: every wave is done reading this tile's slot
//       11..27  the 18 LDS-DMA copies of tile t+2 into that slot (2 scale copies, then one operand copy per slot: a 64-cycle MFMA hides
//               the ~50-cycle issue that costs the bf16 tile 20 % of its K loop)
//       21      vmcnt + barrier #2: tile t+1 has landed for every wave (the copies issued so far stay in flight)
//       22..29  the 16 fragment reads + 8 scale reads of (t+1, k-step 0), three per slot
//   * epilogues (through the idle operand LDS, 16-byte row-contiguous stores): bf16 (+bias)(+residual); bf16(silu(gate)) * up for the
//     MLP's first product (SWIGLU); that result as the MX-fp8 operand of the down projection (QOUT); fp32 = / += (OUTF: weight gradients).
// Roofline: MFMA-bound, 2*M*N*K flop per launch against the 5 PF dense fp8 peak.
#include "common.h"
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int I, int N, class F> __device__ __forceinline__ void q4_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        q4_for<I + 1, N>(f);
    }
}

#define Q4_BM 256
#define Q4_BN 256
#define Q4_SLOT 65536          // one K-tile of both operands: (256 + 256) rows x 128 bytes
#define Q4_ABYTES 32768
#define Q4_SCALES 131072       // behind the two operand slots: [slot][A 256 dwords | B 256 dwords]
#define Q4_SMEM (Q4_SCALES + 4096)

// k-step 0 takes byte 0 of the (shifted) scale dwords, k-step 1 byte 2 (op_sel = low bit, op_sel_hi = high bit of the byte index)
template <int KS> __device__ __forceinline__ void q4_mfma(f32x16& acc, const i32x8& b, const i32x8& a, int sb, int sa) {
    if constexpr (KS == 0)
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(b), "v"(a), "v"(sb), "v"(sa));
    else
        asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]" : "+a"(acc) : "v"(b), "v"(a), "v"(sb), "v"(sa));
}
template <int OFF> __device__ __forceinline__ void q4_read(i32x4& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void q4_read1(int& f, uint32_t addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void q4_shift(int& x, int sh) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(sh)); }
__device__ __forceinline__ void q4_dma(uint32_t voff, i32x4 srd, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void q4_dma1(uint32_t voff, i32x4 srd, uint32_t soff) {      // 4 bytes per lane: 256 bytes per copy
    asm volatile("buffer_load_dword %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void q4_m0_set(uint32_t v) { asm volatile("s_mov_b32 m0, %0" : : "s"(v) : "memory"); }
__device__ __forceinline__ void q4_m0_next() { asm volatile("s_add_u32 m0, m0, 0x400" : : : "memory", "scc"); }
__device__ __forceinline__ void q4_barrier() { asm volatile("s_barrier" : : : "memory"); }
template <int N> __device__ __forceinline__ void q4_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void q4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ i32x8 q4_frag(const i32x4& lo, const i32x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// The K-tile schedule (MFMA slots 0..31) as compile-time data, so that variants can be timed against each other (tools/mx4_ksweep.py):
//   RD1      k-step-1 reads per slot from slot 0 on (24 reads: per 6, four fragment halves + two scale dwords)
//   BAR1     slot of lgkmcnt(0) + barrier #1 (every wave is done reading this tile's slot)
//   SDMA     slot of the two scale copies of tile t+2;  dma_slot(j): slot of operand copy j (0..15: A copies 0..7, then B copies 0..7)
//   BAR2     slot of the vmcnt wait + barrier #2 (tile t+1 has landed for every wave)
//   RD0, RD0N  first slot and reads per slot of the 24 reads of (t+1, k-step 0); the tile ends with lgkmcnt(6): B1..B3 land in the next tile
template <int V> struct Q4Sched;
template <> struct Q4Sched<0> {              // first version: reads bunched at both ends of the tile (6 and 5 per slot), copies on two of every three slots
    static constexpr int RD1 = 6, BAR1 = 6, SDMA = 7, BAR2 = 25, RD0 = 26, RD0N = 5;
    static constexpr int dma_slot(int j) { return 8 + (j >> 1) * 3 + (j & 1); }
};
template <> struct Q4Sched<2> {              // both read groups spread 3 per slot, barrier #1 at 13, copies dense from 14, barrier #2 at 23
    static constexpr int RD1 = 3, BAR1 = 13, SDMA = 13, BAR2 = 23, RD0 = 24, RD0N = 3;
    static constexpr int dma_slot(int j) { return 14 + j; }
};
// default.  By elimination (tools/mx4_ksweep.py, profiles/r04_notes.md §7) the first version lost 14 % to its reads and nothing to its copies
// or barriers: 4 waves x 24 reads in 4 slots are 576 cycles of LDS pipe against 256 cycles of MFMA, and lgkmcnt(0) at slot 6 waited for them.
// Here the reads go out 3 per slot (the LDS pipe keeps up), barrier #1 waits until they have returned anyway (slot 11), the copies follow
// one per slot (a 64-cycle MFMA hides the ~50-cycle issue), barrier #2 sits at 21 and the k-step-0 reads of the next tile run 22..29.
// Measured against schedule 0 in steady state: +14 % (L2-hot 4096 x 4096 x 8192), +3..5 % on the layer's shapes; bit-identical results.
// Nine more placements were timed (barrier #1 at 9..13, copies from 10..14, barrier #2 at 19..23, reads 2..6 per slot): within 2 % of this one.
template <> struct Q4Sched<6> {
    static constexpr int RD1 = 3, BAR1 = 11, SDMA = 11, BAR2 = 21, RD0 = 22, RD0N = 3;
    static constexpr int dma_slot(int j) { return 12 + j; }
};
#ifndef Q4_SCHED
#define Q4_SCHED 6
#endif
template <int V> __host__ __device__ constexpr int q4_copies_before_bar2() {      // copies of tile t+2 already issued at barrier #2: they stay in flight
    int n = Q4Sched<V>::SDMA < Q4Sched<V>::BAR2 ? 2 : 0;
    for (int j = 0; j < 16; ++j) n += Q4Sched<V>::dma_slot(j) < Q4Sched<V>::BAR2 ? 1 : 0;
    return n;
}
template <int V> __host__ __device__ constexpr int q4_copy_at(int sl) {          // operand copy issued at slot sl, or -1
    for (int j = 0; j < 16; ++j) if (Q4Sched<V>::dma_slot(j) == sl) return j;
    return -1;
}

// SWIGLU: B = [gate rows | up rows] (2N x K, N = output width); the B tile interleaves 32 gate rows with the 32 matching up rows, so a
// lane holds gate and up of the same output column in adjacent MFMA column tiles (ni even: gate, ni odd: up) and the epilogue writes
// bf16(silu(gate)) * up for 128 output columns per workgroup — the roundings of gemm_asm4.hip's SwiGLU epilogue (bf16 gate / up / act).
// QOUT (SwiGLU tiles): the result leaves as MX-fp8 — the operand of the down projection — instead of bf16: C = e4m3 bytes (ldc in bytes),
// SQ[tile column][row] = the scale dword of the tile's 128 output columns (= one K-tile of the next GEMM), exactly what st_mxfp8_quantize
// makes of the bf16 result (same rounding to bf16 first).
// DBG (timing experiments only, results are wrong): 1 = no LDS-DMA in the loop, 2 = no barriers, 3 = no fragment / scale reads, 4 = MFMAs only
// OUTF: 0 = bf16 result, 1 = fp32 result (C is a float pointer, ldc in floats), 2 = fp32 result accumulated into C (the weight gradients)
template <bool HAS_BIAS, bool HAS_RES, bool SWIGLU = false, bool QOUT = false, int DBG = 0, int SV = Q4_SCHED, int OUTF = 0>
__global__ __launch_bounds__(256) void gemm_mx4_kernel(const uint8_t* __restrict__ A, int64_t lda, const uint32_t* __restrict__ SA, int64_t sa_rows,
                                                      const uint8_t* __restrict__ B, int64_t ldb, const uint32_t* __restrict__ SB, int64_t sb_rows,
                                                      const uint16_t* __restrict__ bias, const uint16_t* __restrict__ res, int64_t ldr,
                                                      uint16_t* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n,
                                                      uint32_t* __restrict__ SQ, int64_t sq_rows) {
    static_assert(!QOUT || SWIGLU, "fp8 output: SwiGLU tiles only");
    static_assert(OUTF == 0 || (!SWIGLU && !HAS_BIAS && !HAS_RES), "fp32 results: plain products");
    static_assert(Q4Sched<SV>::BAR1 >= (24 + Q4Sched<SV>::RD1 - 1) / Q4Sched<SV>::RD1 - 1 && Q4Sched<SV>::SDMA >= Q4Sched<SV>::BAR1 &&
                  Q4Sched<SV>::dma_slot(0) >= Q4Sched<SV>::SDMA && Q4Sched<SV>::dma_slot(15) <= 31 && Q4Sched<SV>::RD0 > Q4Sched<SV>::BAR2 &&
                  Q4Sched<SV>::BAR2 >= 16 && (31 - Q4Sched<SV>::RD0 + 1) * Q4Sched<SV>::RD0N >= 24, "K-tile schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap + groups of 8 tile rows (as gemm_asm4.hip)
    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int m0 = (first_m + in_g % gsz) * Q4_BM;
    const int n0 = (in_g / gsz) * (SWIGLU ? Q4_BN / 2 : Q4_BN);      // SWIGLU: first OUTPUT column
    static_assert(!SWIGLU || (!HAS_BIAS && !HAS_RES), "SwiGLU tiles carry no bias / residual");

    // ---- buffer resources: base = first row of the tile (operands) / first row-dword of the tile in K-tile 0 (scales)
    auto make_srd = [&](uint64_t base) {
        i32x4 s;
        s.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        s.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu));
        s.z = (int)0xffffffffu;                                      // rows are clamped per lane: nothing to cut off
        s.w = 0x00020000;
        return s;
    };
    const i32x4 srdA = make_srd((uint64_t)(A + (int64_t)m0 * lda)), srdB = make_srd((uint64_t)(B + (int64_t)n0 * ldb));
    const i32x4 srdB2 = make_srd((uint64_t)(B + (int64_t)(n0 + (SWIGLU ? N : 0)) * ldb));       // SWIGLU: the up rows follow the N gate rows
    const i32x4 srdSA = make_srd((uint64_t)(SA + m0)), srdSB = make_srd((uint64_t)(SB + n0));
    const int rows_a = min(Q4_BM, M - m0), cols_b = min(SWIGLU ? Q4_BN / 2 : Q4_BN, N - n0);    // B rows (= output columns) this tile owns
    // source offset of operand copy j of this wave, per lane: row (lane >> 3) of the copy's 8 rows, clamped to the last valid row of the
    // tile (such rows compute values that are never stored), 16-byte chunk (lane & 7) ^ row — the swizzle, on the source
    uint32_t voffA[8], voffB[8];
    q4_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int rl = lane >> 3, ch = ((lane & 7) ^ rl) << 4;
        voffA[j] = (uint32_t)min(wave * 64 + j * 8 + rl, rows_a - 1) * (uint32_t)lda + ch;
        // SWIGLU: tile rows 64w .. 64w+31 are the gate rows of output columns 32w .., rows 64w+32 .. the up rows (copies 4..7, through srdB2)
        const int src = SWIGLU ? wave * 32 + (j & 3) * 8 + rl : wave * 64 + j * 8 + rl;
        voffB[j] = (uint32_t)min(src, cols_b - 1) * (uint32_t)ldb + ch;
    });
    // scale copies: this wave's 64 row-dwords of each operand (rows past the scale arrays clamped: never stored either)
    const uint32_t voffSA = (uint32_t)(min((int64_t)m0 + wave * 64 + lane, sa_rows - 1) - m0) * 4u;
    const uint32_t voffSB = SWIGLU ? (uint32_t)(min((int64_t)n0 + wave * 32 + (lane & 31), (int64_t)N - 1) - n0 + (lane >= 32 ? N : 0)) * 4u
                                   : (uint32_t)(min((int64_t)n0 + wave * 64 + lane, sb_rows - 1) - n0) * 4u;
    const uint32_t stepSA = (uint32_t)sa_rows * 4u, stepSB = (uint32_t)sb_rows * 4u;
    uint32_t koff = 0, soffA = 0, soffB = 0;                         // byte offsets of the K-tile the next copies fetch (SGPRs)
    const uint32_t smem32 = (uint32_t)(uintptr_t)smem;
    const uint32_t m0A = __builtin_amdgcn_readfirstlane(smem32 + wave * 8 * 1024);
    const uint32_t m0B = __builtin_amdgcn_readfirstlane(smem32 + Q4_ABYTES + wave * 8 * 1024);
    const uint32_t m0S = __builtin_amdgcn_readfirstlane(smem32 + Q4_SCALES + wave * 256);

    f32x16 acc[4][4];                                                // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / 128;
    auto advance = [&](bool go) {                                    // the K-tile the NEXT copies fetch
        koff += go ? 128u : 0u;
        soffA += go ? stepSA : 0u;
        soffB += go ? stepSB : 0u;
    };

    // ---- fragment / scale read addresses.  Row = wave tile base + i*32 + (lane & 31) (i*4096 bytes as the instruction's offset), chunk
    // (4s + 2h + g) ^ (row & 7).  The two operand slots are 64 KiB apart, the two scale areas 2 KiB: "the other slot" is an XOR; the
    // k-step-1 reads of tile t go to its slot, the k-step-0 reads of tile t+1 to the other one; all flip after every tile.
    const int frow = lane & 31, fg = lane >> 5, sw = lane & 7;
    uint32_t adA[2][2], adB[2][2];                                   // [k-step][half]; k-step 0 entries point at the OTHER slot
    q4_for<0, 4>([&](auto xc) {
        constexpr int s = decltype(xc)::value >> 1, h = decltype(xc)::value & 1;
        const int kc = (4 * s + 2 * h + fg) ^ sw;
        adA[s][h] = smem32 + (s == 0 ? Q4_SLOT : 0) + (wm * 128 + frow) * 128 + (kc << 4);
        adB[s][h] = smem32 + (s == 0 ? Q4_SLOT : 0) + Q4_ABYTES + (wn * 128 + frow) * 128 + (kc << 4);
    });
    uint32_t adSA[2], adSB[2];                                       // [k-step]
    adSA[1] = smem32 + Q4_SCALES + (wm * 128 + frow) * 4;
    adSB[1] = smem32 + Q4_SCALES + 1024 + (wn * 128 + frow) * 4;
    adSA[0] = adSA[1] + 2048;
    adSB[0] = adSB[1] + 2048;
    const int shv = 8 * fg;

    i32x4 al[2][4], ah[2][4], bl[2][4], bh[2][4];                    // fragment halves: [k-step][tile]
    int sc[2][8];                                                    // [k-step][A tiles 0..3, B tiles 0..3]: scale dword >> 8g

    auto dma_tile = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < 8) q4_dma(voffA[j], srdA, koff);
        else if constexpr (SWIGLU && j >= 12) q4_dma(voffB[j - 8], srdB2, koff);
        else q4_dma(voffB[j - 8], srdB, koff);
    };
    // read r of a k-step: 0..7 A fragment halves (tile r >> 1, half r & 1), 8..15 B; scale read r: 0..3 A tiles, 4..7 B tiles
    auto rd_frag = [&](auto rc, auto sc_) {
        constexpr int r = decltype(rc)::value, s = decltype(sc_)::value;
        constexpr int i = (r & 7) >> 1, h = r & 1;
        if constexpr (r < 8) { if constexpr (h == 0) q4_read<i * 4096>(al[s][i], adA[s][0]); else q4_read<i * 4096>(ah[s][i], adA[s][1]); }
        else { if constexpr (h == 0) q4_read<i * 4096>(bl[s][i], adB[s][0]); else q4_read<i * 4096>(bh[s][i], adB[s][1]); }
    };
    auto rd_scale = [&](auto rc, auto sc_) {
        constexpr int r = decltype(rc)::value, s = decltype(sc_)::value;
        if constexpr (r < 4) q4_read1<r * 128>(sc[s][r], adSA[s]); else q4_read1<(r - 4) * 128>(sc[s][r], adSB[s]);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    // the 24 reads of (t+1, k-step 0) in order of need: A0..A3, B0, the scales, B1..B3
    auto rd_next = [&](auto xc) {
        constexpr int x = decltype(xc)::value;
        if constexpr (x < 10) rd_frag(std::integral_constant<int, x>{}, K0{});
        else if constexpr (x < 18) rd_scale(std::integral_constant<int, (x - 10)>{}, K0{});
        else rd_frag(std::integral_constant<int, (x - 8)>{}, K0{});
    };
    uint32_t m0A_cur = m0A, m0B_cur = m0B, m0S_cur = m0S;            // LDS-DMA destinations of the tile being refilled (slot of tile t)

    auto issue_tile = [&](uint32_t mS, uint32_t mA, uint32_t mB) {   // prologue: all 18 copies of one tile back to back
        q4_m0_set(mS);
        q4_dma1(voffSA, srdSA, soffA);
        q4_m0_set(mS + 1024);
        q4_dma1(voffSB, srdSB, soffB);
        q4_m0_set(mA);
        q4_for<0, 16>([&](auto jc) {
            if constexpr (decltype(jc)::value == 8) q4_m0_set(mB);
            dma_tile(jc);
            q4_m0_next();
        });
    };
    // ---- prologue: tiles 0 and 1 in flight, fragments + scales of (0, k-step 0) in registers
    issue_tile(m0S, m0A, m0B);
    advance(nk > 1);
    issue_tile(m0S + 2048, m0A + Q4_SLOT, m0B + Q4_SLOT);           // (nk == 1: tile 0 once more — keeps the loop's counted waits uniform)
    advance(nk > 2);
    q4_wait_vm<18>();
    q4_barrier();
    {                                                                // (0, k-step 0) sits in slot 0 = "the other slot" of the flipped addresses
        adA[0][0] ^= Q4_SLOT; adA[0][1] ^= Q4_SLOT; adB[0][0] ^= Q4_SLOT; adB[0][1] ^= Q4_SLOT; adSA[0] ^= 2048; adSB[0] ^= 2048;
        q4_for<0, 24>([&](auto xc) { rd_next(xc); });
        adA[0][0] ^= Q4_SLOT; adA[0][1] ^= Q4_SLOT; adB[0][0] ^= Q4_SLOT; adB[0][1] ^= Q4_SLOT; adSA[0] ^= 2048; adSB[0] ^= 2048;
        q4_wait_lgkm<6>();
        q4_for<0, 8>([&](auto ic) { q4_shift(sc[0][decltype(ic)::value], shv); });
    }

    for (int kt = 0; kt < nk; ++kt) {
        q4_for<0, 32>([&](auto ic) {
            constexpr int sl = decltype(ic)::value;
            constexpr int ks = sl >> 4, idx = sl & 15, ni = idx >> 2, mi = idx & 3;
            q4_mfma<ks>(acc[ni][mi], q4_frag(bl[ks][ni], bh[ks][ni]), q4_frag(al[ks][mi], ah[ks][mi]), sc[ks][4 + ni], sc[ks][mi]);
            constexpr bool RD = DBG != 3 && DBG != 4, BAR = DBG != 2 && DBG != 4, DMA = DBG != 1 && DBG != 4;
            using S = Q4Sched<SV>;
            if constexpr (RD && sl * S::RD1 < 24) {                  // k-step 1 of this tile: per 6 reads, 4 fragment halves + 2 scale dwords
                q4_for<sl * S::RD1, (sl * S::RD1 + S::RD1 < 24 ? sl * S::RD1 + S::RD1 : 24)>([&](auto xc) {
                    constexpr int x = decltype(xc)::value, g = x / 6, w6 = x % 6;
                    if constexpr (w6 < 4) rd_frag(std::integral_constant<int, (g * 4 + w6)>{}, K1{});
                    else rd_scale(std::integral_constant<int, (g * 2 + w6 - 4)>{}, K1{});
                });
            }
            // B1..B3 of k-step 0 were read LAST in the previous tile (the six oldest of the <= 30 reads outstanding): needed from slots 4, 8, 12
            if constexpr ((sl == 3 || sl == 7 || sl == 11) && sl < S::BAR1) q4_wait_lgkm<15>();
            if constexpr (sl == S::BAR1) {
                q4_wait_lgkm<0>();
                if constexpr (BAR) q4_barrier();
                if constexpr (RD) q4_for<0, 8>([&](auto ic2) { q4_shift(sc[1][decltype(ic2)::value], shv); });
            }
            if constexpr (DMA && sl == S::SDMA) {
                q4_m0_set(m0S_cur);
                q4_dma1(voffSA, srdSA, soffA);
                q4_m0_set(m0S_cur + 1024);
                q4_dma1(voffSB, srdSB, soffB);
                q4_m0_set(m0A_cur);
            }
            if constexpr (DMA && q4_copy_at<SV>(sl) >= 0) {
                constexpr int j = q4_copy_at<SV>(sl);
                dma_tile(std::integral_constant<int, j>{});
                if constexpr (j == 7) q4_m0_set(m0B_cur); else if constexpr (j < 15) q4_m0_next();
            }
            if constexpr (sl == S::BAR2) { if constexpr (DMA) q4_wait_vm<q4_copies_before_bar2<SV>()>(); if constexpr (BAR) q4_barrier(); }
            if constexpr (RD && sl >= S::RD0 && (sl - S::RD0) * S::RD0N < 24) {
                constexpr int x0 = (sl - S::RD0) * S::RD0N;
                q4_for<x0, (x0 + S::RD0N < 24 ? x0 + S::RD0N : 24)>([&](auto xc) { rd_next(xc); });
            }
            if constexpr (sl == 31) {
                q4_wait_lgkm<6>();
                if constexpr (RD) q4_for<0, 8>([&](auto ic2) { q4_shift(sc[0][decltype(ic2)::value], shv); });
            }
        });
        // flip the slots; advance the source of the next copies (the last two tiles re-fetch tile nk-1: lands in a slot nobody reads)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            adA[s][0] ^= Q4_SLOT; adA[s][1] ^= Q4_SLOT; adB[s][0] ^= Q4_SLOT; adB[s][1] ^= Q4_SLOT;
            adSA[s] ^= 2048; adSB[s] ^= 2048;
        }
        m0A_cur ^= Q4_SLOT; m0B_cur ^= Q4_SLOT; m0S_cur ^= 2048;
        advance(kt + 3 < nk);
    }
    q4_wait_vm<0>();                                                 // the two re-fetched tiles are still landing
    q4_wait_lgkm<0>();
    // MFMA results -> epilogue reads: the hazard checker cannot see into asm; every accumulator passes THROUGH a wait
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[ni][0]), "+a"(acc[ni][1]), "+a"(acc[ni][2]), "+a"(acc[ni][3]));

    // ---- epilogue through LDS: 2 passes of 256 rows x 128 columns of fp32 (rows padded to 528 bytes), as gemm_asm4.hip.
    // Swapped operands (B first): lane holds m = mi*32 + (lane & 31) and, per register group q, n = ni*32 + 8q + 4g + 0..3
    constexpr int ROWB = 128 * 4 + 16;
    const bool interior = m0 + Q4_BM <= M && n0 + (SWIGLU ? Q4_BN / 2 : Q4_BN) <= N && (ldc & (OUTF ? 3 : 7)) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                          (!HAS_BIAS || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                          (!HAS_RES || ((ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    __syncthreads();                                                 // every wave is done with the operand slots; no DMA in flight
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = wm * 128 + mi * 32 + frow, col = wn * 64 + nl * 32 + 8 * q + 4 * fg;
                    const f32x16& a = acc[p * 2 + nl][mi];
                    *reinterpret_cast<f32x4*>(smem + row * ROWB + col * 4) = (f32x4){a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                }
        __syncthreads();
        const int t = threadIdx.x;
        if constexpr (SWIGLU) {
            // image columns: wave wn at 64*wn = [gate 32 | up 32] of output columns (2*wn + p)*32 ..; a thread handles 8 output columns of a row
            const int c8 = (t & 7) * 8, wn_ = c8 >> 5, lcol = wn_ * 64 + (c8 & 31);
            const int n = n0 + (wn_ * 2 + p) * 32 + (c8 & 31);
#pragma unroll 2
            for (int it = 0; it < 8; ++it) {
                const int row = it * 32 + (t >> 3), m = m0 + row;
                if constexpr (QOUT) {
                    // N % 128 == 0 (the next GEMM's K): no column edge; rows past M compute on clamped operands and store nothing.
                    // The 4 threads of an MX block (32 output columns) are 4 consecutive lanes: t & 3.
                    float g[8], u[8], o[8];
                    *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4);
                    *reinterpret_cast<float4*>(g + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4 + 16);
                    *reinterpret_cast<float4*>(u) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 32) * 4);
                    *reinterpret_cast<float4*>(u + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 32) * 4 + 16);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        g[r] = bfround(g[r]); u[r] = bfround(u[r]);
                        o[r] = bfround(bfround(g[r] * sigmoidf_(g[r])) * u[r]);
                    }
                    uint32_t wq[2];
                    const int e = mx_quant8(o, wq);
                    if (m < M) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(C) + (int64_t)m * ldc + n) = make_uint2(wq[0], wq[1]);
                        if ((t & 3) == 0) reinterpret_cast<uint8_t*>(SQ + (int64_t)(n0 >> 7) * sq_rows + m)[wn_ * 2 + p] = (uint8_t)e;
                    }
                    continue;
                }
                if (m >= M || n >= N) continue;
                float g[8], u[8], o[8];
                *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4);
                *reinterpret_cast<float4*>(g + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4 + 16);
                *reinterpret_cast<float4*>(u) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 32) * 4);
                *reinterpret_cast<float4*>(u + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 32) * 4 + 16);
#pragma unroll
                for (int r = 0; r < 8; ++r) {                        // same roundings as the bf16 path: bf16 gate / up, bf16 act
                    g[r] = bfround(g[r]); u[r] = bfround(u[r]);
                    o[r] = bfround(g[r] * sigmoidf_(g[r])) * u[r];
                }
                uint16_t* cp = C + (int64_t)m * ldc + n;
                if (interior || (n + 7 < N && (reinterpret_cast<uintptr_t>(cp) & 15) == 0)) *reinterpret_cast<uint4*>(cp) = pack8(o);
                else for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
            }
            if (p == 0) __syncthreads();
            continue;
        }
        const int c8 = (t & 15) * 8;
        const int n = n0 + (c8 >> 6) * 128 + p * 64 + (c8 & 63);
        if constexpr (OUTF != 0) {
            float* Cf = reinterpret_cast<float*>(C);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 16 + (t >> 4), m = m0 + row;
                if (m >= M || n >= N) continue;
                float4 v0 = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4);
                float4 v1 = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4 + 16);
                float* cp = Cf + (int64_t)m * ldc + n;
                if (interior || (n + 7 < N && (reinterpret_cast<uintptr_t>(cp) & 15) == 0)) {
                    if constexpr (OUTF == 2) {
                        const float4 o0 = *reinterpret_cast<const float4*>(cp), o1 = *reinterpret_cast<const float4*>(cp + 4);
                        v0.x += o0.x; v0.y += o0.y; v0.z += o0.z; v0.w += o0.w; v1.x += o1.x; v1.y += o1.y; v1.z += o1.z; v1.w += o1.w;
                    }
                    *reinterpret_cast<float4*>(cp) = v0;
                    *reinterpret_cast<float4*>(cp + 4) = v1;
                } else {
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = (OUTF == 2 ? cp[r] : 0.f) + v[r];
                }
            }
            if (p == 0) __syncthreads();
            continue;
        }
        if (interior) {
            float bvals[8];
            if constexpr (HAS_BIAS) unpack8(*reinterpret_cast<const uint4*>(bias + n), bvals);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 lo[8], hi[8];
                uint4 rr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (h * 8 + i) * 16 + (t >> 4);
                    lo[i] = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4);
                    hi[i] = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4 + 16);
                    if constexpr (HAS_RES) rr[i] = *reinterpret_cast<const uint4*>(res + (int64_t)(m0 + row) * ldr + n);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (h * 8 + i) * 16 + (t >> 4);
                    float v[8] = {lo[i].x, lo[i].y, lo[i].z, lo[i].w, hi[i].x, hi[i].y, hi[i].z, hi[i].w};
                    if constexpr (HAS_BIAS) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += bvals[r];
                    }
                    if constexpr (HAS_RES) {
                        float r8[8];
                        unpack8(rr[i], r8);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += r8[r];
                    }
                    *reinterpret_cast<uint4*>(C + (int64_t)(m0 + row) * ldc + n) = pack8(v);
                }
            }
        } else {
            float bvals[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool ncols = n + 7 < N;
            if constexpr (HAS_BIAS) {
                for (int r = 0; r < 8; ++r) bvals[r] = n + r < N ? bf2f(bias[n + r]) : 0.f;
            }
#pragma unroll 1
            for (int it = 0; it < 16; ++it) {
                const int row = it * 16 + (t >> 4), m = m0 + row;
                if (m >= M || n >= N) continue;
                float v[8];
                *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4);
                *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4 + 16);
                if constexpr (HAS_BIAS) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bvals[r];
                }
                if constexpr (HAS_RES) {
                    const uint16_t* rp = res + (int64_t)m * ldr + n;
                    for (int r = 0; r < 8; ++r) v[r] += n + r < N ? bf2f(rp[r]) : 0.f;
                }
                uint16_t* cp = C + (int64_t)m * ldc + n;
                if (ncols && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) *reinterpret_cast<uint4*>(cp) = pack8(v);
                else for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
            }
        }
        if (p == 0) __syncthreads();                                 // the second pass overwrites the image
    }
}

// called by st_gemm_mxfp8_nt (gemm_fp8.hip) after its argument checks; dbg = 1..4: the timing-experiment instantiations (wrong results)
int st_launch_gemm_mx4(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                       int64_t sb_rows, const uint16_t* bias, const uint16_t* residual, int64_t ldr, uint16_t* out, int64_t ldc, int M, int N,
                       int K, int dbg, hipStream_t s) {
    const int tiles_m = st_cdiv(M, Q4_BM), tiles_n = st_cdiv(N, Q4_BN);
#define Q4GO(HB, HR, D) Q4GOS(HB, HR, D, Q4_SCHED)
#define Q4GOS(HB, HR, D, SVV)                                                                                                     \
    do {                                                                                                                          \
        auto kern = gemm_mx4_kernel<HB, HR, false, false, D, SVV>;                                                                \
        static bool configured = false;                                                                                           \
        if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_SMEM); configured = true; } \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Q4_SMEM, s, A, lda, SA, sa_rows, B, ldb, SB, sb_rows, bias, residual, \
                           ldr, out, ldc, M, N, K, tiles_m, tiles_n, (uint32_t*)nullptr, (int64_t)0);                             \
    } while (0)
    if (dbg == 1) Q4GO(false, false, 1); else if (dbg == 2) Q4GO(false, false, 2); else if (dbg == 3) Q4GO(false, false, 3);
    else if (dbg == 4) Q4GO(false, false, 4);
    else if (dbg == 10) Q4GOS(false, false, 0, 0); else if (dbg == 12) Q4GOS(false, false, 0, 2); else if (dbg == 16) Q4GOS(false, false, 0, 6);
    else if (bias && residual) Q4GO(true, true, 0); else if (bias) Q4GO(true, false, 0); else if (residual) Q4GO(false, true, 0); else Q4GO(false, false, 0);
#undef Q4GO
#undef Q4GOS
    ST_CHECK_LAUNCH();
    return 0;
}

// out[M, N] = bf16(silu(gate)) * up with [gate | up] = dequant(A) dequant(B)^T, B = (2N, K): the SwiGLU tile (no-grad passes);
// sq != nullptr: out is the MX-fp8 quantisation of that result (bytes, ldc in bytes) with its scale dwords in sq
int st_launch_gemm_mx4_swiglu(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                              int64_t sb_rows, void* out, int64_t ldc, uint32_t* sq, int64_t sq_rows, int M, int N, int K, hipStream_t s) {
    const int tiles_m = st_cdiv(M, Q4_BM), tiles_n = st_cdiv(N, Q4_BN / 2);
#define Q4SW(QO)                                                                                                                  \
    do {                                                                                                                          \
        auto kern = gemm_mx4_kernel<false, false, true, QO>;                                                                      \
        static bool configured = false;                                                                                           \
        if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_SMEM); configured = true; } \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Q4_SMEM, s, A, lda, SA, sa_rows, B, ldb, SB, sb_rows,        \
                           (const uint16_t*)nullptr, (const uint16_t*)nullptr, (int64_t)0, (uint16_t*)out, ldc, M, N, K, tiles_m, tiles_n, sq, sq_rows); \
    } while (0)
    if (sq) Q4SW(true); else Q4SW(false);
#undef Q4SW
    ST_CHECK_LAUNCH();
    return 0;
}

// out[M,N] (fp32) = or += dequant(A) dequant(B)^T: the weight-gradient products (A = dY^T, B = X^T, both from st_mxfp8_quantize_t)
int st_launch_gemm_mx4_f32(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                           int64_t sb_rows, float* out, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s) {
    const int tiles_m = st_cdiv(M, Q4_BM), tiles_n = st_cdiv(N, Q4_BN);
#define Q4F(OF)                                                                                                                   \
    do {                                                                                                                          \
        auto kern = gemm_mx4_kernel<false, false, false, false, 0, Q4_SCHED, OF>;                                                 \
        static bool configured = false;                                                                                           \
        if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_SMEM); configured = true; } \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Q4_SMEM, s, A, lda, SA, sa_rows, B, ldb, SB, sb_rows, (const uint16_t*)nullptr, \
                           (const uint16_t*)nullptr, (int64_t)0, reinterpret_cast<uint16_t*>(out), ldc, M, N, K, tiles_m, tiles_n, (uint32_t*)nullptr, (int64_t)0); \
    } while (0)
    if (accumulate) Q4F(2); else Q4F(1);
#undef Q4F
    ST_CHECK_LAUNCH();
    return 0;
}
