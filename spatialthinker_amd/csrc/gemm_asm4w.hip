// gemm_asm4w.hip — the 4-wave hand-scheduled training tile of gemm_asm4.hip on the WIDE bf16 MFMA:  C[M,N] = A[M,K] * B[N,K]^T (+bias)
// (+residual) or bf16(silu(gate)) * up, bf16 in, fp32 accumulate, bf16 out; both operands row-major (the forward products).
//
// Why: gemm_asm4.hip issues 128 v_mfma_f32_16x16x32_bf16 per K-tile and wave, 16 cycles each; an LDS-DMA issue costs the wave ~50 cycles, so
// each of the 16 copies per K-tile leaves the matrix pipe idle for ~34 cycles — by elimination 20 % of that kernel's K loop
// (tools/gemm_ksweep.py debug; profiles/r03_notes.md).  On the 64-cycle block-scaled fp8 MFMA of gemm_mx4.hip the same copies cost nothing
// (tools/mx4_ksweep.py).  v_mfma_f32_32x32x16_bf16 is the longest bf16 MFMA (32 cycles): 64 per K-tile and wave instead of 128, the same
// LDS image, copies and fragment reads.
//   * one wave = 128 x 128 = 4 x 4 MFMA tiles of 32 x 32 (256 accumulator registers in the AGPR half); a K-tile of 64 is FOUR k-steps of 16;
//   * operand layout: lane (r = lane & 31, g = lane >> 5) holds k = 8g .. 8g+7 of its row: the 16-byte chunk 2s + g of the row for k-step s
//     (LDS image [row][128 B], chunk position c ^ (row & 7), as gemm_asm4.hip): 8 ds_read_b128 per k-step, 32 per K-tile;
//   * all four k-steps of a tile live in registers (128 VGPRs, as the two k-steps of gemm_asm4.hip): k-steps 0 and 1 are read at the end
//     of the previous tile, k-steps 2 and 3 in the first slots of the tile, so its LDS slot is free again after ~1/3 of the tile;
//   * schedule per K-tile and wave (64 MFMA slots of 32 cycles), as compile-time data (W4Sched).
#include "common.h"
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F> __device__ __forceinline__ void w4_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w4_for<I + 1, N>(f);
    }
}

#define W4_BM 256
#define W4_BN 256
#define W4_SLOT 65536          // one K-tile of both operands: (256 + 256) rows x 128 bytes
#define W4_ABYTES 32768
#define W4_SMEM (256 * 528)    // the epilogue's fp32 image (256 rows x 132 floats) is the larger user

__device__ __forceinline__ void w4_mfma(f32x16& acc, const bf16x8& b, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
template <int OFF> __device__ __forceinline__ void w4_read(bf16x8& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void w4_dma(uint32_t voff, i32x4 srd, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void w4_m0_set(uint32_t v) { asm volatile("s_mov_b32 m0, %0" : : "s"(v) : "memory"); }
__device__ __forceinline__ void w4_m0_next() { asm volatile("s_add_u32 m0, m0, 0x400" : : : "memory", "scc"); }
__device__ __forceinline__ void w4_barrier() { asm volatile("s_barrier" : : : "memory"); }
template <int N> __device__ __forceinline__ void w4_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void w4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// The K-tile schedule (MFMA slots 0..63):
//   RD1      reads per slot, from slot 0 on, of the 16 fragment reads of this tile's k-steps 2 and 3 (A0..A3, B0..B3 of k-step 2, then 3)
//   BAR1     slot of lgkmcnt(0) + barrier #1 (every wave is done reading this tile's LDS slot)
//   dma_slot(j)  slot of copy j of tile t+2 (0..15: A copies 0..7, then B copies 0..7)
//   BAR2     slot of the vmcnt wait + barrier #2 (tile t+1 has landed for every wave); copies issued behind it stay in flight
//   RD0, RD0N  first slot (> 31: k-step 1's registers are in use until then) and reads per slot of the 16 reads of (t+1, k-steps 0 and 1);
//            the tile ends with lgkmcnt(8): k-step 0 has landed, k-step 1 is waited for at slot 15 of the next tile
template <int V> struct W4Sched;
template <> struct W4Sched<0> {
    static constexpr int RD1 = 1, BAR1 = 20, BAR2 = 44, RD0 = 45, RD0N = 1;
    static constexpr int dma_slot(int j) { return 21 + j; }
};
template <> struct W4Sched<1> {              // copies ~80 cycles apart instead of one per slot
    static constexpr int RD1 = 1, BAR1 = 20, BAR2 = 44, RD0 = 45, RD0N = 1;
    static constexpr int dma_slot(int j) { return 21 + (j * 5) / 2; }
};
template <> struct W4Sched<2> {              // reads 2 per slot: barrier #1 at 12, copies 2 slots apart
    static constexpr int RD1 = 2, BAR1 = 12, BAR2 = 46, RD0 = 47, RD0N = 1;
    static constexpr int dma_slot(int j) { return 13 + 2 * j; }
};
template <> struct W4Sched<3> {              // as 2 with the copies 3 slots apart (the bf16 tile's ~96 cycles)
    static constexpr int RD1 = 2, BAR1 = 12, BAR2 = 48, RD0 = 49, RD0N = 2;
    static constexpr int dma_slot(int j) { return 13 + 3 * j; }
};
#ifndef W4_SCHED
#define W4_SCHED 0
#endif
template <int V> __host__ __device__ constexpr int w4_copies_before_bar2() {
    int n = 0;
    for (int j = 0; j < 16; ++j) n += W4Sched<V>::dma_slot(j) < W4Sched<V>::BAR2 ? 1 : 0;
    return n;
}
template <int V> __host__ __device__ constexpr int w4_copy_at(int sl) {
    for (int j = 0; j < 16; ++j) if (W4Sched<V>::dma_slot(j) == sl) return j;
    return -1;
}

// SWIGLU: B = [gate rows | up rows] (2N x K, N = output width); the B tile interleaves 32 gate rows with the 32 matching up rows (ni even:
// gate, ni odd: up); the epilogue writes bf16(silu(gate)) * up for 128 output columns per workgroup (+ optionally the bf16 gate | up values
// the backward needs) with the roundings of gemm_asm4.hip's SwiGLU epilogue.
// DBG (timing experiments only, results are wrong): 1 = no LDS-DMA in the loop, 3 = no fragment reads, 4 = MFMAs only
template <bool HAS_BIAS, bool HAS_RES, bool SWIGLU = false, int SV = W4_SCHED, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_nt4w_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ B, int64_t ldb,
                                                       const uint16_t* __restrict__ bias, const uint16_t* __restrict__ res, int64_t ldr,
                                                       uint16_t* __restrict__ C, int64_t ldc, uint16_t* __restrict__ gu, int64_t ldgu, int M, int N, int K,
                                                       int tiles_m, int tiles_n) {
    using S = W4Sched<SV>;
    static_assert(S::BAR1 >= (16 + S::RD1 - 1) / S::RD1 - 1 && S::BAR1 < 32 && S::dma_slot(0) > S::BAR1 && S::dma_slot(15) <= 63 && S::RD0 > S::BAR2 &&
                  S::RD0 > 31 && (63 - S::RD0 + 1) * S::RD0N >= 16, "K-tile schedule");
    static_assert(!SWIGLU || (!HAS_BIAS && !HAS_RES), "SwiGLU tiles carry no bias / residual");
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
    const int m0 = (first_m + in_g % gsz) * W4_BM;
    const int n0 = (in_g / gsz) * (SWIGLU ? W4_BN / 2 : W4_BN);      // SWIGLU: first OUTPUT column

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
    const int rows_a = min(W4_BM, M - m0), cols_b = min(SWIGLU ? W4_BN / 2 : W4_BN, N - n0);
    uint32_t voffA[8], voffB[8];
    w4_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int rl = lane >> 3, ch = ((lane & 7) ^ rl) << 4;
        voffA[j] = (uint32_t)min(wave * 64 + j * 8 + rl, rows_a - 1) * (uint32_t)(lda * 2) + ch;
        // SWIGLU: tile rows 64w .. 64w+31 are the gate rows of output columns 32w .., rows 64w+32 .. the up rows (copies 4..7, through srdB2)
        const int src = SWIGLU ? wave * 32 + (j & 3) * 8 + rl : wave * 64 + j * 8 + rl;
        voffB[j] = (uint32_t)min(src, cols_b - 1) * (uint32_t)(ldb * 2) + ch;
    });
    uint32_t koff = 0;                                               // byte offset of the K-tile the next copies fetch (SGPR)
    const uint32_t smem32 = (uint32_t)(uintptr_t)smem;
    const uint32_t m0A = __builtin_amdgcn_readfirstlane(smem32 + wave * 8 * 1024);
    const uint32_t m0B = __builtin_amdgcn_readfirstlane(smem32 + W4_ABYTES + wave * 8 * 1024);

    f32x16 acc[4][4];                                                // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / 64;
    auto advance = [&](bool go) { koff += go ? 128u : 0u; };

    // fragment read addresses: row = wave tile base + i*32 + (lane & 31) (i*4096 bytes as the instruction's offset), chunk (2s + g) ^ (row & 7).
    // The two LDS slots are 64 KiB apart: "the other slot" is an XOR.  k-steps 2, 3 of tile t are read from its slot, k-steps 0, 1 of tile
    // t+1 from the other one; all flip after every tile.
    const int frow = lane & 31, fg = lane >> 5, sw = lane & 7;
    uint32_t adA[4], adB[4];                                         // [k-step]; k-steps 0, 1 point at the OTHER slot
    w4_for<0, 4>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int kc = (2 * s + fg) ^ sw;
        adA[s] = smem32 + (s < 2 ? W4_SLOT : 0) + (wm * 128 + frow) * 128 + (kc << 4);
        adB[s] = smem32 + (s < 2 ? W4_SLOT : 0) + W4_ABYTES + (wn * 128 + frow) * 128 + (kc << 4);
    });
    bf16x8 af[4][4], bfr[4][4];                                      // [k-step][tile]

    auto dma_tile = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < 8) w4_dma(voffA[j], srdA, koff);
        else if constexpr (SWIGLU && j >= 12) w4_dma(voffB[j - 8], srdB2, koff);
        else w4_dma(voffB[j - 8], srdB, koff);
    };
    // read r of k-step s: 0..3 A tiles, 4..7 B tiles
    auto rd_frag = [&](auto rc, auto sc) {
        constexpr int r = decltype(rc)::value, s = decltype(sc)::value;
        if constexpr (r < 4) w4_read<r * 4096>(af[s][r], adA[s]); else w4_read<(r - 4) * 4096>(bfr[s][r - 4], adB[s]);
    };
    // read x (0..15) of the first half of a tile (k-steps 0, 1; BASE = 0) or the second (k-steps 2, 3; BASE = 2)
    auto rd_half = [&](auto xc, auto basec) {
        constexpr int x = decltype(xc)::value, base = decltype(basec)::value;
        rd_frag(std::integral_constant<int, (x & 7)>{}, std::integral_constant<int, (base + (x >> 3))>{});
    };
    using H0 = std::integral_constant<int, 0>;
    using H2 = std::integral_constant<int, 2>;
    uint32_t m0A_cur = m0A, m0B_cur = m0B;                           // LDS-DMA destinations of the tile being refilled (slot of tile t)
    auto issue_tile = [&](uint32_t mA, uint32_t mB) {                // prologue: all 16 copies of one tile back to back
        w4_m0_set(mA);
        w4_for<0, 16>([&](auto jc) {
            if constexpr (decltype(jc)::value == 8) w4_m0_set(mB);
            dma_tile(jc);
            w4_m0_next();
        });
    };
    // ---- prologue: tiles 0 and 1 in flight, fragments of (0, k-steps 0 and 1) in registers
    issue_tile(m0A, m0B);
    advance(nk > 1);
    issue_tile(m0A + W4_SLOT, m0B + W4_SLOT);                        // (nk == 1: tile 0 once more — keeps the loop's counted waits uniform)
    advance(nk > 2);
    w4_wait_vm<16>();
    w4_barrier();
    {                                                                // (0, k-steps 0, 1) sit in slot 0 = "the other slot" of the flipped addresses
        adA[0] ^= W4_SLOT; adA[1] ^= W4_SLOT; adB[0] ^= W4_SLOT; adB[1] ^= W4_SLOT;
        w4_for<0, 16>([&](auto xc) { rd_half(xc, H0{}); });
        adA[0] ^= W4_SLOT; adA[1] ^= W4_SLOT; adB[0] ^= W4_SLOT; adB[1] ^= W4_SLOT;
        w4_wait_lgkm<8>();
    }

    for (int kt = 0; kt < nk; ++kt) {
        w4_for<0, 64>([&](auto ic) {
            constexpr int sl = decltype(ic)::value;
            constexpr int ks = sl >> 4, idx = sl & 15, ni = idx >> 2, mi = idx & 3;
            w4_mfma(acc[ni][mi], bfr[ks][ni], af[ks][mi]);
            constexpr bool RD = DBG != 3 && DBG != 4, BAR = DBG != 4, DMA = DBG != 1 && DBG != 4;
            if constexpr (RD && sl * S::RD1 < 16) {                  // k-steps 2, 3 of this tile
                w4_for<sl * S::RD1, (sl * S::RD1 + S::RD1 < 16 ? sl * S::RD1 + S::RD1 : 16)>([&](auto xc) { rd_half(xc, H2{}); });
            }
            // k-step 1 was read LAST in the previous tile (the eight oldest of the reads outstanding): needed from slot 16
            if constexpr (sl == 15 && S::BAR1 > 15) w4_wait_lgkm<15>();      // <= 24 outstanding: the 9 oldest are done
            if constexpr (sl == S::BAR1) { w4_wait_lgkm<0>(); if constexpr (BAR) w4_barrier(); w4_m0_set(m0A_cur); }
            if constexpr (DMA && w4_copy_at<SV>(sl) >= 0) {
                constexpr int j = w4_copy_at<SV>(sl);
                dma_tile(std::integral_constant<int, j>{});
                if constexpr (j == 7) w4_m0_set(m0B_cur); else if constexpr (j < 15) w4_m0_next();
            }
            if constexpr (sl == S::BAR2) { if constexpr (DMA) w4_wait_vm<w4_copies_before_bar2<SV>()>(); if constexpr (BAR) w4_barrier(); }
            if constexpr (RD && sl >= S::RD0 && (sl - S::RD0) * S::RD0N < 16) {
                constexpr int x0 = (sl - S::RD0) * S::RD0N;
                w4_for<x0, (x0 + S::RD0N < 16 ? x0 + S::RD0N : 16)>([&](auto xc) { rd_half(xc, H0{}); });
            }
            if constexpr (sl == 63) w4_wait_lgkm<8>();
        });
        // flip the slots; advance the source of the next copies (the last two tiles re-fetch tile nk-1: lands in a slot nobody reads)
#pragma unroll
        for (int s = 0; s < 4; ++s) { adA[s] ^= W4_SLOT; adB[s] ^= W4_SLOT; }
        m0A_cur ^= W4_SLOT; m0B_cur ^= W4_SLOT;
        advance(kt + 3 < nk);
    }
    w4_wait_vm<0>();                                                 // the two re-fetched tiles are still landing
    w4_wait_lgkm<0>();
    // MFMA results -> epilogue reads: the hazard checker cannot see into asm; every accumulator passes THROUGH a wait
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[ni][0]), "+a"(acc[ni][1]), "+a"(acc[ni][2]), "+a"(acc[ni][3]));

    // ---- epilogue through LDS: 2 passes of 256 rows x 128 columns of fp32 (rows padded to 528 bytes), as gemm_asm4.hip.
    // Swapped operands (B first): lane holds m = mi*32 + (lane & 31) and, per register group q, n = ni*32 + 8q + 4g + 0..3
    constexpr int ROWB = 128 * 4 + 16;
    const bool interior = m0 + W4_BM <= M && n0 + (SWIGLU ? W4_BN / 2 : W4_BN) <= N && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
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
                if (gu) {                                            // the bf16 gate | up values the backward needs
                    uint16_t* gp = gu + (int64_t)m * ldgu + n;
                    if (n + 7 < N && (reinterpret_cast<uintptr_t>(gp) & 15) == 0 && ((N * 2) & 15) == 0 && (ldgu & 7) == 0) {
                        *reinterpret_cast<uint4*>(gp) = pack8(g);
                        *reinterpret_cast<uint4*>(gp + N) = pack8(u);
                    } else for (int r = 0; r < 8; ++r) if (n + r < N) { gp[r] = f2bf(g[r]); gp[N + r] = f2bf(u[r]); }
                }
            }
            if (p == 0) __syncthreads();
            continue;
        }
        const int c8 = (t & 15) * 8;
        const int n = n0 + (c8 >> 6) * 128 + p * 64 + (c8 & 63);
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


// st_gemm_tile_dispatch (gemm_tiles_train.hip): row-major operands, bf16 result
int st_gemm_asm4w_dispatch(int sched, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res,
                           int64_t ldr, uint16_t* Cb, int64_t ldc, int M, int N, int K, hipStream_t s) {
    if (!Cb || (K % 64) || lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    const int tiles_m = st_cdiv(M, W4_BM), tiles_n = st_cdiv(N, W4_BN);
#define W4GO(HB, HR, SVV) W4GOD(HB, HR, SVV, 0)
#define W4GOD(HB, HR, SVV, D)                                                                                                     \
    do {                                                                                                                          \
        auto kern = gemm_nt4w_kernel<HB, HR, false, SVV, D>;                                                                      \
        static bool configured = false;                                                                                           \
        if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4_SMEM); configured = true; } \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), W4_SMEM, s, A, lda, B, ldb, bias, res, ldr, Cb, ldc, (uint16_t*)nullptr, \
                           (int64_t)0, M, N, K, tiles_m, tiles_n);                                                                \
    } while (0)
    if (sched == 1) W4GO(false, false, 1); else if (sched == 2) W4GO(false, false, 2); else if (sched == 3) W4GO(false, false, 3);
    else if (sched == 4) W4GOD(false, false, 1, 1); else if (sched == 5) W4GOD(false, false, 1, 3); else if (sched == 6) W4GOD(false, false, 1, 4);
    else if (bias && res) W4GO(true, true, W4_SCHED); else if (bias) W4GO(true, false, W4_SCHED); else if (res) W4GO(false, true, W4_SCHED);
    else W4GO(false, false, W4_SCHED);
#undef W4GO
#undef W4GOD
    ST_CHECK_LAUNCH();
    return 0;
}

int st_gemm_asm4w_swiglu(const uint16_t* A, int64_t lda, const uint16_t* gate_up_w, int64_t ldb, uint16_t* gu_out, int64_t ldgu, uint16_t* m_out,
                         int64_t ldm, int M, int I, int K, hipStream_t s) {
    if ((K % 64) || lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    const int tiles_m = st_cdiv(M, W4_BM), tiles_n = st_cdiv(I, W4_BN / 2);
    auto kern = gemm_nt4w_kernel<false, false, true, W4_SCHED>;
    static bool configured = false;
    if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4_SMEM); configured = true; }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), W4_SMEM, s, A, lda, gate_up_w, ldb, (const uint16_t*)nullptr, (const uint16_t*)nullptr,
                       (int64_t)0, m_out, ldm, gu_out, ldgu, M, I, K, tiles_m, tiles_n);
    ST_CHECK_LAUNCH();
    return 0;
}
