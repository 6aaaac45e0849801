// gemm_asm4.hip — the training GEMM tile with a hand-scheduled K loop:  C[M,N] = A[M,K] * B[N,K]^T, bf16 in, fp32 accumulate.
//
// 256x256x64 tile, FOUR waves (one per SIMD), each wave 128x128 = 8x8 MFMA 16x16x32 tiles whose 256 accumulator registers fill the
// AGPR half of the SIMD's register file.  Why this shape (round-3 measurements): with two waves per SIMD the older
// wave takes the matrix pipe and the younger one is left alone with its LDS-DMA issue stalls (~55 cycles each, in-order wave);
// with ONE wave per SIMD nothing competes for issue slots, so the K loop can be written as a fixed instruction stream in which
// every non-MFMA instruction (fragment read, LDS-DMA issue, M0 update, wait, barrier) sits in the shadow of a 16-cycle MFMA — at
// most one or two "fillers" between two MFMAs (MI355X_MICROARCH.md: a 16x16x32 MFMA hides ~2 single-issue instructions).
// Every instruction of the loop is an `asm volatile` statement: hipcc only allocates registers and keeps the statement order.
//   * operands: LDS-DMA through the BUFFER path (`buffer_load_dwordx4 ... offen lds`): one lane-offset VGPR per 1-KiB copy (8 per
//     operand, constant over the K loop, rows / columns past the matrix edge clamped to the last valid one), the K-tile position
//     in an SGPR offset (row-major operands) or in the resource's base address (contraction-major ones), the LDS destination in M0
//     — no address arithmetic on the VALU inside the loop;
//   * LDS image [row][64 k] with the 16-byte chunk position XOR (row & 7), applied on the SOURCE address: conflict-free for the
//     lane groups of ds_read_b128; contraction-major operands (AS / BS: dX = dY W, dW = dY^T X read their operands as stored) use
//     the [64 k][256 m] image of gemm_tile_kernel.h and two ds_read_b64_tr_b16 per fragment;
//   * per K-tile and wave: 128 MFMAs, 32 ds_read_b128 (fragments of the next k-step under the MFMAs of the current one), 16 LDS-DMA
//     issues of tile t+2 into the slot tile t has just vacated, spread over the rest of the tile, two barriers (slot free / next
//     tile landed), counted vmcnt so that the copies already issued stay in flight across the second barrier.
// Epilogue: the accumulators pass through the (idle) operand LDS so that bias / residual / C / the fp32 accumulate target move as
// 16-byte row-contiguous vectors; interior tiles with aligned pointers (the production case) run straight-line code: 16 LDS reads in
// flight, then the stores back to back.  (Round 3: the integer bf16 rounding + the per-row branches of the first version cost 10 us
// of an 87-us tile — profiles/r03_notes.md; f2bf2 / pack8 are v_cvt_pk_bf16_f32 now.)
#include "common.h"
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: the body receives std::integral_constant<int, I> — every index below is a constant expression, so register
// arrays are addressed statically (a loop the optimizer declines to unroll would turn them into indexed VGPR / scratch accesses)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define A4_BM 256
#define A4_BN 256
#define A4_SLOT 65536          // one K-tile of both operands: (256 + 256) rows x 128 bytes
#define A4_ABYTES 32768

__device__ __forceinline__ void a4_mfma(f32x4& acc, const bf16x8& b, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
}
template <int OFF> __device__ __forceinline__ void a4_read(bf16x8& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void a4_dma(uint32_t voff, i32x4 srd, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// K-major operand fragment: 4 consecutive k of one m per lane (gfx950 LDS transpose read), two of them make an MFMA operand
template <int OFF> __device__ __forceinline__ void a4_read_tr(s16x4_t& f, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void a4_dma0(uint32_t voff, i32x4 srd) {        // the K-tile offset sits in the resource's base address
    asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(srd) : "memory");
}
__device__ __forceinline__ void a4_flip(uint32_t& addr) { asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(addr)); }
__device__ __forceinline__ void a4_m0_set(uint32_t v) { asm volatile("s_mov_b32 m0, %0" : : "s"(v) : "memory"); }
__device__ __forceinline__ void a4_m0_next() { asm volatile("s_add_u32 m0, m0, 0x400" : : : "memory", "scc"); }
__device__ __forceinline__ void a4_barrier() { asm volatile("s_barrier" : : : "memory"); }
__device__ __forceinline__ void a4_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }
template <int N> __device__ __forceinline__ void a4_wait_lgkm() {          // LDS reads return in order: "at most N still outstanding"
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    else static_assert(N < 0, "add the lgkmcnt literal");
}
template <int N> __device__ __forceinline__ void a4_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

// SWIGLU: B = [gate rows | up rows] (2N x K); the B tile interleaves 16 gate rows with the 16 matching up rows, so a lane holds gate
// and up of the same output column in adjacent MFMA column tiles (ni even: gate, ni odd: up) and the epilogue writes
// silu(gate) * up for 128 output columns per workgroup (+ optionally the bf16 gate|up values the backward needs).
// DBG (timing experiments only, results are wrong): 1 = no LDS-DMA in the loop, 2 = no barriers, 3 = no fragment reads, 4 = MFMAs only, 5 = no epilogue, 6 = epilogue without the global stores, 7 = the real kernel + a per-workgroup time trace in the workspace,

// DEC: no effect on the code — the decode entry's instantiation gets its own symbol so that kernel traces / PMC summaries can tell the
// 257..512-row decode launches from the prefill's launches of the same tile
template <bool HAS_BIAS, bool HAS_RES, bool OUT_BF16, bool ACCUM, bool SWIGLU, int DBG = 0, bool AS = false, bool BS = false, bool DEC = false>
__global__ __launch_bounds__(256) void gemm_nt4_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ B, int64_t ldb,
                                                      const uint16_t* __restrict__ bias, const uint16_t* __restrict__ res, int64_t ldr,
                                                      uint16_t* __restrict__ Cb, float* __restrict__ Cf, int64_t ldc, uint16_t* __restrict__ gu,
                                                      int64_t ldgu, int M, int N, int K, int tiles_m, int tiles_n, float* __restrict__ tail_ws,
                                                      int full_blocks, int split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    uint64_t t_start = 0;
    if constexpr (DBG == 7) t_start = __builtin_amdgcn_s_memrealtime();   // timing trace: [start, epilogue start, end, hw id] per workgroup in tail_ws

    // XCD-aware bijective remap + grouped tile order (as gemm_tile_kernel.h): each XCD walks a contiguous range of tiles in groups
    // of 8 tile-rows, so co-resident tiles share their A and B panels in the XCD's L2
    // Tail split (as launch_tile in gemm_tile_kernel.h): one 128-KiB workgroup per CU means q*CUs + r tiles cost q + 1 rounds; the last r
    // tiles are cut into `split` K-slices ("pieces", dispatched last) that leave fp32 partial tiles in tail_ws, summed in a fixed
    // order by gemm_a4_finish_kernel, which also runs the bias / residual / accumulate epilogue.
    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    int piece = -1;
    if (bid >= full_blocks) { piece = bid - full_blocks; bid = full_blocks + piece / split; }
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * A4_BM;
    const int n0 = SWIGLU ? tn * (A4_BN / 2) : tn * A4_BN;           // SWIGLU: first OUTPUT column (N = output width I)

    // ---- buffer resources: base = first row of the tile, num_records cuts the rows beyond the matrix (they load as zero)
    static_assert(!(AS || BS) || (!SWIGLU && !HAS_BIAS && !HAS_RES), "contraction-major operands: plain bf16 or fp32 (+=) results");
    // AS / BS: the operand is stored CONTRACTION-MAJOR (A[k][m], lda = pitch of a k-row): dX = dY W reads W that way (BS), dW = dY^T X
    // both operands (AS + BS).  Its K-tile is staged as [64 k-rows][256 m] (512-byte rows) and read with the LDS transpose read.
    uint64_t a_base = AS ? (uint64_t)(A + m0) : (uint64_t)(A + (int64_t)m0 * lda);
    const int rows_a = min(A4_BM, M - m0);
    i32x4 srdA, srdB, srdB2;
    srdA.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a_base);
    srdA.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a_base >> 32) & 0xffffu));
    srdA.z = (int)0xffffffffu;                                       // rows are clamped per lane: nothing to cut off
    srdA.w = 0x00020000;
    const int cols_b = SWIGLU ? min(A4_BN / 2, N - n0) : min(A4_BN, N - n0);        // B rows (= output columns) this tile owns
    uint64_t b_base = BS ? (uint64_t)(B + n0) : (uint64_t)(B + (int64_t)n0 * ldb);
    srdB.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b_base);
    srdB.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b_base >> 32) & 0xffffu));
    srdB.z = (int)0xffffffffu;
    srdB.w = 0x00020000;
    srdB2 = srdB;
    if constexpr (SWIGLU) {                                          // the up-projection rows follow the N gate rows
        const uint64_t u_base = (uint64_t)(B + (int64_t)(n0 + N) * ldb);
        srdB2.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)u_base);
        srdB2.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((u_base >> 32) & 0xffffu));
    }
    // source offset of copy j of this wave, per lane: row (lane >> 3) of the copy's 8 rows — clamped to the last valid row of the
    // tile, such rows compute values that are never stored — and the 16-byte chunk (lane & 7) ^ row (the swizzle, on the source).
    // These 16 registers are constant over the K loop: the K-tile offset goes through the instruction's SGPR offset.
    uint32_t voffA[8], voffB[8];
    static_for<0, 8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int rl = lane >> 3, ch = ((lane & 7) ^ rl) << 4;
        // contraction-major image: copy j of this wave = k-rows kr = wave*16 + 2j + (lane >> 5); the 32-byte chunk c (16 m) of k-row kr
        // sits at chunk position c ^ g(kr) (gemm_tile_kernel.h, AS / BS): lane piece sl covers m = ((sl >> 1) ^ g) * 16 + (sl & 1) * 8 ..+7;
        // columns past the matrix edge are clamped (they compute values that are never stored)
        const int kr = wave * 16 + j * 2 + (lane >> 5), sl = lane & 31, g = (kr & 3) | (((kr >> 3) & 1) << 2);
        const int lc = (((sl >> 1) ^ g) << 4) + (sl & 1) * 8;
        if constexpr (AS) voffA[j] = (uint32_t)kr * (uint32_t)(lda * 2) + (uint32_t)min(lc, rows_a - 8) * 2u;
        else {
            const int ra = min(wave * 64 + j * 8 + rl, rows_a - 1);
            voffA[j] = (uint32_t)ra * (uint32_t)(lda * 2) + ch;
        }
        if constexpr (BS) voffB[j] = (uint32_t)kr * (uint32_t)(ldb * 2) + (uint32_t)min(lc, cols_b - 8) * 2u;
        else {
            const int r = wave * 64 + j * 8 + rl;                    // tile row (SWIGLU: tile rows alternate 16 gate / 16 up rows)
            const int src = min(SWIGLU ? ((r >> 5) * 16 + (r & 15)) : r, cols_b - 1);
            voffB[j] = (uint32_t)src * (uint32_t)(ldb * 2) + ch;
        }
    });
    uint32_t koff = 0;                                               // byte offset of the K-tile the next copies fetch (SGPR)
    const uint32_t smem32 = (uint32_t)(uintptr_t)smem;
    const uint32_t m0A = __builtin_amdgcn_readfirstlane(smem32 + wave * 8 * 1024);
    const uint32_t m0B = __builtin_amdgcn_readfirstlane(smem32 + A4_ABYTES + wave * 8 * 1024);

    f32x4 acc[8][8];                                                 // [ni][mi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int nk = K / 64;
    if (piece >= 0) {                                                // this workgroup's K-slice of a tail tile
        const int per = (nk + split - 1) / split, kt0 = (piece % split) * per;
        nk = min(nk - kt0, per);
        koff = (uint32_t)kt0 * 128u;
        if constexpr (AS) a_base += (uint64_t)kt0 * 64u * (uint64_t)lda * 2u;
        if constexpr (BS) b_base += (uint64_t)kt0 * 64u * (uint64_t)ldb * 2u;
    }
    // contraction-major operands advance their resource's BASE by 64 k-rows per K-tile (the 32-bit offsets could not span a
    // [tokens][vocabulary] operand); row-major ones keep the base and move the SGPR offset by 128 bytes
    const uint64_t stepA = (uint64_t)lda * 128u, stepB = (uint64_t)ldb * 128u;
    auto set_base = [&](i32x4& srd, uint64_t base) {
        srd.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        srd.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffffu));
    };
    if constexpr (AS) set_base(srdA, a_base);
    if constexpr (BS) set_base(srdB, b_base);
    auto advance = [&](bool go) {                                    // the K-tile the NEXT copies fetch
        koff += go ? 128u : 0u;
        if constexpr (AS) { a_base += go ? stepA : 0; set_base(srdA, a_base); }
        if constexpr (BS) { b_base += go ? stepB : 0; set_base(srdB, b_base); }
    };
    // fragment addresses: row = wave tile base + i*16 + (lane & 15), 16-byte chunk (s*4 + lane>>4) ^ (row & 7)
    const int frow = lane & 15, fk = lane >> 4;
    // fragment read addresses; the two LDS slots are 64 KiB apart, so "the other slot" is an XOR with 0x10000: the k-step-1 reads of
    // tile t go to its slot (adK1), the k-step-0 reads of tile t+1 to the other one (adK0n); both flip after every tile
    uint32_t adA_k1, adB_k1, adA_k0n, adB_k0n;
    {
        const int kc0 = fk ^ (frow & 7), kc1 = (4 + fk) ^ (frow & 7);
        adA_k1 = smem32 + (wm * 128 + frow) * 128 + (kc1 << 4);
        adB_k1 = smem32 + A4_ABYTES + (wn * 128 + frow) * 128 + (kc1 << 4);
        adA_k0n = smem32 + A4_SLOT + (wm * 128 + frow) * 128 + (kc0 << 4);
        adB_k0n = smem32 + A4_SLOT + A4_ABYTES + (wn * 128 + frow) * 128 + (kc0 << 4);
    }
    bf16x8 af[2][8], bfr[2][8];
    // contraction-major fragments: m = chunk*16 + (lane & 15) with chunk = 8*w + i, k = (lane >> 4)*8 + 0..7 as two transpose reads
    // (k-rows +0..3 and +4..7 = 2048 bytes further).  Chunk c of k-row r sits at position c ^ g(r); for the rows a lane addresses
    // g = tr_g, so fragment i needs its own per-lane address (i ^ tr_g) << 5: 8 registers per operand, pointing at the slot of the
    // tile whose k-step 1 is read next; they flip ONCE per tile, between the k-step-1 reads (tile t) and the k-step-0 reads (t+1).
    uint32_t trA[8], trB[8];
    union Frag { bf16x8 v; s16x4_t h[2]; };
    if constexpr (AS || BS) {
        const int tr_g = ((lane >> 2) & 3) | (((lane >> 4) & 1) << 2);
        const int tr_row = (lane >> 4) * 8 + ((lane & 15) >> 2);
        static_for<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            trA[i] = smem32 + tr_row * 512 + wm * 256 + ((i ^ tr_g) << 5) + (lane & 3) * 8;
            trB[i] = smem32 + A4_ABYTES + tr_row * 512 + wn * 256 + ((i ^ tr_g) << 5) + (lane & 3) * 8;
        });
    }
    auto rd_tr = [&](bf16x8& f, uint32_t addr, auto s2c, auto hc) {  // half h of the fragment of k-step s2
        constexpr int s2 = decltype(s2c)::value, h = decltype(hc)::value;
        a4_read_tr<s2 * 16384 + h * 2048>(reinterpret_cast<Frag&>(f).h[h], addr);
    };

    auto dma_tile = [&](auto jc) {                                   // copy j of the NEXT issue (voff already points at its K-tile)
        constexpr int j = decltype(jc)::value;
        if constexpr (j < 8) { if constexpr (AS) a4_dma0(voffA[j], srdA); else a4_dma(voffA[j], srdA, koff); }
        else if constexpr (BS) a4_dma0(voffB[j - 8], srdB);
        else if (SWIGLU && (((wave * 64 + (j - 8) * 8) >> 4) & 1)) a4_dma(voffB[j - 8], srdB2, koff);
        else a4_dma(voffB[j - 8], srdB, koff);
    };
    // i < 8: A fragment i, else B fragment i - 8 (16 rows = 2048 bytes apart)
    auto rd_k1 = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        if constexpr (i < 8) a4_read<i * 2048>(af[1][i], adA_k1);
        else a4_read<(i - 8) * 2048>(bfr[1][i - 8], adB_k1);
    };
    auto rd_k0n = [&](auto i_c) {                                    // order of issue: A0..A7, B0 first (the next tile's first MFMAs need them)
        constexpr int i = decltype(i_c)::value;
        if constexpr (i < 8) a4_read<i * 2048>(af[0][i], adA_k0n);
        else a4_read<(i - 8) * 2048>(bfr[0][i - 8], adB_k0n);
    };
    // contraction-major kernels: read number r of a k-step, A first (AS: 16 transpose reads, else 8 ds_read_b128), then B
    constexpr int NRA = AS ? 16 : 8, NRB = BS ? 16 : 8, NRD = NRA + NRB;
    auto rd_km = [&](auto r_c, auto s2c) {
        constexpr int r = decltype(r_c)::value, s2 = decltype(s2c)::value;
        if constexpr (r < NRA) {
            if constexpr (AS) rd_tr(af[s2][r >> 1], trA[r >> 1], s2c, std::integral_constant<int, (r & 1)>{});
            else if constexpr (s2 == 1) a4_read<r * 2048>(af[1][r], adA_k1);
            else a4_read<r * 2048>(af[0][r], adA_k0n);
        } else {
            constexpr int q = r - NRA;
            if constexpr (BS) rd_tr(bfr[s2][q >> 1], trB[q >> 1], s2c, std::integral_constant<int, (q & 1)>{});
            else if constexpr (s2 == 1) a4_read<q * 2048>(bfr[1][q], adB_k1);
            else a4_read<q * 2048>(bfr[0][q], adB_k0n);
        }
    };
    uint32_t m0A_cur = m0A, m0B_cur = m0B;                           // LDS-DMA destination of the tile being refilled (slot of tile t)

    // ---- prologue: tiles 0 and 1 in flight, fragments of (0, k-step 0) in registers
    a4_m0_set(m0A);
    static_for<0, 16>([&](auto jc) {
        if constexpr (decltype(jc)::value == 8) a4_m0_set(m0B);
        dma_tile(jc);
        a4_m0_next();
    });
    advance(nk > 1);
    a4_m0_set(m0A + A4_SLOT);
    static_for<0, 16>([&](auto jc) {                                 // (nk == 1: tile 0 once more — keeps the loop's counted waits uniform)
        if constexpr (decltype(jc)::value == 8) a4_m0_set(m0B + A4_SLOT);
        dma_tile(jc);
        a4_m0_next();
    });
    advance(nk > 2);
    a4_wait_vm<16>();
    a4_barrier();
    if constexpr (AS || BS) {                                        // (0, k-step 0) from slot 0: where the transpose-read addresses point
        adA_k0n ^= A4_SLOT; adB_k0n ^= A4_SLOT;
        static_for<0, NRD>([&](auto rc) { rd_km(rc, std::integral_constant<int, 0>{}); });
        adA_k0n ^= A4_SLOT; adB_k0n ^= A4_SLOT;
        a4_wait_lgkm<0>();
    } else {                                                         // (0, k-step 0) sits in slot 0 = "the other slot" of the flipped addresses
        adA_k0n ^= A4_SLOT; adB_k0n ^= A4_SLOT;
        static_for<0, 16>([&](auto ic) { rd_k0n(ic); });
        adA_k0n ^= A4_SLOT; adB_k0n ^= A4_SLOT;
        a4_wait_lgkm<7>();
    }

    // ---- one K-tile: 128 MFMA slots; every other instruction sits between two of them
    //   slot   0..63  MFMAs of k-step 0          64..127  MFMAs of k-step 1
    //   1..16          the 16 fragment reads of k-step 1 (this tile's slot), one per MFMA
    //   7,15,23        counted lgkmcnt waits: B fragments 1..7 of k-step 0 were issued LAST in the previous tile and land now
    //   30 / 31        lgkmcnt(0) + barrier #1: every wave is done reading this tile's slot
    //   33..123        the 16 LDS-DMA copies of tile t+2 into that slot, one every SIXTH MFMA: an issue costs the wave ~30-55 cycles of
    //                  MFMA issue time (measured by elimination, tools/gemm_ksweep.py debug), the more the closer the four waves' copies
    //                  follow each other through the CU's one texture addresser — so they are spread over the whole rest of the tile
    //   104 / 105      vmcnt(12) + barrier #2: tile t+1 has landed for every wave (12 copies of tile t+2 issued so far stay in flight;
    //                  the last four go out behind this barrier and are covered by the next tile's wait)
    //   106..121       the 16 fragment reads of (t+1, k-step 0): A0..A7, B0, then B1..B7; lgkmcnt(7) closes the tile
    // Contraction-major kernels (NRD = 24 or 32 reads per k-step, one per MFMA slot):
    //   1..NRD               reads of (t, k-step 1)
    //   NRD+1..NRD+16        the transpose-read addresses flip to the other slot (one v_xor per slot)
    //   NRD+14 / NRD+15      lgkmcnt(0) + barrier #1
    //   NRD+17 ..            the 16 copies of tile t+2, one every 4th (NRD = 32) / 5th (24) MFMA
    //   114-NRD / 115-NRD    vmcnt (the copies issued so far stay in flight) + barrier #2
    //   116-NRD..115         reads of (t+1, k-step 0); lgkmcnt(0) at 126
    constexpr int KM_DMA0 = NRD + 17, KM_STRIDE = NRD == 32 ? 4 : 5, KM_VM = 114 - NRD, KM_RD0 = 116 - NRD;
    constexpr int KM_INFLIGHT = (KM_VM - KM_DMA0 + KM_STRIDE - 1) / KM_STRIDE;      // copies of tile t+2 issued before the vmcnt wait
    static_assert(!(AS || BS) || (KM_INFLIGHT == (NRD == 32 ? 9 : 10) && KM_DMA0 + 15 * KM_STRIDE < 127), "contraction-major schedule");
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (AS || BS) {
            static_for<0, 128>([&](auto ic) {
                constexpr int sl = decltype(ic)::value;
                constexpr int ks = sl >> 6, idx = sl & 63;
                a4_mfma(acc[idx >> 3][idx & 7], bfr[ks][idx >> 3], af[ks][idx & 7]);
                if constexpr (sl >= 1 && sl <= NRD) rd_km(std::integral_constant<int, (sl - 1)>{}, std::integral_constant<int, 1>{});
                if constexpr (sl > NRD && sl <= NRD + 16) {
                    constexpr int x = sl - NRD - 1;
                    if constexpr (x < 8) { if constexpr (AS) a4_flip(trA[x]); } else { if constexpr (BS) a4_flip(trB[x - 8]); }
                }
                if constexpr (sl == NRD + 14) a4_wait_lgkm<0>();
                if constexpr (sl == NRD + 15) a4_barrier();
                if constexpr (sl == KM_DMA0 - 1) a4_m0_set(m0A_cur);
                if constexpr (sl >= KM_DMA0 && sl <= KM_DMA0 + 15 * KM_STRIDE && (sl - KM_DMA0) % KM_STRIDE == 0) dma_tile(std::integral_constant<int, ((sl - KM_DMA0) / KM_STRIDE)>{});
                if constexpr (sl > KM_DMA0 && sl <= KM_DMA0 + 15 * KM_STRIDE && (sl - KM_DMA0 - 1) % KM_STRIDE == 0 && (sl - KM_DMA0 - 1) / KM_STRIDE != 7 && (sl - KM_DMA0 - 1) / KM_STRIDE < 15) a4_m0_next();
                if constexpr (sl == KM_DMA0 + 7 * KM_STRIDE + 2) a4_m0_set(m0B_cur);     // after the 8th A copy, before the first B copy
                if constexpr (sl == KM_VM) a4_wait_vm<KM_INFLIGHT>();
                if constexpr (sl == KM_VM + 1) a4_barrier();
                if constexpr (sl >= KM_RD0 && sl < KM_RD0 + NRD) rd_km(std::integral_constant<int, (sl - KM_RD0)>{}, std::integral_constant<int, 0>{});
                if constexpr (sl == 126) a4_wait_lgkm<0>();
            });
        } else
        static_for<0, 128>([&](auto ic) {
            constexpr int sl = decltype(ic)::value;
            constexpr int ks = sl >> 6, idx = sl & 63;
            a4_mfma(acc[idx >> 3][idx & 7], bfr[ks][idx >> 3], af[ks][idx & 7]);
            constexpr bool RD = DBG != 3 && DBG != 4, BAR = DBG != 2 && DBG != 4, DMA = DBG != 1 && DBG != 4;
            if constexpr (sl == 7) a4_wait_lgkm<12>();           // before MFMA 8 (B1): B2..B7 + the 6 k-step-1 reads issued so far
            if constexpr (sl == 15) a4_wait_lgkm<15>();          // B2 (the counter saturates at 15: at least the 5 oldest of <= 20 are done)
            if constexpr (sl == 23) a4_wait_lgkm<15>();          // B3..B7: the 6 oldest of the <= 21 then outstanding
            if constexpr (RD && sl >= 1 && sl <= 16) rd_k1(std::integral_constant<int, (sl - 1)>{});
            if constexpr (sl == 30) a4_wait_lgkm<0>();
            if constexpr (BAR && sl == 31) a4_barrier();
            if constexpr (DMA) {
                if constexpr (sl == 32) a4_m0_set(m0A_cur);
                if constexpr (sl >= 33 && sl <= 123 && (sl - 33) % 6 == 0) dma_tile(std::integral_constant<int, ((sl - 33) / 6)>{});
                if constexpr (sl >= 34 && sl <= 123 && (sl - 34) % 6 == 0 && (sl - 34) / 6 != 7 && (sl - 34) / 6 < 15) a4_m0_next();
                if constexpr (sl == 78) a4_m0_set(m0B_cur);      // after the 8th A copy (slot 75), before the first B copy (slot 81)
                if constexpr (sl == 104) a4_wait_vm<12>();
            }
            if constexpr (BAR && sl == 105) a4_barrier();
            if constexpr (RD && sl >= 106 && sl <= 121) rd_k0n(std::integral_constant<int, (sl - 106)>{});
            if constexpr (sl == 127) a4_wait_lgkm<7>();
        });
        // flip the slots; advance the source of the next copies (the last two tiles re-fetch tile nk-1: lands in a slot nobody reads)
        adA_k1 ^= A4_SLOT; adB_k1 ^= A4_SLOT; adA_k0n ^= A4_SLOT; adB_k0n ^= A4_SLOT;
        m0A_cur ^= A4_SLOT; m0B_cur ^= A4_SLOT;
        advance(kt + 3 < nk);
    }
    a4_wait_vm<0>();                                                 // the two re-fetched tiles are still landing
    a4_wait_lgkm<0>();
    // MFMA results -> epilogue reads: the hazard checker cannot see into asm; every accumulator passes THROUGH a wait
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
        asm volatile("s_nop 7" : "+a"(acc[ni][0]), "+a"(acc[ni][1]), "+a"(acc[ni][2]), "+a"(acc[ni][3]), "+a"(acc[ni][4]), "+a"(acc[ni][5]),
                     "+a"(acc[ni][6]), "+a"(acc[ni][7]));

    // ---- epilogue through LDS: 2 passes of 256 rows x 128 columns of fp32 (rows padded to 528 bytes: conflict-free b128 writes)
    constexpr int ROWB = 128 * 4 + 16;
    bool interior;                                                   // uniform: no edge handling, vector accesses everywhere
    if constexpr (SWIGLU) {
        interior = m0 + A4_BM <= M && n0 + A4_BN / 2 <= N && (N & 7) == 0 && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(Cb) & 15) == 0 &&
                   (gu == nullptr || ((ldgu & 7) == 0 && (reinterpret_cast<uintptr_t>(gu) & 15) == 0));
    } else {
        interior = m0 + A4_BM <= M && n0 + A4_BN <= N && (ldc & (OUT_BF16 ? 7 : 3)) == 0 &&
                   (reinterpret_cast<uintptr_t>(OUT_BF16 ? (const void*)Cb : (const void*)Cf) & 15) == 0 &&
                   (!HAS_BIAS || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                   (!HAS_RES || ((ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0));
    }
    if constexpr (DBG == 5) { if (M > 0) return; }
    uint64_t t_epi = 0;
    if constexpr (DBG == 7) t_epi = __builtin_amdgcn_s_memrealtime();                   // timing: no epilogue at all
    __syncthreads();                                                 // every wave is done with the operand slots; no DMA in flight
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int nl = 0; nl < 4; ++nl)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int row = wm * 128 + mi * 16 + (lane & 15), col = wn * 64 + nl * 16 + (lane >> 4) * 4;
                // SWIGLU: pass p holds tile columns ni = p*4 + nl: even ni = gate, odd = up of output columns (ni >> 1) * 16 ..
                *reinterpret_cast<f32x4*>(smem + row * ROWB + col * 4) = acc[p * 4 + nl][mi];
            }
        __syncthreads();
        const int t = threadIdx.x;
        if (piece >= 0) {                                            // K-slice of a tail tile: the raw fp32 tile, row-major [256][256]
            // (SwiGLU tiles too — round 4: the silu(gate) * up of a split tile is formed by gemm_a4_swiglu_finish_kernel from the summed slices;
            // raw column wn*128 + p*64 + x holds image column 64*wn + x of pass p, see the column map there)
            const int c8 = (t & 15) * 8;
            const int col = (c8 >> 6) * 128 + p * 64 + (c8 & 63);
            float* wp = tail_ws + (int64_t)piece * (A4_BM * A4_BN);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 16 + (t >> 4);
                *reinterpret_cast<float4*>(wp + row * A4_BN + col) = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4);
                *reinterpret_cast<float4*>(wp + row * A4_BN + col + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4 + 16);
            }
        } else if constexpr (SWIGLU) {
            // image columns: wave wn at 64*wn; inside, [gate16 | up16 | gate16 | up16]; thread handles 8 output columns of one row
            const int c8 = (t & 7) * 8;                              // 64 output columns per pass: (wn, pair q, half h)
            const int wn_ = c8 >> 5, q = (c8 >> 4) & 1, h = c8 & 8;
            const int lcol = wn_ * 64 + q * 32 + h;                  // gate columns at lcol .. lcol+7, up at +16
            const int n = n0 + wn_ * 64 + p * 32 + q * 16 + h;      // output column
            if (interior) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 q4[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (h * 4 + i) * 32 + (t >> 3);
                        q4[i][0] = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4);
                        q4[i][1] = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4 + 16);
                        q4[i][2] = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 16) * 4);
                        q4[i][3] = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 16) * 4 + 16);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (h * 4 + i) * 32 + (t >> 3), m = m0 + row;
                        float g[8] = {q4[i][0].x, q4[i][0].y, q4[i][0].z, q4[i][0].w, q4[i][1].x, q4[i][1].y, q4[i][1].z, q4[i][1].w};
                        float u[8] = {q4[i][2].x, q4[i][2].y, q4[i][2].z, q4[i][2].w, q4[i][3].x, q4[i][3].y, q4[i][3].z, q4[i][3].w};
                        float o[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            g[r] = bfround(g[r]); u[r] = bfround(u[r]);
                            o[r] = bfround(g[r] * sigmoidf_(g[r])) * u[r];
                        }
                        *reinterpret_cast<uint4*>(Cb + (int64_t)m * ldc + n) = pack8(o);
                        if (gu) {
                            uint16_t* gp = gu + (int64_t)m * ldgu + n;
                            *reinterpret_cast<uint4*>(gp) = pack8(g);
                            *reinterpret_cast<uint4*>(gp + N) = pack8(u);
                        }
                    }
                }
            } else
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                const int row = it * 32 + (t >> 3), m = m0 + row;
                if (m >= M || n >= N) continue;
                float g[8], u[8], o[8];
                *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4);
                *reinterpret_cast<float4*>(g + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + lcol * 4 + 16);
                *reinterpret_cast<float4*>(u) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 16) * 4);
                *reinterpret_cast<float4*>(u + 4) = *reinterpret_cast<const float4*>(smem + row * ROWB + (lcol + 16) * 4 + 16);
#pragma unroll
                for (int r = 0; r < 8; ++r) {                        // same roundings as st_gemm_nt + st_swiglu_fwd: bf16 gate/up, bf16 act
                    g[r] = bfround(g[r]); u[r] = bfround(u[r]);
                    o[r] = bfround(g[r] * sigmoidf_(g[r])) * u[r];
                }
                const bool full = n + 7 < N;
                uint16_t* cp = Cb + (int64_t)m * ldc + n;
                if (full && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) *reinterpret_cast<uint4*>(cp) = pack8(o);
                else for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
                if (gu) {
                    uint16_t* gp = gu + (int64_t)m * ldgu + n;
                    if (full && (reinterpret_cast<uintptr_t>(gp) & 15) == 0 && ((N * 2) & 15) == 0) {
                        *reinterpret_cast<uint4*>(gp) = pack8(g);
                        *reinterpret_cast<uint4*>(gp + N) = pack8(u);
                    } else for (int r = 0; r < 8; ++r) if (n + r < N) { gp[r] = f2bf(g[r]); gp[N + r] = f2bf(u[r]); }
                }
            }
        } else if (interior) {
            // whole tile inside the matrix, every pointer 16-byte aligned (the production case): straight-line code, two batches of 8
            // rows — 16 LDS reads (and 8 residual loads) in flight, then the stores back to back.  The generic path below pays a
            // branch and a serialised LDS round trip per row (measured: 10 us of an 87-us tile, tools/epilogue_cost.py).
            const int c8 = (t & 15) * 8;
            const int n = n0 + (c8 >> 6) * 128 + p * 64 + (c8 & 63);
            float bvals[8];
            if constexpr (HAS_BIAS) unpack8(*reinterpret_cast<const uint4*>(bias + n), bvals);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 lo[8], hi[8];
                uint4 rr[8];
                float4 c0[8], c1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = (h * 8 + i) * 16 + (t >> 4);
                    lo[i] = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4);
                    hi[i] = *reinterpret_cast<const float4*>(smem + row * ROWB + c8 * 4 + 16);
                    if constexpr (HAS_RES) rr[i] = *reinterpret_cast<const uint4*>(res + (int64_t)(m0 + row) * ldr + n);
                    if constexpr (!OUT_BF16 && ACCUM) {
                        c0[i] = *reinterpret_cast<const float4*>(Cf + (int64_t)(m0 + row) * ldc + n);
                        c1[i] = *reinterpret_cast<const float4*>(Cf + (int64_t)(m0 + row) * ldc + n + 4);
                    }
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
                    if constexpr (DBG == 6) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])); if (M > 0) continue; }
                    if constexpr (OUT_BF16) {
                        *reinterpret_cast<uint4*>(Cb + (int64_t)(m0 + row) * ldc + n) = pack8(v);
                    } else {
                        float* cp = Cf + (int64_t)(m0 + row) * ldc + n;
                        float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                        if constexpr (ACCUM) { o0.x += c0[i].x; o0.y += c0[i].y; o0.z += c0[i].z; o0.w += c0[i].w; o1.x += c1[i].x; o1.y += c1[i].y; o1.z += c1[i].z; o1.w += c1[i].w; }
                        *reinterpret_cast<float4*>(cp) = o0; *reinterpret_cast<float4*>(cp + 4) = o1;
                    }
                }
            }
        } else {
            const int c8 = (t & 15) * 8;
            const int n = n0 + (c8 >> 6) * 128 + p * 64 + (c8 & 63);
            float bvals[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool ncols = n + 7 < N;
            if constexpr (HAS_BIAS) {
                if (ncols && (reinterpret_cast<uintptr_t>(bias + n) & 15) == 0) unpack8(*reinterpret_cast<const uint4*>(bias + n), bvals);
                else for (int r = 0; r < 8; ++r) bvals[r] = n + r < N ? bf2f(bias[n + r]) : 0.f;
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
                    float rr[8];
                    if (ncols && (reinterpret_cast<uintptr_t>(rp) & 15) == 0) unpack8(*reinterpret_cast<const uint4*>(rp), rr);
                    else for (int r = 0; r < 8; ++r) rr[r] = n + r < N ? bf2f(rp[r]) : 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += rr[r];
                }
                if constexpr (DBG == 6) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7])); if (M > 0) continue; }
                if constexpr (OUT_BF16) {
                    uint16_t* cp = Cb + (int64_t)m * ldc + n;
                    if (ncols && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) *reinterpret_cast<uint4*>(cp) = pack8(v);
                    else for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
                } else {
                    float* cp = Cf + (int64_t)m * ldc + n;
                    if (ncols && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) {
                        float4 o0 = ACCUM ? *reinterpret_cast<float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
                        float4 o1 = ACCUM ? *reinterpret_cast<float4*>(cp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                        o0.x += v[0]; o0.y += v[1]; o0.z += v[2]; o0.w += v[3]; o1.x += v[4]; o1.y += v[5]; o1.z += v[6]; o1.w += v[7];
                        *reinterpret_cast<float4*>(cp) = o0; *reinterpret_cast<float4*>(cp + 4) = o1;
                    } else {
                        for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = (ACCUM ? cp[r] : 0.f) + v[r];
                    }
                }
            }
        }
        if (p == 0) __syncthreads();                             // the second pass overwrites the image
    }
    if constexpr (DBG == 7) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the stores have been accepted by L2
        const uint64_t t_end = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            uint64_t* tr = reinterpret_cast<uint64_t*>(tail_ws) + (int64_t)blockIdx.x * 4;
            tr[0] = t_start; tr[1] = t_epi; tr[2] = t_end; tr[3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
        }
    }
}

// Sum the K-slices of the tail tiles (fixed order: deterministic) and run the epilogue.  grid (tail tiles, 32): a block owns 8 rows of a
// tile, a thread 8 consecutive columns of one row.
template <bool HAS_BIAS, bool HAS_RES, bool OUT_BF16, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_a4_finish_kernel(const float* __restrict__ ws, int split, int full_blocks, const uint16_t* __restrict__ bias,
                                                            const uint16_t* __restrict__ res, int64_t ldr, uint16_t* __restrict__ Cb,
                                                            float* __restrict__ Cf, int64_t ldc, int M, int N, int tiles_m, int tiles_n) {
    const int nb = tiles_m * tiles_n;
    int bid = full_blocks + blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int row = blockIdx.y * 8 + (threadIdx.x >> 5), col = (threadIdx.x & 31) * 8;
    const int m = tm * A4_BM + row, n = tn * A4_BN + col;
    if (m >= M || n >= N) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < split; ++sp) {
        const float* wp = ws + ((int64_t)blockIdx.x * split + sp) * (A4_BM * A4_BN) + row * A4_BN + col;
        const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const bool ncols = n + 7 < N;
    if constexpr (HAS_BIAS) {
        for (int r = 0; r < 8; ++r) if (n + r < N) v[r] += bf2f(bias[n + r]);
    }
    if constexpr (HAS_RES) {
        const uint16_t* rp = res + (int64_t)m * ldr + n;
        for (int r = 0; r < 8; ++r) if (n + r < N) v[r] += bf2f(rp[r]);
    }
    if constexpr (OUT_BF16) {
        uint16_t* cp = Cb + (int64_t)m * ldc + n;
        if (ncols && (reinterpret_cast<uintptr_t>(cp) & 15) == 0) *reinterpret_cast<uint4*>(cp) = pack8(v);
        else for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
    } else {
        float* cp = Cf + (int64_t)m * ldc + n;
        for (int r = 0; r < 8; ++r) if (n + r < N) cp[r] = (ACCUM ? cp[r] : 0.f) + v[r];
    }
}

// SwiGLU tiles cut into K-slices (round 4: the 257..512-row decode gate/up GEMM is 296 tiles = 1.16 rounds of CUs): sum the raw slices in
// a fixed order, round gate and up to bf16 and write bf16(silu(gate)) * up — the roundings of the unsplit epilogue.  Raw tile column
// wn*128 + p*64 + q*32 + j (j < 16) holds the GATE of output column wn*64 + p*32 + q*16 + j, the UP value sits 16 columns further.
// grid (tail tiles, 32): a block owns 8 rows of a tile, a thread 4 consecutive output columns of one row.
__global__ __launch_bounds__(256) void gemm_a4_swiglu_finish_kernel(const float* __restrict__ ws, int split, int full_blocks, uint16_t* __restrict__ Cb,
                                                                   int64_t ldc, int M, int N, int tiles_m, int tiles_n) {
    const int nb = tiles_m * tiles_n;
    int bid = full_blocks + blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per_group = 8 * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * 8;
    const int gsz = min(tiles_m - first_m, 8);
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int row = blockIdx.y * 8 + (threadIdx.x >> 5), nl = (threadIdx.x & 31) * 4;
    const int m = tm * A4_BM + row, n = tn * (A4_BN / 2) + nl;
    if (m >= M || n >= N) return;
    const int gcol = (nl >> 6) * 128 + ((nl >> 5) & 1) * 64 + ((nl >> 4) & 1) * 32 + (nl & 15);
    float g[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < split; ++sp) {
        const float* wp = ws + ((int64_t)blockIdx.x * split + sp) * (A4_BM * A4_BN) + row * A4_BN + gcol;
        const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 16);
        g[0] += a.x; g[1] += a.y; g[2] += a.z; g[3] += a.w; u[0] += b.x; u[1] += b.y; u[2] += b.z; u[3] += b.w;
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        g[r] = bfround(g[r]); u[r] = bfround(u[r]);
        o[r] = bfround(g[r] * sigmoidf_(g[r])) * u[r];
    }
    uint16_t* cp = Cb + (int64_t)m * ldc + n;
    if (n + 3 < N && (reinterpret_cast<uintptr_t>(cp) & 7) == 0) {
        uint2 w;
        w.x = f2bf2(o[0], o[1]); w.y = f2bf2(o[2], o[3]);
        *reinterpret_cast<uint2*>(cp) = w;
    } else for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(o[r]);
}

extern float* g_tail_ws;          // st_gemm_set_workspace (gemm_tiles.hip)
extern int64_t g_tail_ws_bytes;

template <bool HB, bool HR, bool OB, bool AC, bool SW, bool AS = false, bool BS = false, bool DEC = false>
static int launch_asm4(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res, int64_t ldr,
                       uint16_t* Cb, float* Cf, int64_t ldc, uint16_t* gu, int64_t ldgu, int M, int N, int K, hipStream_t s, bool sw_tail = false) {
    constexpr int smem = 256 * 528;                              // >= the two 64-KiB operand slots
    auto kern = gemm_nt4_kernel<HB, HR, OB, AC, SW, 0, AS, BS, DEC>;
    static bool configured = false;
    if (!configured) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    const int tiles_m = st_cdiv(M, A4_BM), tiles_n = st_cdiv(N, SW ? A4_BN / 2 : A4_BN);
    const int nb = tiles_m * tiles_n;
    int full = nb, split = 1, tail_tiles = 0;
    // SwiGLU tiles are split only on request (the decode entry, where gate|up is not kept): the training / log-prob passes keep the
    // unsplit epilogue, bit-identical to st_gemm_nt + st_swiglu_fwd
    if ((!SW || (sw_tail && gu == nullptr)) && g_tail_ws) {     // cost model of launch_tile (K-tile steps, fitted to kernel traces)
        const int ncu = st_num_cus(), nkt = K / 64, r = nb % ncu;
        const int64_t cap = g_tail_ws_bytes / ((int64_t)A4_BM * A4_BN * 4);
        int best = 1, best_cost = nkt + 4;
        for (int S = 2; S <= 8 && r > 0; ++S) {
            if ((int64_t)r * S > cap || S * 4 > nkt) break;
            const int cost = st_cdiv(r * S, ncu) * (st_cdiv(nkt, S) + 14) + 20;
            if (cost < best_cost) { best = S; best_cost = cost; }
        }
        if (best > 1) { full = nb - r; split = best; tail_tiles = r; }
    }
    hipLaunchKernelGGL(kern, dim3(full + tail_tiles * split), dim3(256), smem, s, A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, gu, ldgu, M, N, K, tiles_m,
                       tiles_n, g_tail_ws, full, split);
    if (tail_tiles) {
        if constexpr (SW)
            hipLaunchKernelGGL(gemm_a4_swiglu_finish_kernel, dim3(tail_tiles, 32), dim3(256), 0, s, g_tail_ws, split, full, Cb, ldc, M, N, tiles_m, tiles_n);
        else
            hipLaunchKernelGGL((gemm_a4_finish_kernel<HB, HR, OB, AC>), dim3(tail_tiles, 32), dim3(256), 0, s, g_tail_ws, split, full, bias, res, ldr, Cb, Cf,
                               ldc, M, N, tiles_m, tiles_n);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int st_gemm_asm4_debug(int dbg, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, int64_t ldc, int M, int N, int K, hipStream_t s) {
    constexpr int smem = 256 * 528;
    const int tiles_m = st_cdiv(M, A4_BM), tiles_n = st_cdiv(N, A4_BN);
#define A4DBG(D) { auto kern = gemm_nt4_kernel<false, false, true, false, false, D>; hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem, s, A, lda, B, ldb, (const uint16_t*)nullptr, (const uint16_t*)nullptr, (int64_t)0, Cb, (float*)nullptr, ldc, \
                           (uint16_t*)nullptr, (int64_t)0, M, N, K, tiles_m, tiles_n, D == 7 ? g_tail_ws : (float*)nullptr, tiles_m * tiles_n, 1); }
    switch (dbg) { case 1: A4DBG(1); break; case 2: A4DBG(2); break; case 3: A4DBG(3); break; case 4: A4DBG(4); break; case 5: A4DBG(5); break; case 6: A4DBG(6); break; case 7: if (!g_tail_ws || g_tail_ws_bytes < (int64_t)tiles_m * tiles_n * 32) return ST_EINVAL; A4DBG(7); break; default: return ST_EINVAL; }
#undef A4DBG
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// C = A B^T with the epilogue kinds of st_gemm_nt (exactly one of Cb / Cf; bias / residual only with Cb).  Row pitches must keep a
// 256-row operand tile within 4 GiB (buffer offsets are 32-bit): lda, ldb < 2^22 elements.
int st_gemm_asm4_dispatch(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res, int64_t ldr,
                          uint16_t* Cb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s) {
    if (lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    if (Cb) {
        if (bias && res) return launch_asm4<true, true, true, false, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
        if (bias) return launch_asm4<true, false, true, false, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
        if (res) return launch_asm4<false, true, true, false, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
        return launch_asm4<false, false, true, false, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
    }
    if (accumulate) return launch_asm4<false, false, false, true, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
    return launch_asm4<false, false, false, false, false>(A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, nullptr, 0, M, N, K, s);
}

// out[M,N] (bf16) = A[M,K] B[K,N], B contraction-major (the dX = dY W form of st_gemm_nn)
int st_gemm_asm4_nn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, int64_t ldc, int M, int N, int K, hipStream_t s) {
    if (lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    return launch_asm4<false, false, true, false, false, false, true>(A, lda, B, ldb, nullptr, nullptr, 0, Cb, nullptr, ldc, nullptr, 0, M, N, K, s);
}
// out_f32[M,N] (+)= A[K,M]^T B[K,N], both contraction-major (the dW = dY^T X form of st_gemm_tn)
int st_gemm_asm4_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s) {
    if (lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    if (accumulate) return launch_asm4<false, false, false, true, false, true, true>(A, lda, B, ldb, nullptr, nullptr, 0, nullptr, Cf, ldc, nullptr, 0, M, N, K, s);
    return launch_asm4<false, false, false, false, false, true, true>(A, lda, B, ldb, nullptr, nullptr, 0, nullptr, Cf, ldc, nullptr, 0, M, N, K, s);
}

// gate/up projection with the SwiGLU epilogue: m_out[M, I] = silu(A gate^T) * (A up^T), gu_out (optional) = bf16 gate | up
int st_gemm_asm4_swiglu(const uint16_t* A, int64_t lda, const uint16_t* gate_up_w, int64_t ldb, uint16_t* gu_out, int64_t ldgu, uint16_t* m_out,
                        int64_t ldm, int M, int I, int K, hipStream_t s, int split_tail) {
    if (lda >= (1 << 22) || ldb >= (1 << 22)) return ST_EINVAL;
    if (split_tail)                                              // the decode entry (257..512 rows): own symbol, K-split tail
        return launch_asm4<false, false, true, false, true, false, false, true>(A, lda, gate_up_w, ldb, nullptr, nullptr, 0, m_out, nullptr, ldm, gu_out, ldgu,
                                                                                M, I, K, s, true);
    return launch_asm4<false, false, true, false, true>(A, lda, gate_up_w, ldb, nullptr, nullptr, 0, m_out, nullptr, ldm, gu_out, ldgu, M, I, K, s, false);
}
