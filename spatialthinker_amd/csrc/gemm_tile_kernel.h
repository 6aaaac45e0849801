#pragma once
// gemm_tile_kernel.h — tile-shape / pipeline-depth family of the bf16 MFMA GEMM (C = A B^T), used (a) by the tuning
// harness tools/gemm_tune.py through st_gemm_nt_variant and (b) by st_gemm_nt for the shapes where a larger tile wins.
//
// Same building blocks as gemm.hip (LDS-DMA staging, XOR-swizzled lane-linear LDS image, MFMA 16x16x32 with swapped
// operands) generalised over:
//   BM x BN  output tile, WM x WN waves (each wave (BM/WM) x (BN/WN)), STAGES LDS ring slots of (BM+BN) x 64 bf16.
// STAGES == 2: one `vmcnt(0)` + barrier per K-tile (loads of tile t+1 fly during the MFMAs of tile t).
// STAGES == 3: counted `vmcnt(N)` + raw s_barrier: two K-tiles of LDS-DMA stay in flight across the barrier
//              (cdna_hip_programming.md §5 "Pipelining across barriers"); the wait counts only THIS wave's own loads,
//              the barrier publishes every wave's landed pieces.
// Why larger tiles: the 128x128x64 tile moves 32 KiB per 2.1 MFLOP = 64 flop/B, i.e. ~39 TB/s of L2->LDS traffic at the
// 2.5 PF MFMA peak — above the ~34 TB/s aggregate L2 bandwidth; 256x128 needs 29 TB/s, 256x256 19.5 TB/s.
#include "common.h"
#include <type_traits>
#ifndef ST_DMA_RUNS
#define ST_DMA_RUNS 1
#endif
#include <stdlib.h>

__device__ __forceinline__ void glds16t(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// Copy J of a wave's run of consecutive 1-KiB pieces: the 1024*J bytes go into the instruction's immediate offset (which the hardware
// adds to BOTH the global and the LDS address), so the run shares ONE M0 value; gsrc is the piece's true source address.
// AUX = 2: non-temporal ("nt") cache policy — for operands every workgroup reads exactly once (decode weights at one row tile):
// issued -> landed latency of the weight stream -18 % (MI355X_MICROARCH.md price list, "nt-weights")
template <int J, int AUX = 0> __device__ __forceinline__ void glds16_run(const void* gsrc, char* lds_run_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(gsrc) - J * 1024),
                                     (__attribute__((address_space(3))) void*)lds_run_base, 16, J * 1024, AUX);
}
template <int AUX = 0> __device__ __forceinline__ void glds16_run_j(int j, const void* gsrc, char* lds_run_base) {
    switch (j) {
        case 0: glds16_run<0, AUX>(gsrc, lds_run_base); break;
        case 1: glds16_run<1, AUX>(gsrc, lds_run_base); break;
        case 2: glds16_run<2, AUX>(gsrc, lds_run_base); break;
        default: glds16_run<3, AUX>(gsrc, lds_run_base); break;
    }
}
template <int AUX> __device__ __forceinline__ void glds16t_aux(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, AUX);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

// ST_GEMM_TRACE builds (tools/gemm_phase_trace.py, never the shipped library): every wave sums the shader-clock cycles it
// spends between fixed points of the K loop and leaves the sums in st_gemm_trace_ptr[workgroup][wave][8].
#ifdef ST_GEMM_TRACE
__device__ unsigned long long* st_gemm_trace_ptr = nullptr;
extern "C" int st_gemm_trace_set(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(st_gemm_trace_ptr), &buf, sizeof(buf)); }
#define TR_DECL unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = 0
#define TR_START() do { __builtin_amdgcn_sched_barrier(0); tr_prev = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TR_POINT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); tr_acc[i] += t_ - tr_prev; tr_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define TR_FLUSH() do { if (st_gemm_trace_ptr && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) st_gemm_trace_ptr[((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 8 + i_] = tr_acc[i_]; } } while (0)
#else
#define TR_DECL
#define TR_START()
#define TR_POINT(i)
#define TR_FLUSH()
#endif

// LDS-DMA from inline asm (M0 = wave-uniform LDS destination, restored afterwards): hipcc does not know a copy is in flight.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// LDS-DMA with a wave-uniform 64-bit base in SGPRs and a 32-bit lane offset: no address arithmetic on the VALU
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}

// Compile-time unrolled sched_group_barrier pattern (the builtin wants literal arguments): per slot one DS read, one MFMA,
// optionally one VMEM (LDS-DMA) issue, then the slot's remaining MFMAs.
template <int I, int SLOTS, int BASE, int EXTRA, int NVMEM>
__device__ __forceinline__ void pin_schedule() {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if constexpr (I < NVMEM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    constexpr int REST = BASE - 1 + (I < EXTRA ? 1 : 0);
    if constexpr (REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, REST, 0);
    if constexpr (I + 1 < SLOTS) pin_schedule<I + 1, SLOTS, BASE, EXTRA, NVMEM>();
}

// ND LDS reads merged evenly with NM MFMAs (Bresenham), NV LDS-DMA issues spread over the first reads.
template <int I, int ND, int NM, int NV>
__device__ __forceinline__ void pin_mix() {
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    constexpr int M_NOW = ((I + 1) * NM) / ND - (I * NM) / ND;
    if constexpr (M_NOW > 0) __builtin_amdgcn_sched_group_barrier(0x008, M_NOW, 0);
    constexpr int V_NOW = ((I + 1) * NV) / ND - (I * NV) / ND;
    if constexpr (V_NOW > 0) __builtin_amdgcn_sched_group_barrier(0x010, V_NOW, 0);
    if constexpr (I + 1 < ND) pin_mix<I + 1, ND, NM, NV>();
}

// Tail split (1 workgroup per CU tiles only): a grid of q * CUs + r tiles spends a whole round on its last r tiles.  The launch
// keeps the first `full_blocks` tiles as they are and cuts every remaining tile into `split` K-slices ("pieces", dispatched last):
// a piece leaves its fp32 accumulators in the workspace in fragment order, and a second launch of the same kernel (mode 2, one
// mode 2, no K loop) sums the slices in a fixed order and runs the regular epilogue; a tail tile is shared by `fin_sub`
// workgroups there (grid.y, each takes TN / fin_sub of every wave's column tiles) because one CU streams the slices of a whole
// tile far slower than the pieces produce them.  Deterministic.
struct TailArgs {
    float* ws;
    int full_blocks, split, mode, fin_sub;
};

// SWIGLU (decode MLP): B = [gate rows | up rows] (2N x K); the B tile interleaves 16 gate rows with the 16 matching up rows
// inside every wave's column slice, so a lane holds gate and up of the same output column in adjacent MFMA tiles and the
// epilogue writes act(gate) * up for BN/2 output columns — the (M, 2N) intermediate and the SwiGLU launch disappear.
// MIDBAR (2 stages only): the per-tile barrier sits BETWEEN the two k-steps of a tile instead of in front of it.  At that point
// every wave holds both k-steps' fragments of tile t in registers (slot t is free for the DMA of tile t+2) and tile t+1 has
// landed, so the fragment reads of tile t+1 run under the MFMAs of (t, k-step 1): no LDS read is ever exposed behind a barrier.
// SW8 (with SWIGLU): gate and up are interleaved at 8-column granularity INSIDE every 16-column MFMA tile (cols 0-7 gate j..j+7,
// cols 8-15 up j..j+7), so any WTN that is a multiple of 16 works (e.g. 80 = a 256x160 tile, 237 workgroups for the 7B MLP); the
// up values sit 32 lanes above their gate values and come down with one v_permlane32_swap per accumulator register.
// LEPI (256x256, 8 waves): LDS-staged epilogue.  The MFMA fragment layout gives a lane 4 consecutive columns of 16 different rows
// per instruction, so the direct epilogue moves C, the residual and the fp32 accumulate target as 8..16-byte pieces of 16 rows (32-byte
// segments: a residual + bias epilogue costs 41 us per round of 256 tiles, more than a third of a K = 3584 main loop).  Here the
// accumulators go through the (now idle) operand LDS in two 256 x 128-column passes of fp32 rows padded to 528 bytes (conflict-free
// for the 8-lane groups of ds_write_b128), and every lane then handles 8 consecutive columns of ONE row: bias / residual / C move as
// 16-byte (fp32 target: 32-byte) row-contiguous vectors, 512 contiguous bytes per row and instruction.
// AS / BS (256x256 MIDBAR tile only): operand A / B is stored CONTRACTION-MAJOR — A[k][m] (lda = pitch of a k-row) instead of A[m][k]
// — as the backward GEMMs find their operands in memory: dX = dY W reads W[n][k'] with the contraction index n as the row (BS),
// dW = dY^T X reads both dY[t][n] and X[t][k'] with the token index t as the row (AS + BS).  A K-tile of such an operand is staged as
// [64 k-rows][256 m] (512-byte rows) by LDS-DMA with the swizzle on the source address; an MFMA fragment (8 consecutive k of one m)
// is two ds_read_b64_tr_b16 (4 k each): no transposed copy of an operand ever exists in HBM.  32-byte chunk c (16 m) of k-row r sits
// at chunk position c ^ ((r & 3) | ((r >> 3) & 1) << 2): the 8 k-rows a 32-lane half of a transpose read touches use 8 distinct
// 32-byte slots of the 256-byte bank row.  hipcc has no memory operand for the transpose-read builtin and would drain every pending
// LDS-DMA in front of it (s_waitcnt vmcnt(0) in the middle of each tile), so in these variants the LDS-DMA is issued from inline asm
// (invisible to the compiler's wait insertion; the kernel's own counted waits order it) and placed by hand between MFMA chunks.
// PP (8 waves, 2 slots): PING-PONG schedule.  Waves w and w + 4 share a SIMD (tools/probes/wave_simd_map.hip).  With the mid-tile
// barrier both of them issue their MFMAs in the same phase and wait at the barrier in the same phase: a phase timer in wave 0
// (tools/gemm_phase_trace.py) shows 37 % of a K-tile parked at the barrier, the matrix pipe idle.  Here the K-tile has four slots with a
// barrier after each, and the two wave groups run the same program ONE SLOT APART, so in every slot one wave of a SIMD issues its 32
// MFMAs of a k-step alone while the other does its LDS fragment reads / LDS-DMA issues / waits:
//     group 0:  M0(t)  R0(t)  M1(t)  R1(t)  M0(t+1) ...        M_k = MFMAs of k-step k (fragments already in registers)
//     group 1:  R1(t-1) M0(t) R0(t)  M1(t)  R1(t)   ...        R0  = read fragments (t, k1);  R1 = issue DMA(t+2) into slot t, read (t+1, k0)
// Tile t+1 is published by the barrier that closes the slot in which group 0 runs M1(t) and group 1 R0(t): both wait for their own
// copies there.  Slot t is refilled from R1(t) on, after both groups' reads of (t, k1) have returned (lgkmcnt(0) before their barrier).
#define KMAJ_ANY(a, b) ((a) || (b))
template <int BM, int BN, int WM, int WN, int STAGES, bool HAS_BIAS, bool HAS_RES, bool OUT_BF16, bool ACCUM, bool SWIGLU = false,
          bool MIDBAR = false, bool SW8 = false, bool LEPI = false, bool AS = false, bool BS = false, bool PP = false, bool NTB = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_tile_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                                const uint16_t* __restrict__ B, int64_t ldb,
                                                                const uint16_t* __restrict__ bias,
                                                                const uint16_t* __restrict__ res, int64_t ldr,
                                                                uint16_t* __restrict__ Cb, float* __restrict__ Cf, int64_t ldc,
                                                                int M, int N, int K, int tiles_m, int tiles_n, int kt_per_split,
                                                                int64_t slab_stride, TailArgs tail) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;            // wave tile
    constexpr int TM = WTM / 16, TN = WTN / 16;            // MFMA tiles per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INST = BM / 8, B_INST = BN / 8;        // 1 KiB wave-instructions per operand tile
    constexpr int A_PER = (A_INST + NW - 1) / NW, B_PER = (B_INST + NW - 1) / NW;
    constexpr int PER_WAVE = A_PER + B_PER;                // LDS-DMA instructions a wave issues per K-tile (upper bound if uneven)
    constexpr bool EVEN_DMA = (A_INST % NW == 0) && (B_INST % NW == 0);
    static_assert(EVEN_DMA || !MIDBAR, "mid-tile barrier prologue counts DMA instructions");
    static_assert(!SWIGLU || ((SW8 || WTN % 32 == 0) && OUT_BF16 && !HAS_BIAS && !HAS_RES), "SwiGLU epilogue pairs 16-column MFMA tiles");
    static_assert(!SW8 || SWIGLU, "SW8 is a flavour of the SwiGLU epilogue");
    static_assert(!LEPI || (BM == 256 && BN == 256 && NW == 8 && !SWIGLU), "LDS-staged epilogue: 256x256 tile, 8 waves");
    static_assert(!(AS || BS) || (MIDBAR && BM == 256 && BN == 256 && NW == 8 && !SWIGLU && STAGES == 2), "contraction-major operands: 256x256 mid-barrier tile");
    static_assert(!PP || (MIDBAR && NW == 8 && STAGES == 2 && !KMAJ_ANY(AS, BS)), "ping-pong schedule: 8 waves, two LDS slots, row-major operands");
    constexpr bool KMAJ = AS || BS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    int piece = -1;
    if constexpr (!SWIGLU) {
        if (tail.mode == 2) bid = tail.full_blocks + blockIdx.x;
        else if (bid >= tail.full_blocks) { piece = bid - tail.full_blocks; bid = tail.full_blocks + piece / tail.split; }
    }
    // Split-K slab launches (decode shapes, grid.y = K-slices): tiles AND K-slices share ONE XCD-contiguous order (round 4).  With the
    // per-slice map below every XCD owned the same tile range in EVERY K-slice, i.e. it read the whole activation operand: 8 x 19 MB
    // per down-projection at 512 rows, 3x the weights' own traffic (profiles/r03_decode_pmc.md).  Here the work items are numbered
    // K-slice-major and each XCD takes a contiguous run of them, so an XCD touches one or two K-slices of A and both row tiles of a
    // weight tile still meet in its L2.
    int kslice = blockIdx.y;
    bool flat = false;
    if constexpr (!SWIGLU) flat = gridDim.y > 1 && tail.mode == 0 && piece < 0;
    if (flat) {
        const int Wt = nb * (int)gridDim.y, Lb = (int)blockIdx.y * nb + (int)blockIdx.x;
        const int xcd = Lb & 7, idx = Lb >> 3, q = Wt >> 3, r = Wt & 7;
        const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        kslice = item / nb;
        bid = item - kslice * nb;
    } else {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GM = (BM >= 256) ? 8 : 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = SWIGLU ? tn * (BN / 2) : tn * BN;      // SWIGLU: n0 = first OUTPUT column, N = output width

    auto stage = [&](int kt, char* dst) {
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int inst = EVEN_DMA ? wave * A_PER + j : j * NW + wave;    // uneven split: round-robin, counts differ by <= 1
            if (!EVEN_DMA && inst >= A_INST) break;
            const int p = inst * 64 + lane, r = p >> 3, cpos = p & 7, kc = cpos ^ ((r >> 1) & 7);
            int gr = m0 + r; gr = gr < M ? gr : M - 1;
            if constexpr (EVEN_DMA && A_PER <= 4 && ST_DMA_RUNS) glds16_run_j(j, A + (int64_t)gr * lda + kt * 64 + kc * 8, dst + wave * A_PER * 1024);
            else glds16t(A + (int64_t)gr * lda + kt * 64 + kc * 8, dst + inst * 1024);
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int inst = EVEN_DMA ? wave * B_PER + j : j * NW + wave;
            if (!EVEN_DMA && inst >= B_INST) break;
            const int p = inst * 64 + lane, r = p >> 3, cpos = p & 7, kc = cpos ^ ((r >> 1) & 7);
            int gr;
            if (SW8) {
                const int j8 = (r >> 4) * 8 + (r & 7);
                gr = n0 + j8; gr = gr < N ? gr : N - 1;
                if (r & 8) gr += N;
            } else if (SWIGLU) {
                const int within = r % WTN;
                const int j = (r / WTN) * (WTN / 2) + (within >> 5) * 16 + (within & 15);
                gr = n0 + j; gr = gr < N ? gr : N - 1;
                if (within & 16) gr += N;                    // the up-projection rows follow the N gate rows
            } else {
                gr = n0 + r; gr = gr < N ? gr : N - 1;
            }
            constexpr int BAUX = NTB ? 2 : 0;                // NTB: the B operand (weights) streams with the non-temporal policy
            if constexpr (EVEN_DMA && B_PER <= 4 && ST_DMA_RUNS) glds16_run_j<BAUX>(j, B + (int64_t)gr * ldb + kt * 64 + kc * 8, dst + A_BYTES + wave * B_PER * 1024);
            else glds16t_aux<BAUX>(B + (int64_t)gr * ldb + kt * 64 + kc * 8, dst + A_BYTES + inst * 1024);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K (decode shapes): grid.y slices of kt_per_split K-tiles, fp32 partial slab per slice (summed by a finish kernel)
    int kt_begin = kslice * kt_per_split;
    int nk = min(K / 64 - kt_begin, kt_per_split);
    int ni_lo = 0, ni_hi = TN;                               // column tiles this workgroup finishes (all of them unless mode 2)
    if constexpr (!SWIGLU) {
        if (tail.mode == 2) { kt_begin = 0; ni_lo = blockIdx.y * (TN / tail.fin_sub); ni_hi = ni_lo + TN / tail.fin_sub; }
        if (piece >= 0) {
            const int per = (K / 64 + tail.split - 1) / tail.split;
            kt_begin = (piece % tail.split) * per;
            nk = min(K / 64 - kt_begin, per);
        }
        if (tail.mode == 2) nk = 0;
    }
    A += AS ? (int64_t)kt_begin * 64 * lda : (int64_t)kt_begin * 64;
    B += BS ? (int64_t)kt_begin * 64 * ldb : (int64_t)kt_begin * 64;
    if (!OUT_BF16 && !SWIGLU && tail.mode != 2) Cf += kslice * slab_stride;
    if constexpr (!KMAJ) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) if (s < nk) stage(s, smem + s * STAGE);
    }

    const int frow = lane & 15, fk = lane >> 4;
    int slot = 0;
    TR_DECL;
    TR_START();
    auto tile_body = [&](int kt, auto prefetch_tag) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value;
        // tile kt must have landed; up to STAGES-2 younger tiles may stay in flight
        if (STAGES == 2 || !PREFETCH) wait_vmcnt<0>();          // tail tiles: nothing younger is in flight to count
        else if constexpr (EVEN_DMA) wait_vmcnt<(STAGES - 2) * PER_WAVE>();
        else {                                               // round-robin split: this wave issued PER_WAVE or PER_WAVE-1 copies
            static_assert(EVEN_DMA || STAGES == 3, "uneven LDS-DMA split is wired for the 3-slot ring");
            const int mine = (wave < A_INST % NW || A_INST % NW == 0 ? A_PER : A_PER - 1) + (wave < B_INST % NW || B_INST % NW == 0 ? B_PER : B_PER - 1);
            if (mine == PER_WAVE) wait_vmcnt<PER_WAVE>(); else wait_vmcnt<PER_WAVE - 1>();
        }
        TR_POINT(0);                                         // 0: waiting for this wave's copies of tile kt
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                       // keep LDS reads / DMA issue below the barrier
        TR_POINT(1);                                         // 1: barrier (the other waves' copies / their reads of the slot)
        const char* la = smem + slot * STAGE;
        const char* lb = la + A_BYTES;
        // fragment double-buffering: the LDS reads of k-step s+1 are issued BEFORE the MFMAs of k-step s, so the ~130-cycle
        // LDS latency hides under 2*TM*TN MFMAs instead of being exposed in front of every MFMA group
        bf16x8 af[2][TM], bfr[2][TN];
        auto load_frags = [&](int s, bf16x8 (&a_)[TM], bf16x8 (&b_)[TN]) {
            const int kc = s * 4 + fk;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ra = wm * WTM + i * 16 + frow;
                a_[i] = *reinterpret_cast<const bf16x8*>(la + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int rb = wn * WTN + i * 16 + frow;
                b_[i] = *reinterpret_cast<const bf16x8*>(lb + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
            }
        };
        load_frags(0, af[0], bfr[0]);
        __builtin_amdgcn_sched_barrier(0);
#ifdef ST_GEMM_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TR_POINT(2);                                         // 2: fragment reads of k-step 0 issued and returned
#endif
        // the LDS-DMA issues of the next K-tile (~100 cycles of issue each) are spread between the MFMAs of k-step 0 as well,
        // instead of sitting in front of them with the matrix pipe idle
        if constexpr (PREFETCH) {
            int ns = slot + STAGES - 1; ns = ns >= STAGES ? ns - STAGES : ns;
            stage(kt + STAGES - 1, smem + ns * STAGE);
        }
        load_frags(1, af[1], bfr[1]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][ni], af[0][mi], acc[ni][mi], 0, 0, 0);
        // pin the interleave: one LDS fragment read of k-step 1 per two MFMAs of k-step 0 (hipcc otherwise re-serialises
        // the reads in front of their consumers to save registers)
        {
            constexpr int SLOTS = TM + TN, BASE = (TM * TN) / SLOTS, EXTRA = TM * TN - BASE * SLOTS;
            pin_schedule<0, SLOTS, BASE, EXTRA, PREFETCH ? PER_WAVE : 0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        TR_POINT(3);                                         // 3: DMA issues + MFMAs of k-step 0 + reads of k-step 1 (all issued)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][ni], af[1][mi], acc[ni][mi], 0, 0, 0);
        TR_POINT(4);                                         // 4: MFMAs of k-step 1 issued
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    };
    constexpr bool BIG4 = MIDBAR && NW == 4 && TM == 8 && TN == 8;
    if constexpr (!MIDBAR) {
        int kt = 0;
        for (; kt + STAGES - 1 < nk; ++kt) tile_body(kt, std::true_type{});
        for (; kt < nk; ++kt) tile_body(kt, std::false_type{});
    } else if constexpr (BIG4) {
        // One wave per SIMD, 128 x 128 per wave: the 64 accumulator tiles fill the 256 AGPRs, the LDS fragment traffic per MFMA is
        // half that of the 8-wave layout.  Nothing else runs on the SIMD, so the schedule is written out by hand: the MFMAs are
        // volatile asm with the accumulator tied in place ("+a": hipcc otherwise rotates the 64 tuples through VGPRs, 280 moves
        // per K-tile), and the LDS reads / LDS-DMA issues sit between them in source order (volatile asm pins memory operations).
        static_assert(STAGES == 2 && EVEN_DMA, "two LDS slots, even DMA split");
        bf16x8 af[2][TM], bfr[2][TN];
        const int ra0 = wm * WTM + frow, rb0 = wn * WTN + frow;
        int offA[2], offB[2];                                // rows i*16 further down share the swizzle: + i * 2048 bytes
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int kc = s2 * 4 + fk;
            offA[s2] = ra0 * 128 + ((kc ^ ((ra0 >> 1) & 7)) << 4);
            offB[s2] = A_BYTES + rb0 * 128 + ((kc ^ ((rb0 >> 1) & 7)) << 4);
        }
        // LDS-DMA sources: uniform 64-bit base (advanced per K-tile in SGPRs) + one 32-bit lane offset per copy, relative to the
        // tile's first row — 16 VGPRs stay live instead of 16 address pairs recomputed or spilled
        const char* Abase = reinterpret_cast<const char*>(A + (int64_t)m0 * lda);
        const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
        uint32_t aoff[A_PER], boff[B_PER];
#pragma unroll
        for (int j = 0; j < A_PER; ++j) {
            const int inst = wave * A_PER + j, p2 = inst * 64 + lane, r = p2 >> 3, kc = (p2 & 7) ^ ((r >> 1) & 7);
            const int rr = min(r, M - 1 - m0);
            aoff[j] = (uint32_t)rr * (uint32_t)(lda * 2) + kc * 16;
        }
#pragma unroll
        for (int j = 0; j < B_PER; ++j) {
            const int inst = wave * B_PER + j, p2 = inst * 64 + lane, r = p2 >> 3, kc = (p2 & 7) ^ ((r >> 1) & 7);
            const int rr = min(r, N - 1 - n0);
            boff[j] = (uint32_t)rr * (uint32_t)(ldb * 2) + kc * 16;
        }
        auto dma_one = [&](int kt, char* dst, int j) {       // j < A_PER: A copy j, else B copy j - A_PER
            if (j < A_PER) glds16t(Abase + (int64_t)kt * 128 + aoff[j], dst + (wave * A_PER + j) * 1024);
            else glds16t(Bbase + (int64_t)kt * 128 + boff[j - A_PER], dst + A_BYTES + (wave * B_PER + j - A_PER) * 1024);
        };
        auto rd = [&](const char* base, int s2, int i) {      // i < TM: A fragment i, else B fragment i - TM
            if (i < TM) af[s2][i] = *reinterpret_cast<const bf16x8*>(base + offA[s2] + i * 2048);
            else bfr[s2][i - TM] = *reinterpret_cast<const bf16x8*>(base + offB[s2] + (i - TM) * 2048);
        };
        auto mma = [&](int s2, int idx) {
            const int ni = idx / TM, mi = idx % TM;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ni][mi]) : "v"(bfr[s2][ni]), "v"(af[s2][mi]));
        };
        if (nk > 1) { stage(1, smem + STAGE); wait_vmcnt<PER_WAVE>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nk > 0) {
#pragma unroll
            for (int i = 0; i < TM + TN; ++i) rd(smem, 0, i);
        }
        auto tile = [&](int kt, auto next_tag, auto dma_tag) {
            constexpr bool HAS_NEXT = decltype(next_tag)::value, HAS_DMA = decltype(dma_tag)::value;
            char* cur = smem + (kt & 1) * STAGE;
            const char* nxt = smem + ((kt + 1) & 1) * STAGE;
            // phase A: MFMAs of k-step 0; the 16 fragment reads of k-step 1 go out under the first half
#pragma unroll
            for (int idx = 0; idx < TM * TN; ++idx) {
                mma(0, idx);
                if ((idx & 1) == 1 && idx / 2 < TM + TN) rd(cur, 1, idx / 2);
            }
            if constexpr (HAS_NEXT) {
                wait_vmcnt<0>();                             // tile kt+1 has landed (issued one tile ago)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of slot kt have RETURNED before anyone may refill it
                __builtin_amdgcn_s_barrier();                // ... for every wave, and every wave holds tile kt in registers
                asm volatile("" ::: "memory");
            }
            // phase B: MFMAs of k-step 1; DMA of tile kt+2 into the slot just vacated + fragment reads of (kt+1, k-step 0)
#pragma unroll
            for (int idx = 0; idx < TM * TN; ++idx) {
                mma(1, idx);
                if constexpr (HAS_NEXT) {
                    if ((idx & 3) == 1 && idx / 4 < TM + TN) rd(nxt, 0, idx / 4);
                    if constexpr (HAS_DMA) { if ((idx & 3) == 3 && idx / 4 < PER_WAVE) dma_one(kt + 2, cur, idx / 4); }
                }
            }
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { tile(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) tile(kt, std::false_type{}, std::false_type{});
        // MFMA results -> epilogue reads: the hazard checker cannot see into asm, and a bare s_nop statement does not stop hipcc from
        // scheduling the first v_accvgpr_read of the epilogue right behind the last MFMA (seen once in ~6 runs as 4 stale rows of the
        // last accumulator).  Every accumulator therefore passes THROUGH a wait: row ni of the tiles per statement, oldest first, so
        // the most recent MFMAs have 64 wait states behind them when their results become readable.
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
            asm volatile("s_nop 7" : "+a"(acc[ni][0]), "+a"(acc[ni][1]), "+a"(acc[ni][2]), "+a"(acc[ni][3]), "+a"(acc[ni][4]), "+a"(acc[ni][5]),
                         "+a"(acc[ni][6]), "+a"(acc[ni][7]));
    } else if constexpr (KMAJ) {
        // ---- contraction-major operands: mid-tile-barrier schedule with hand-placed asm LDS-DMA (see the template comment)
        const uint32_t smem32 = (uint32_t)(uintptr_t)smem;
        // one 1-KiB piece: j < A_PER -> A piece wave*A_PER + j, else B piece wave*B_PER + (j - A_PER)
        auto dma_piece = [&](int kt, int slot_, int j) {
            const bool isA = j < A_PER;
            const int inst = isA ? wave * A_PER + j : wave * B_PER + (j - A_PER);
            const uint32_t dst = smem32 + slot_ * STAGE + (isA ? 0 : A_BYTES) + inst * 1024;
            const uint16_t* X = isA ? A : B;
            const int64_t ldx = isA ? lda : ldb;
            const int c0 = isA ? m0 : n0, C = isA ? M : N;
            if ((isA && AS) || (!isA && BS)) {              // [64 k][256 m] image: this instruction = k-rows 2*inst, 2*inst + 1
                const int kr = inst * 2 + (lane >> 5), sl = lane & 31;
                const int g = (kr & 3) | (((kr >> 3) & 1) << 2);
                int col = c0 + (((sl >> 1) ^ g) << 4) + (sl & 1) * 8;
                col = col < C - 8 ? col : C - 8;            // columns past the edge compute garbage that is never stored
                glds16_asm(X + (int64_t)(kt * 64 + kr) * ldx + col, dst);
            } else {                                        // [256 rows][64 k] image, as in the NT kernel
                const int p2 = inst * 64 + lane, r = p2 >> 3, kc = (p2 & 7) ^ ((r >> 1) & 7);
                int gr = c0 + r; gr = gr < C ? gr : C - 1;
                glds16_asm(X + (int64_t)gr * ldx + kt * 64 + kc * 8, dst);
            }
        };
        bf16x8 af[2][TM], bfr[2][TN];
        const int tr_g = ((lane >> 2) & 3) | (((lane >> 4) & 1) << 2);
        const int tr_row = (lane >> 4) * 8 + ((lane & 15) >> 2);
        auto tr_frag = [&](const char* img, int s2, int chunk) {   // two transpose reads: k +0..3 and +4..7 of column m = chunk*16 + (lane & 15)
            const char* p0 = img + (s2 * 32 + tr_row) * 512 + ((chunk ^ tr_g) << 5) + (lane & 3) * 8;
            const s16x4_t lo = st_lds_tr16(p0), hi = st_lds_tr16(p0 + 4 * 512);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto rd_a = [&](const char* la, int s2, int i) {
            if constexpr (AS) af[s2][i] = tr_frag(la, s2, wm * (WTM / 16) + i);
            else { const int ra = wm * WTM + i * 16 + frow, kc = s2 * 4 + fk; af[s2][i] = *reinterpret_cast<const bf16x8*>(la + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4)); }
        };
        auto rd_b = [&](const char* la, int s2, int i) {
            const char* lb = la + A_BYTES;
            if constexpr (BS) bfr[s2][i] = tr_frag(lb, s2, wn * (WTN / 16) + i);
            else { const int rb = wn * WTN + i * 16 + frow, kc = s2 * 4 + fk; bfr[s2][i] = *reinterpret_cast<const bf16x8*>(lb + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4)); }
        };
        static_assert(TM == 4 && TN == 8 && PER_WAVE == 8, "chunking below: 4 chunks of (1 A + 2 B fragments, 8 MFMAs, 2 DMA pieces)");
        constexpr int RD_CHUNK = (AS ? 2 : 1) + (BS ? 4 : 2);       // LDS read instructions per chunk
        // chunk c of a k-step: fragments A[c], B[2c], B[2c+1] of the NEXT k-step are read while the 8 MFMAs of B tiles 2c, 2c+1 run
        auto chunk = [&](const char* rd_img, int rd_s, bool do_rd, int mm_s, int c) {
            if (do_rd) { rd_a(rd_img, rd_s, c); rd_b(rd_img, rd_s, 2 * c); rd_b(rd_img, rd_s, 2 * c + 1); }
#pragma unroll
            for (int ni = 2 * c; ni < 2 * c + 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[mm_s][ni], af[mm_s][mi], acc[ni][mi], 0, 0, 0);
            if (do_rd) pin_mix<0, RD_CHUNK, 8, 0>(); 
        };
        // prologue: tiles 0 and 1 in flight, fragments of (0, k-step 0) in registers
        if (nk > 0) { for (int j = 0; j < PER_WAVE; ++j) dma_piece(0, 0, j); }
        if (nk > 1) { for (int j = 0; j < PER_WAVE; ++j) dma_piece(1, 1, j); wait_vmcnt<PER_WAVE>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nk > 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { rd_a(smem, 0, c); rd_b(smem, 0, 2 * c); rd_b(smem, 0, 2 * c + 1); }
        }
        auto tile = [&](int kt, auto next_tag, auto dma_tag) {
            constexpr bool HAS_NEXT = decltype(next_tag)::value, HAS_DMA = decltype(dma_tag)::value;
            const int sc = kt & 1;
            const char* cur = smem + sc * STAGE;
            const char* nxt = smem + (sc ^ 1) * STAGE;
            // phase A: MFMAs of k-step 0 with the reads of k-step 1 in their shadow
#pragma unroll
            for (int c = 0; c < 4; ++c) chunk(cur, 1, true, 0, c);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HAS_NEXT) {
                wait_vmcnt<0>();                             // tile kt+1 (issued one tile ago) has landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of slot kt have returned before anyone refills it
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // phase B: MFMAs of k-step 1; reads of (kt+1, k-step 0) and the DMA of tile kt+2 into the slot just vacated, 2 pieces per chunk
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                chunk(nxt, 0, HAS_NEXT, 1, c);
                if constexpr (HAS_DMA) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma_piece(kt + 2, sc, 2 * c); dma_piece(kt + 2, sc, 2 * c + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { tile(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) tile(kt, std::false_type{}, std::false_type{});
    } else {
        static_assert(!MIDBAR || STAGES == 2, "mid-tile barrier schedule uses exactly two LDS slots");
        bf16x8 af[2][TM], bfr[2][TN];
        auto load_frags = [&](const char* la, int s, bf16x8 (&a_)[TM], bf16x8 (&b_)[TN]) {
            const char* lb = la + A_BYTES;
            const int kc = s * 4 + fk;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ra = wm * WTM + i * 16 + frow;
                a_[i] = *reinterpret_cast<const bf16x8*>(la + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int rb = wn * WTN + i * 16 + frow;
                b_[i] = *reinterpret_cast<const bf16x8*>(lb + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
            }
        };
        auto mfmas = [&](bf16x8 (&a_)[TM], bf16x8 (&b_)[TN]) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_[ni], a_[mi], acc[ni][mi], 0, 0, 0);
        };
        constexpr int SLOTS = TM + TN, BASE = (TM * TN) / SLOTS, EXTRA = TM * TN - BASE * SLOTS;
        if constexpr (PP) {
            const int grp = wave >> 2;                       // 0: waves 0-3, 1: their SIMD partners 4-7
            // LDS-DMA sources: wave-uniform 64-bit base (advanced per K-tile in SGPRs) + one lane-constant 32-bit offset per copy, so an
            // issue costs no VALU instruction (plain VALU work of the partner wave takes issue slots from the MFMA wave)
            const char* Abase = reinterpret_cast<const char*>(A + (int64_t)m0 * lda);
            const char* Bbase = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
            uint32_t aoff[A_PER], boff[B_PER];
#pragma unroll
            for (int j = 0; j < A_PER; ++j) {
                const int inst = wave * A_PER + j, p2 = inst * 64 + lane, r = p2 >> 3, kc = (p2 & 7) ^ ((r >> 1) & 7);
                aoff[j] = (uint32_t)min(r, M - 1 - m0) * (uint32_t)(lda * 2) + kc * 16;
            }
#pragma unroll
            for (int j = 0; j < B_PER; ++j) {
                const int inst = wave * B_PER + j, p2 = inst * 64 + lane, r = p2 >> 3, kc = (p2 & 7) ^ ((r >> 1) & 7);
                boff[j] = (uint32_t)min(r, N - 1 - n0) * (uint32_t)(ldb * 2) + kc * 16;
            }
            const uint32_t smem32 = (uint32_t)(uintptr_t)smem;
            // copy j of tile kt_ (j < A_PER: A piece, else B piece) into slot_
            auto dma_pp = [&](int kt_, int slot_, int j) {
                const uint32_t dst = smem32 + slot_ * STAGE;
                if (j < A_PER) glds16_saddr(Abase + (int64_t)kt_ * 128, aoff[j], dst + (wave * A_PER + j) * 1024);
                else glds16_saddr(Bbase + (int64_t)kt_ * 128, boff[j - A_PER], dst + A_BYTES + (wave * B_PER + j - A_PER) * 1024);
            };
            constexpr int DMA_R1 = PER_WAVE / 2;             // copies issued in R1(kt); the rest ride between the MFMAs of M0(kt+1)
            auto mfmas_dma = [&](bf16x8 (&a_)[TM], bf16x8 (&b_)[TN], int kt_, int slot_, bool on) {
                // MFMAs in source order with one LDS-DMA issue after every (TM*TN / (PER_WAVE - DMA_R1))-th of them: a burst of issues
                // would park the wave (in-order issue) behind the texture addresser while the matrix pipe drains
                constexpr int NREST = PER_WAVE - DMA_R1, EVERY = (TM * TN) / NREST;
#pragma unroll
                for (int idx = 0; idx < TM * TN; ++idx) {
                    const int ni = idx / TM, mi = idx % TM;
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ni][mi]) : "v"(b_[ni]), "v"(a_[mi]));
                    if (idx % EVERY == EVERY / 2 && idx / EVERY < NREST && on) dma_pp(kt_, slot_, DMA_R1 + idx / EVERY);
                }
            };
            if (nk > 1) stage(1, smem + STAGE);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (nk > 0) load_frags(smem, 0, af[0], bfr[0]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grp == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }          // group 1 runs one slot behind
            TR_START();
            for (int kt = 0; kt < nk; ++kt) {
                const char* cur = smem + (kt & 1) * STAGE;
                const char* nxt = smem + ((kt + 1) & 1) * STAGE;
                const bool has_next = kt + 1 < nk;
                // ---- M0: MFMAs of k-step 0, alone on the SIMD; the second half of tile kt+1's copies (slot kt-1 is free since R1(kt-1))
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(2);
                mfmas_dma(af[0], bfr[0], kt + 1, (kt + 1) & 1, kt >= 1 && has_next);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                TR_POINT(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                TR_POINT(1);
                // ---- R0: fragments of (kt, k-step 1); group 1 also waits here for its copies of tile kt+1
                load_frags(cur, 1, af[1], bfr[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (grp == 1 && has_next) wait_vmcnt<0>();
                TR_POINT(2);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                TR_POINT(3);
                // ---- M1: MFMAs of k-step 1; group 0 waits for its copies of tile kt+1 behind them
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(2);
                mfmas(af[1], bfr[1]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (grp == 0 && has_next) wait_vmcnt<0>();
                TR_POINT(4);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                TR_POINT(5);
                // ---- R1: refill slot kt with tile kt+2 (every wave's reads of it have returned), fragments of (kt+1, k-step 0)
                if (has_next) {
                    if (kt + 2 < nk) {
#pragma unroll
                        for (int j = 0; j < DMA_R1; ++j) dma_pp(kt + 2, kt & 1, j);
                    }
                    load_frags(nxt, 0, af[0], bfr[0]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                TR_POINT(6);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                TR_POINT(7);
            }
            if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        } else {
        // prologue: tiles 0 and 1 in flight (tile 0 was issued above), fragments of (0, k-step 0) in registers
        if (nk > 1) { stage(1, smem + STAGE); wait_vmcnt<PER_WAVE>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (nk > 0) load_frags(smem, 0, af[0], bfr[0]);
        TR_START();
        auto tile = [&](int kt, auto next_tag, auto dma_tag) {
            constexpr bool HAS_NEXT = decltype(next_tag)::value, HAS_DMA = decltype(dma_tag)::value;
            const char* cur = smem + (kt & 1) * STAGE;
            const char* nxt = smem + ((kt + 1) & 1) * STAGE;
            // phase A: MFMAs of k-step 0 with the reads of k-step 1 in their shadow
            TR_POINT(4);                                     // 4: phase B of the previous tile (MFMAs k-step 1, reads, DMA issues)
            load_frags(cur, 1, af[1], bfr[1]);
            mfmas(af[0], bfr[0]);
            // the reads are spread over the first 5/8 of the phase's MFMAs so that the last of them has time to return before the
            // lgkmcnt wait in front of the barrier (all reads up front measured 3-8 % slower: LDS port contention with the DMA)
            constexpr int FRONT = (TM * TN * 5) / 8;          // reads spread over the first 5/8 of the MFMAs
            constexpr int FBASE = FRONT / SLOTS, FEXTRA = FRONT - FBASE * SLOTS;
            pin_schedule<0, SLOTS, FBASE, FEXTRA, 0>();
            if constexpr (TM * TN > FRONT) __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - FRONT, 0);
            __builtin_amdgcn_sched_barrier(0);
            TR_POINT(0);                                     // 0: phase A issued (MFMAs of k-step 0 + reads of k-step 1)
            if constexpr (HAS_NEXT) {
                wait_vmcnt<0>();                             // tile kt+1 (issued one tile ago) has landed
                TR_POINT(1);                                 // 1: waiting for this wave's copies of tile kt+1
                // the fragment reads of slot kt must have RETURNED (not merely been issued) before another wave's LDS-DMA may refill
                // the slot; the k-step-1 MFMAs behind the barrier need them anyway, so the wait costs nothing here
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                TR_POINT(2);                                 // 2: fragment reads returned
                __builtin_amdgcn_s_barrier();                // ... for every wave, and every wave is done reading slot kt
                asm volatile("" ::: "memory");
                TR_POINT(3);                                 // 3: barrier
                if constexpr (HAS_DMA) stage(kt + 2, smem + (kt & 1) * STAGE);
                load_frags(nxt, 0, af[0], bfr[0]);           // phase B: reads of (kt+1, k-step 0) under the MFMAs of (kt, k-step 1)
            }
            mfmas(af[1], bfr[1]);
            if constexpr (HAS_NEXT) pin_schedule<0, SLOTS, BASE, EXTRA, HAS_DMA ? PER_WAVE : 0>();
            __builtin_amdgcn_sched_barrier(0);
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { tile(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) tile(kt, std::false_type{}, std::false_type{});
        }
    }

    TR_FLUSH();
    if constexpr (!SWIGLU) {
        if (piece >= 0) {                                    // K-slice of a tail tile: accumulators to the workspace, fragment order
            float4* wp = reinterpret_cast<float4*>(tail.ws) + ((int64_t)(piece * NW + wave) * (TN * TM)) * 64 + lane;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    wp[(ni * TM + mi) * 64] = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
            return;
        }
        if (tail.mode == 2) {                                // sum the slices of this tail tile, then the regular epilogue
            for (int sp = 0; sp < tail.split; ++sp) {
                const float4* wp = reinterpret_cast<const float4*>(tail.ws) + ((int64_t)((blockIdx.x * tail.split + sp) * NW + wave) * (TN * TM)) * 64 + lane;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        if (ni < ni_lo || ni >= ni_hi) continue;
                        const float4 v = wp[(ni * TM + mi) * 64];
                        acc[ni][mi][0] += v.x; acc[ni][mi][1] += v.y; acc[ni][mi][2] += v.z; acc[ni][mi][3] += v.w;
                    }
            }
        }
    }
    if constexpr (LEPI) {
        if (tail.mode != 2) {                                // whole tiles; the finish launch of split tail tiles has no LDS: direct path
            constexpr int ROWB = 128 * 4 + 16;               // fp32 row of one pass + 16 bytes: rows 4 banks apart
            __syncthreads();                                 // every wave is done reading the operand slots; no DMA is in flight
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int nl = 0; nl < TN / 2; ++nl)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        const int row = wm * WTM + mi * 16 + (lane & 15), col = wn * (WTN / 2) + nl * 16 + (lane >> 4) * 4;
                        *reinterpret_cast<f32x4*>(smem + row * ROWB + col * 4) = acc[p * (TN / 2) + nl][mi];
                    }
                __syncthreads();
                const int t = threadIdx.x, c8 = (t & 15) * 8;
                const int n = n0 + (c8 >> 6) * WTN + p * (WTN / 2) + (c8 & 63);
                float bvals[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const bool ncols = n + 7 < N;
                if constexpr (HAS_BIAS) {
                    if (ncols && (reinterpret_cast<uintptr_t>(bias + n) & 15) == 0) unpack8(*reinterpret_cast<const uint4*>(bias + n), bvals);
                    else for (int r = 0; r < 8; ++r) bvals[r] = n + r < N ? bf2f(bias[n + r]) : 0.f;
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = it * 32 + (t >> 4), m = m0 + row;
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
                if (p == 0) __syncthreads();                 // the second pass overwrites the image
            }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * WTM + mi * 16 + (lane & 15);
        if (m >= M) continue;
        if constexpr (SW8) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {               // lanes 0..31 hold gate, lanes 32..63 the matching up value
                    const uint32_t mine = __float_as_uint(acc[ni][mi][r]);
                    auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                    const float g = bfround(acc[ni][mi][r]), u = bfround(__uint_as_float(sw[1]));
                    o[r] = f2bf(bfround(g * sigmoidf_(g)) * u);
                }
                const int n = n0 + (wn * (WTN / 16) + ni) * 8 + (lane >> 4) * 4;
                if (lane < 32 && n < N) {
                    uint16_t* cp = Cb + (int64_t)m * ldc + n;
                    if (n + 3 < N && ((ldc & 3) == 0)) {
                        uint2 w;
                        w.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                        w.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                        *reinterpret_cast<uint2*>(cp) = w;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = o[r];
                    }
                }
            }
            continue;
        }
        if constexpr (SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < TN; ni += 2) {
                const int n = n0 + wn * (WTN / 2) + (ni / 2) * 16 + (lane >> 4) * 4;
                if (n >= N) continue;
                uint16_t o[4], gq[4], uq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {               // same roundings as the unfused path: bf16 gate/up, bf16 act(gate)
                    gq[r] = f2bf(acc[ni][mi][r]); uq[r] = f2bf(acc[ni + 1][mi][r]);
                    const float g = bf2f(gq[r]), u = bf2f(uq[r]);
                    o[r] = f2bf(bfround(g * sigmoidf_(g)) * u);
                }
                if (Cf) {                                    // training: keep gate|up (N + N columns) for the backward; ld in `ldr`
                    uint16_t* gp = reinterpret_cast<uint16_t*>(Cf) + (int64_t)m * ldr + n;
                    if (n + 3 < N && ((ldr & 3) == 0) && ((N & 3) == 0)) {
                        uint2 w;
                        w.x = (uint32_t)gq[0] | ((uint32_t)gq[1] << 16); w.y = (uint32_t)gq[2] | ((uint32_t)gq[3] << 16);
                        *reinterpret_cast<uint2*>(gp) = w;
                        w.x = (uint32_t)uq[0] | ((uint32_t)uq[1] << 16); w.y = (uint32_t)uq[2] | ((uint32_t)uq[3] << 16);
                        *reinterpret_cast<uint2*>(gp + N) = w;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (n + r < N) { gp[r] = gq[r]; gp[N + r] = uq[r]; }
                    }
                }
                uint16_t* cp = Cb + (int64_t)m * ldc + n;
                if (n + 3 < N && ((ldc & 3) == 0)) {
                    uint2 w;
                    w.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                    w.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                    *reinterpret_cast<uint2*>(cp) = w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = o[r];
                }
            }
            continue;
        }
        // bias / residual first, as 8-byte vectors and for all column tiles of this row at once: one memory round trip per row of
        // MFMA tiles instead of four 2-byte loads per tile (59 -> 41 us of fixed cost per round of 256 tiles).  Wider batches
        // measured slower (all 32 residual vectors of a wave at once: o-proj 305 -> 323 us), and so did batching the fp32
        // accumulate reads ahead of their stores (36 -> 52 us per round).
        uint2 rv[TN], bv[TN];
        if constexpr (HAS_BIAS) {
            const bool vec = (reinterpret_cast<uintptr_t>(bias) & 7) == 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * WTN + ni * 16 + (lane >> 4) * 4;
                bv[ni] = make_uint2(0u, 0u);
                if (n >= N || ni < ni_lo || ni >= ni_hi) continue;
                if (n + 3 < N && vec) bv[ni] = *reinterpret_cast<const uint2*>(bias + n);
                else {
                    uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) e[r] = bias[n + r];
                    bv[ni] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
                }
            }
        }
        if constexpr (HAS_RES) {
            const bool vec = ((ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(res) & 7) == 0);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * WTN + ni * 16 + (lane >> 4) * 4;
                rv[ni] = make_uint2(0u, 0u);
                if (n >= N || ni < ni_lo || ni >= ni_hi) continue;
                const uint16_t* rp = res + (int64_t)m * ldr + n;
                if (n + 3 < N && vec) rv[ni] = *reinterpret_cast<const uint2*>(rp);
                else {
                    uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) e[r] = rp[r];
                    rv[ni] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
                }
            }
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * WTN + ni * 16 + (lane >> 4) * 4;
            if (n >= N || ni < ni_lo || ni >= ni_hi) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            const bool full = (n + 3 < N);
            if (HAS_BIAS) {
                v[0] += bf2f((uint16_t)(bv[ni].x & 0xffffu)); v[1] += bf2f((uint16_t)(bv[ni].x >> 16));
                v[2] += bf2f((uint16_t)(bv[ni].y & 0xffffu)); v[3] += bf2f((uint16_t)(bv[ni].y >> 16));
            }
            if (HAS_RES) {
                v[0] += bf2f((uint16_t)(rv[ni].x & 0xffffu)); v[1] += bf2f((uint16_t)(rv[ni].x >> 16));
                v[2] += bf2f((uint16_t)(rv[ni].y & 0xffffu)); v[3] += bf2f((uint16_t)(rv[ni].y >> 16));
            }
            if (OUT_BF16) {
                uint16_t* cp = Cb + (int64_t)m * ldc + n;
                if (full && ((ldc & 3) == 0)) {
                    uint2 o;
                    o.x = f2bf2(v[0], v[1]);
                    o.y = f2bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(cp) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
                }
            } else {
                float* cp = Cf + (int64_t)m * ldc + n;
                if (full && ((ldc & 3) == 0)) {                 // (batching these reads ahead of the stores measured slower)
                    float4 o = ACCUM ? *reinterpret_cast<float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                    *reinterpret_cast<float4*>(cp) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = (ACCUM ? cp[r] : 0.f) + v[r];
                }
            }
        }
    }
}


extern float* g_tail_ws;          // st_gemm_set_workspace: fp32 slices of split tail tiles (one stream at a time)
extern int64_t g_tail_ws_bytes;
extern int g_decode_nt;            // decode tiles stream their weights non-temporally when only one row tile reads them (ST_DECODE_NT)

template <int BM, int BN, int WM, int WN, int STAGES, bool HB, bool HR, bool OB, bool AC, bool MB = false, bool LE = false, bool PP_ = false, bool NT_ = false>
static int launch_tile(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res,
                       int64_t ldr, uint16_t* Cb, float* Cf, int64_t ldc, int M, int N, int K, hipStream_t s, int splits = 1,
                       int64_t slab_stride = 0) {
    constexpr int smem = LE ? (STAGES * (BM + BN) * 128 > 256 * 528 ? STAGES * (BM + BN) * 128 : 256 * 528) : STAGES * (BM + BN) * 128;
    auto kern = gemm_tile_kernel<BM, BN, WM, WN, STAGES, HB, HR, OB, AC, false, MB, false, LE, false, false, PP_, NT_>;
    static bool configured = false;
    if (!configured) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    const int tiles_m = st_cdiv(M, BM), tiles_n = st_cdiv(N, BN);
    const int kt_per_split = st_cdiv(K / 64, splits);
    const int nb = tiles_m * tiles_n;
    TailArgs tail{nullptr, nb, 1, 0, 1};
    int tail_tiles = 0;
    if (BM * BN >= 256 * 256 && splits == 1 && g_tail_ws) {  // one workgroup per CU: whole rounds of CUs tiles, then the tail
        const int ncu = st_num_cus(), nkt = K / 64, r = nb % ncu;
        const int64_t cap = g_tail_ws_bytes / ((int64_t)BM * BN * 4);
        // cost in K-tile steps (~1.5 us each at 256x256x64), fitted to kernel traces at 7B shapes: a whole tile pays ~4 steps of
        // prologue + epilogue, a piece ~14 (cold start, 256 KiB fp32 slice written), the finish launch ~20
        int best = 1, best_cost = nkt + 4;
        for (int S = 2; S <= 8 && r > 0; ++S) {
            if ((int64_t)r * S > cap || S * 4 > nkt) break;
            const int cost = st_cdiv(r * S, ncu) * (st_cdiv(nkt, S) + 14) + 20;
            if (cost < best_cost) { best = S; best_cost = cost; }
        }
        if (best > 1) { tail = TailArgs{g_tail_ws, nb - r, best, 0, 1}; tail_tiles = r; }
    }
    hipLaunchKernelGGL(kern, dim3(tail.full_blocks + tail_tiles * tail.split, st_cdiv(K / 64, kt_per_split)), dim3(64 * WM * WN), smem, s, A,
                       lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K, tiles_m, tiles_n, kt_per_split, slab_stride, tail);
    if (tail_tiles) {
        tail.mode = 2;
        constexpr int TN_ = BN / WN / 16;
        while (tail.fin_sub * 2 <= TN_ && tail_tiles * tail.fin_sub * 2 <= 4 * st_num_cus()) tail.fin_sub *= 2;
        hipLaunchKernelGGL(kern, dim3(tail_tiles, tail.fin_sub), dim3(64 * WM * WN), 0 /* no K loop: no LDS */, s, A, lda, B, ldb, bias, res, ldr, Cb, Cf, ldc, M, N, K,
                           tiles_m, tiles_n, kt_per_split, slab_stride, tail);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// dX / dW forms on the production tile (256x256, 8 waves, mid-tile barrier, LDS-staged epilogue)
template <bool AS_, bool BS_, bool OB, bool AC>
static int launch_tile_layout(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, float* Cf, int64_t ldc, int M, int N,
                              int K, hipStream_t s) {
    constexpr int smem = 256 * 528 > 2 * (256 + 256) * 128 ? 256 * 528 : 2 * (256 + 256) * 128;
    auto kern = gemm_tile_kernel<256, 256, 4, 2, 2, false, false, OB, AC, false, true, false, true, AS_, BS_>;
    static bool configured = false;
    if (!configured) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    const int tiles_m = st_cdiv(M, 256), tiles_n = st_cdiv(N, 256), nb = tiles_m * tiles_n;
    TailArgs tail{nullptr, nb, 1, 0, 1};
    int tail_tiles = 0;
    if (g_tail_ws) {                                          // same tail split as launch_tile
        const int ncu = st_num_cus(), nkt = K / 64, r = nb % ncu;
        const int64_t cap = g_tail_ws_bytes / ((int64_t)256 * 256 * 4);
        int best = 1, best_cost = nkt + 4;
        for (int S = 2; S <= 8 && r > 0; ++S) {
            if ((int64_t)r * S > cap || S * 4 > nkt) break;
            const int cost = st_cdiv(r * S, ncu) * (st_cdiv(nkt, S) + 14) + 20;
            if (cost < best_cost) { best = S; best_cost = cost; }
        }
        if (best > 1) { tail = TailArgs{g_tail_ws, nb - r, best, 0, 1}; tail_tiles = r; }
    }
    hipLaunchKernelGGL(kern, dim3(tail.full_blocks + tail_tiles * tail.split, 1), dim3(512), smem, s, A, lda, B, ldb, (const uint16_t*)nullptr,
                       (const uint16_t*)nullptr, (int64_t)0, Cb, Cf, ldc, M, N, K, tiles_m, tiles_n, K / 64, (int64_t)0, tail);
    if (tail_tiles) {
        tail.mode = 2;
        while (tail.fin_sub * 2 <= 8 && tail_tiles * tail.fin_sub * 2 <= 4 * st_num_cus()) tail.fin_sub *= 2;
        hipLaunchKernelGGL(kern, dim3(tail_tiles, tail.fin_sub), dim3(512), 0, s, A, lda, B, ldb, (const uint16_t*)nullptr,
                           (const uint16_t*)nullptr, (int64_t)0, Cb, Cf, ldc, M, N, K, tiles_m, tiles_n, K / 64, (int64_t)0, tail);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int BM, int BN, int WM, int WN, int STAGES, bool MB = false, bool S8 = false, bool NT_ = false>
static int launch_tile_swiglu(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, uint16_t* Cb, int64_t ldc, int M, int N,
                              int K, hipStream_t s, uint16_t* gu = nullptr, int64_t ldgu = 0) {
    constexpr int smem = STAGES * (BM + BN) * 128;
    auto kern = gemm_tile_kernel<BM, BN, WM, WN, STAGES, false, false, true, false, true, MB, S8, false, false, false, false, NT_>;
    static bool configured = false;
    if (!configured) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    const int tiles_m = st_cdiv(M, BM), tiles_n = st_cdiv(N, BN / 2);
    // SWIGLU mode: the fp32-output pointer slot carries the optional bf16 gate|up buffer, the residual stride slot its row stride
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1), dim3(64 * WM * WN), smem, s, A, lda, B, ldb, (const uint16_t*)nullptr,
                       (const uint16_t*)nullptr, ldgu, Cb, reinterpret_cast<float*>(gu), ldc, M, N, K, tiles_m, tiles_n, K / 64, (int64_t)0,
                       TailArgs{nullptr, tiles_m * tiles_n, 1, 0, 1});
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

