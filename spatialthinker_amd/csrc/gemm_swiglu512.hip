// gemm_swiglu512.hip — the decode gate/up projection + SwiGLU for 257..512 rows with ONE pass over the weights (round 5).
//
//   out[M, I] = silu(A Wg^T) * (A Wu^T),  A [M <= 512, K] bf16,  W = [Wg ; Wu] (2I x K) bf16,  K % 64 == 0
//
// Why: at 257..512 rows the 256-row decode tiles run every weight tile twice (two row tiles = two rounds of 237 workgroups at 71 us per
// round whatever the rows: 143 us per layer, the largest kernel of the 512-row decode phase, profiles/r05_notes.md §0).  Both row tiles
// inside one workgroup need 84 KB of LDS per 64-wide K-tile (A 512 x 128 B + B 160 x 128 B): two ring slots do not fit 160 KB.  Here the
// ring moves in K-steps of 32 (64-byte LDS rows): three slots of 42 KB, two in flight.
//
// Workgroup = 8 waves = all 512 rows x 80 output columns (80 gate + 80 up weight rows); 237 workgroups for the 7B MLP: one round.
// Wave w owns rows [64 w, 64 w + 64) x all 160 weight rows: TM = 4 x TN = 10 MFMA 16x16x32 tiles, 160 accumulator registers; a wave
// whose rows are all >= M skips its MFMAs and fragment reads (the decode phases shrink in steps of 64 rows, not 256).
// LDS image: row-major [rows][64 B], written by LDS-DMA (1 KiB pieces = 16 rows x 64 B, lane-linear), 16-byte chunk c of row r holds
// global chunk c ^ f(r), f(r) = (0, 2, 3, 1)[(r >> 2) & 3]: the four 16-lane service groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...)
// then touch 16 distinct 16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md §LDS) — derivation in profiles/r05_notes.md.
// Weight rows: 16 gate rows, then the 16 matching up rows, alternating (accumulator tiles 2t / 2t + 1 hold gate / up of the same 16
// output columns in the same lanes): the SwiGLU is elementwise in registers, with the roundings of the unfused chain (bf16 gate, bf16
// up, bf16 act(gate)) — bit-identical to the 256-row tiles' epilogue, and the K summation runs in the same order.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int S5_ROWS = 512, S5_COLS = 80, S5_BROWS = 160;
constexpr int S5_A_BYTES = S5_ROWS * 64, S5_B_BYTES = S5_BROWS * 64, S5_SLOT = S5_A_BYTES + S5_B_BYTES;      // 32 KB + 10 KB
constexpr int S5_SLOTS = 3;
constexpr int S5_A_PIECES = S5_ROWS / 16, S5_B_PIECES = S5_BROWS / 16;                                    // 32, 10 one-KiB pieces per K-step

__device__ __forceinline__ int s5_swz(int row) {            // f(r) = (0, 2, 3, 1)[(r >> 2) & 3]
    const int t = (row >> 2) & 3;
    return (((t ^ (t >> 1)) & 1) << 1) | (t >> 1);
}

__device__ __forceinline__ void s5_glds16(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void s5_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

}  // namespace

// ST_GU512_TRACE builds (tools/gu512_phase_trace.py, never the shipped library): every wave sums the shader-clock cycles between fixed
// points of the ping-pong loop and leaves the sums in st_gu512_trace_ptr[workgroup][wave][8].
#ifdef ST_GU512_TRACE
__device__ unsigned long long* st_gu512_trace_ptr = nullptr;
extern "C" int st_gu512_trace_set(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(st_gu512_trace_ptr), &buf, sizeof(buf)); }
#define G5_DECL unsigned long long tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = 0
#define G5_START() do { __builtin_amdgcn_sched_barrier(0); tr_prev = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G5_POINT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); tr_acc[i] += t_ - tr_prev; tr_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define G5_FLUSH() do { if (st_gu512_trace_ptr && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) st_gu512_trace_ptr[(blockIdx.x * 8 + wave) * 8 + i_] = tr_acc[i_]; } } while (0)
#else
#define G5_DECL
#define G5_START()
#define G5_POINT(i)
#define G5_FLUSH()
#endif

template <bool PP>
__global__ __launch_bounds__(512) void gemm_swiglu512_kernel(const uint16_t* __restrict__ A, int64_t lda, const uint16_t* __restrict__ W,
                                                            int64_t ldw, uint16_t* __restrict__ out, int64_t ldo, int M, int I, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-contiguous tile order: neighbouring column tiles share nothing but the activations, which every XCD holds anyway; the map
    // keeps an XCD on a contiguous range of weight rows (its HBM pages)
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = bid * S5_COLS;                            // first output column of this tile
    const int nks = K / 32;

    // ---- LDS-DMA sources.  A: pieces wave*4 .. +3 (rows 16 p .. 16 p + 15); B: pieces 0..9 dealt round-robin (waves 0, 1 take two)
    const int prow = lane >> 2, pchunk = lane & 3;           // row / 16-byte chunk of this lane inside a 1-KiB piece
    const uint16_t* asrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 16 + prow;
        const int gr = r < M ? r : M - 1;
        asrc[j] = A + (int64_t)gr * lda + ((pchunk ^ s5_swz(r)) << 3);
    }
    const int nbp = wave < S5_B_PIECES - 8 ? 2 : 1;          // B pieces of this wave: w and w + 8
    const uint16_t* bsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + 8 * j) * 16 + prow;            // B-tile row: pair t = r / 32, 16 gate rows then 16 up rows
        const int t = r >> 5, within = r & 31;
        int col = n0 + t * 16 + (within & 15);
        col = col < I ? col : I - 1;
        const int64_t wrow = (within & 16) ? (int64_t)I + col : col;
        bsrc[j] = W + wrow * ldw + ((pchunk ^ s5_swz(r)) << 3);
    }
    auto stage = [&](int ks, int slot) {
        char* base = smem + slot * S5_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) s5_glds16(asrc[j] + (int64_t)ks * 32, base + (wave * 4 + j) * 1024);
        s5_glds16(bsrc[0] + (int64_t)ks * 32, base + S5_A_BYTES + wave * 1024);
        if (nbp == 2) s5_glds16(bsrc[1] + (int64_t)ks * 32, base + S5_A_BYTES + (wave + 8) * 1024);
    };

    f32x4 acc[10][4];
#pragma unroll
    for (int ni = 0; ni < 10; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool live = wave * 64 < M;                         // this wave's rows hold at least one valid row
    const int frow = lane & 15, fk = lane >> 4;
    // fragment byte offsets inside a slot: A rows wave*64 + mi*16 + frow, B rows ni*16 + frow; the swizzle depends on (row >> 2) & 3 only,
    // i.e. on frow (tile bases are multiples of 16)
    const int chunk = (fk ^ s5_swz(frow)) << 4;
    const int a_off = (wave * 64 + frow) * 64 + chunk;
    const int b_off = S5_A_BYTES + frow * 64 + chunk;

    if constexpr (PP) {
        // ---- ping-pong schedule.  Waves w and w + 4 share a SIMD (tools/probes/wave_simd_map.hip); in lockstep both read their fragments
        // at the same time and then both queue for the matrix pipe (first version: 142 us = the two-round tile's time, MFMA pipe ~50 % busy).
        // Here the two groups run HALF a K-step apart, two barriers per K-step:
        //     phase A(ks): group 0 MFMAs(ks)  | group 1 reads fragments(ks)   | every wave waits for ITS copies of K-step ks + 1
        //     phase B(ks): group 0 reads (ks + 1) | group 1 MFMAs(ks)         | slot ks is free (both groups have read it): DMA of ks + 3
        // so one wave of a SIMD feeds the matrix pipe while its partner does LDS reads / DMA issues / waits.
        // Written as ONE program for every wave (no per-group code paths: the first attempt with `if (group) mfmas() else reads()` spilled
        // 508 registers): loop body = MFMAs(ks) | barrier | DMA(ks + 3) + reads(ks + 1) | barrier, and group 1 simply ENTERS the loop one
        // barrier later, so its MFMA phase coincides with group 0's read phase.  K-step ks + 1 must have landed in front of the barrier
        // that ends global phase 2 ks: group 0 stands at the end of its MFMAs(ks) there, group 1 at the end of its reads(ks).
        const int grp = wave >> 2;
        bf16x8 af[4], bfr[10];
        auto wait_copies = [&](int x) {                      // this wave's copies of K-step x have landed (x + 1 may stay in flight)
            if (x >= nks) return;
            if (x + 1 < nks) { if (nbp == 2) s5_wait_vmcnt<6>(); else s5_wait_vmcnt<5>(); }
            else s5_wait_vmcnt<0>();
        };
        for (int j = 0; j < 3; ++j) if (j < nks) stage(j, j);
        if (nks > 2) { if (nbp == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        else wait_copies(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (live) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(smem + a_off + mi * 1024);
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(smem + b_off + ni * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {                                      // group 1 idles through global phase 0
            wait_copies(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        int slot = 0;
        G5_DECL;
        G5_START();
        for (int ks = 0; ks < nks; ++ks) {
            const int s1 = slot + 1 == S5_SLOTS ? 0 : slot + 1;
            if (live) {
                __builtin_amdgcn_s_setprio(2);
#ifdef ST_GU512_NOMFMA                                       /* timing experiment: one MFMA per K-step instead of 40 */
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0], af[0], acc[0][0], 0, 0, 0);
#pragma unroll
                for (int ni = 1; ni < 10; ++ni) asm volatile("" :: "v"(bfr[ni]));
#pragma unroll
                for (int mi = 1; mi < 4; ++mi) asm volatile("" :: "v"(af[mi]));
#else
#pragma unroll
                for (int ni = 0; ni < 10; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
#endif
                __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            G5_POINT(0);
            if (grp == 0) wait_copies(ks + 1);
            G5_POINT(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            G5_POINT(2);
            if (ks + 3 < nks) stage(ks + 3, slot);            // slot of K-step ks: both groups have read it by now
            G5_POINT(3);
            if (live && ks + 1 < nks) {
                const char* base = smem + s1 * S5_SLOT;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(base + a_off + mi * 1024);
#pragma unroll
                for (int ni = 0; ni < 10; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(base + b_off + ni * 1024);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            G5_POINT(4);
            if (grp == 1) wait_copies(ks + 2);
            G5_POINT(5);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            G5_POINT(6);
            slot = s1;
        }
        if (grp == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        G5_FLUSH();
    } else {
    if (nks > 0) stage(0, 0);
    if (nks > 1) stage(1, 1);
    int slot = 0;
    for (int ks = 0; ks < nks; ++ks) {
        // K-step ks has landed when at most the copies of K-step ks + 1 are still in flight (this wave's own; the barrier publishes the others')
        if (ks + 1 < nks) { if (nbp == 2) s5_wait_vmcnt<6>(); else s5_wait_vmcnt<5>(); }
        else s5_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // every wave has finished its reads of K-step ks - 1 (they precede its MFMAs, which precede this barrier): refill that slot
        if (ks + 2 < nks) { int ns = slot + 2; ns = ns >= S5_SLOTS ? ns - S5_SLOTS : ns; stage(ks + 2, ns); }
        if (live) {
            const char* base = smem + slot * S5_SLOT;
            bf16x8 af[4], bfr[10];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = *reinterpret_cast<const bf16x8*>(base + a_off + mi * 1024);
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) bfr[ni] = *reinterpret_cast<const bf16x8*>(base + b_off + ni * 1024);
#pragma unroll
            for (int ni = 0; ni < 10; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
        slot = slot + 1 == S5_SLOTS ? 0 : slot + 1;
    }
    }
    if (!live) return;
    // ---- SwiGLU epilogue: tiles 2t (gate) and 2t + 1 (up) of output columns n0 + 16 t + 4 (lane >> 4) .. + 3, row = lane & 15
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = wave * 64 + mi * 16 + frow;
        if (m >= M) continue;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int n = n0 + t * 16 + fk * 4;
            if (n >= I) continue;
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                   // same roundings as the unfused chain: bf16 gate / up, bf16 act(gate)
                const float g = bfround(acc[2 * t][mi][r]), u = bfround(acc[2 * t + 1][mi][r]);
                o[r] = f2bf(bfround(g * sigmoidf_(g)) * u);
            }
            uint16_t* cp = out + (int64_t)m * ldo + n;
            if (n + 3 < I && ((ldo & 3) == 0)) {
                uint2 w;
                w.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                w.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                *reinterpret_cast<uint2*>(cp) = w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < I) cp[r] = o[r];
            }
        }
    }
}

// out[M, I] = silu(A gate_w^T) * (A up_w^T) for 1 <= M <= 512 in one pass over the weights (8-wave 512 x 80-column tiles, K-steps of 32)
int st_gemm_swiglu512_launch(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, uint16_t* out, int64_t ldo, int M, int I, int K,
                             hipStream_t s) {
    constexpr int smem = S5_SLOTS * S5_SLOT;                 // 129,024 bytes
    // ST_GU512_MODE=1: the lockstep first version (A/B).  Also measured and removed (commit 317cd87, profiles/r05_notes.md
    // §1b): nt weight stream (157 vs 134 us: with 64-byte row pieces the second half of every 128-byte line comes from HBM again), the loop
    // without its data waits (same time: it does not wait for data), two of the six copies per wave moved into the MFMA phase (same time
    // at 512 rows, slower below), every workgroup walking K from its own starting point (143 vs 136 us: the 237 workgroups reading the SAME
    // activation lines at the same time is an L2 benefit, not a conflict): the tile runs at the rate the CU can fill its LDS (4.8 MB per
    // workgroup at ~36 GB/s).
    // Round 6, measured and removed (profiles/r06_notes.md §1b, §1c; all bit-identical): activations loaded straight into registers (156.6 vs
    // 132.8 us), activations staged as 8-row x 128-byte pieces into per-wave rings of K-step pairs (137.3 vs 133.9: 38 % fewer L2 requests,
    // TA busy -19 %, no conflicts — and no faster), weights staged as 128-byte pieces five K-steps ahead (129.5 vs 126.6).  With 39 of its 40
    // MFMAs per K-step removed the launch still takes 118 us (at 2.18 GHz instead of 1.65): see the phase trace, tools/gu512_phase_trace.py.
    static const int mode = [] { const char* e = getenv("ST_GU512_MODE"); return e ? atoi(e) : 0; }();
    static bool configured = false;
    if (!configured) {
        hipFuncSetAttribute((const void*)gemm_swiglu512_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipFuncSetAttribute((const void*)gemm_swiglu512_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        configured = true;
    }
    const dim3 grid(st_cdiv(I, S5_COLS)), block(512);
    if (mode == 1) hipLaunchKernelGGL((gemm_swiglu512_kernel<false>), grid, block, smem, s, A, lda, W, ldw, out, ldo, M, I, K);
    else hipLaunchKernelGGL((gemm_swiglu512_kernel<true>), grid, block, smem, s, A, lda, W, ldw, out, ldo, M, I, K);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
