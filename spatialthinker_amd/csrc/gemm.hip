// gemm.hip — bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  (+bias, +residual), fp32 accumulate.
//
// Both operands are K-contiguous (y = x W^T with W stored (out,in) as HF does), so every MFMA fragment is
// one 16-byte LDS read.  Structure (cdna_hip_programming.md §5, "step-3" + the 2-phase T3/T4 recipe):
//   * 128x128 output tile, BK = 64, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 MFMA 16x16x32 tiles;
//   * operands staged HBM -> LDS by `global_load_lds` 16 B/lane (no VGPR round trip), two LDS stages
//     (2 x 32 KiB) so the loads of K-tile t+1 fly while tile t is multiplied; one barrier per K-tile;
//   * LDS image is lane-linear (what the LDS-DMA writes); bank conflicts are removed by XOR-swizzling the
//     16-byte chunk index with (row>>1)&7 on the SOURCE address and on the fragment read (rule 21);
//   * the MFMA is issued with swapped operands (D = Bfrag x Afrag^T) so each lane ends up with 4 consecutive
//     output columns of one row -> 8-byte bf16 / 16-byte fp32 stores;
//   * workgroup -> tile map is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous range of tiles,
//     walked in groups of 8 tile-rows so the A and B panels of neighbours stay L2-resident.
// Roofline: MFMA-bound, 2*M*N*K flop per launch.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (BM * BK * 2 + BN * BK * 2)   // 32 KiB

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// stage one 128x64 operand tile: 16 KiB = 16 wave-instructions of 1 KiB; wave w issues 4 of them.
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ X, int64_t ldx, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int inst = wave * 4 + j;
        const int p = inst * 64 + lane;          // 16-byte chunk index inside the tile
        const int r = p >> 3, cpos = p & 7;
        const int kc = cpos ^ ((r >> 1) & 7);    // which global chunk lands at this LDS position
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;        // clamp: tail rows compute garbage that is never stored
        glds16(X + (int64_t)gr * ldx + k0 + kc * 8, lds_tile + inst * 1024);
    }
}

template <bool HAS_BIAS, bool HAS_RES, bool OUT_BF16, bool OUT_F32, bool ACCUM>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                        const uint16_t* __restrict__ B, int64_t ldb,
                                                        const uint16_t* __restrict__ bias,
                                                        const uint16_t* __restrict__ res, int64_t ldr,
                                                        uint16_t* __restrict__ Cb, float* __restrict__ Cf, int64_t ldc,
                                                        int M, int N, int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap + grouped tile order
    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GM = 8;
    const int per_group = GM * tiles_n;
    const int group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage_tile(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile(B, ldb, n0, N, 0, smem + BM * BK * 2, wave, lane);

    // per-lane fragment addressing (constant over the K loop)
    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * STAGE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // tile kt landed; everyone is done with the other stage
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            stage_tile(A, lda, m0, M, (kt + 1) * BK, nxt, wave, lane);
            stage_tile(B, ldb, n0, N, (kt + 1) * BK, nxt + BM * BK * 2, wave, lane);
        }
        const char* la = cur;
        const char* lb = cur + BM * BK * 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + frow;
                const int rb = wn * 64 + i * 16 + frow;
                const int kc = s * 4 + fk;
                af[i] = *reinterpret_cast<const bf16x8*>(la + ra * 128 + ((kc ^ ((ra >> 1) & 7)) << 4));
                bfr[i] = *reinterpret_cast<const bf16x8*>(lb + rb * 128 + ((kc ^ ((rb >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);
        }
    }

    // epilogue: lane holds D[n = (lane>>4)*4 + r][m = lane&15] for each (ni, mi)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            const bool full = (n + 3 < N);
            if (HAS_BIAS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (full || n + r < N) v[r] += bf2f(bias[n + r]);
            }
            if (HAS_RES) {
                const uint16_t* rp = res + (int64_t)m * ldr + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (full || n + r < N) v[r] += bf2f(rp[r]);
            }
            if (OUT_F32) {
                float* cp = Cf + (int64_t)m * ldc + n;
                if (full && ((ldc & 3) == 0)) {
                    float4 o = ACCUM ? *reinterpret_cast<float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                    *reinterpret_cast<float4*>(cp) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = (ACCUM ? cp[r] : 0.f) + v[r];
                }
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
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Skinny variant for the rollout decode step (M <= 256 rows: one token per live sequence).  There the GEMM is a
// pure WEIGHT STREAM (HBM-bound: 2*N*K bytes per launch, ~15 GB per decode step of the 7B LM), so the mapping
// maximises the number of workgroups pulling weights from HBM and keeps the activation traffic on chip:
//   * workgroup = 4 waves = 64 output columns x ALL rows x one K-slice (grid = N/64 x SPLITK, SPLITK chosen so the
//     grid is >= ~2 workgroups per CU);
//   * weights go HBM -> VGPR directly (each wave owns 16 columns; a W line is used by exactly one wave: no LDS);
//   * the activation K-chunk (M x 64) is staged ONCE per workgroup into LDS by LDS-DMA (double-buffered, same XOR
//     swizzle as the big kernel) and shared by the 4 waves, so x traffic is M/64 of the W traffic instead of M/16;
//   * SPLITK > 1: fp32 partial slabs [split][M][N] go to a caller-provided scratch with plain 16-byte stores and a finish
//     kernel applies bias/residual, rounds to bf16 and re-zeroes the scratch (self-cleaning).
// ------------------------------------------------------------------------------------------------------
template <int MT, bool DIRECT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const uint16_t* __restrict__ A, int64_t lda,
                                                         const uint16_t* __restrict__ B, int64_t ldb,
                                                         const uint16_t* __restrict__ bias, const uint16_t* __restrict__ res,
                                                         int64_t ldr, uint16_t* __restrict__ C, int64_t ldc,
                                                         float* __restrict__ scratch, int M, int N, int K, int k_per_split) {
    constexpr int ROWS = MT * 16;
    constexpr int XBYTES = ROWS * 128;                    // one 64-wide K-chunk of x
    __shared__ __attribute__((aligned(16))) char smem[2 * XBYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 64 + wave * 16;
    const int k_begin = blockIdx.y * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);
    const int nchunks = (k_end - k_begin) / 64;
    const int fr = lane & 15, fk = lane >> 4;
    int bn = n0 + fr; bn = bn < N ? bn : N - 1;
    const uint16_t* bp = B + (int64_t)bn * ldb + k_begin + fk * 8;

    auto stage_x = [&](int chunk, char* dst) {
        // ROWS*8 16-byte pieces, 64 per wave-instruction; wave w issues instructions w, w+4, ...
#pragma unroll
        for (int inst = wave; inst < ROWS / 8; inst += 4) {
            const int p = inst * 64 + lane;
            const int r = p >> 3, cpos = p & 7;
            const int kc = cpos ^ ((r >> 1) & 7);
            const int gr = r < M ? r : M - 1;
            glds16(A + (int64_t)gr * lda + k_begin + chunk * 64 + kc * 8, dst + inst * 1024);
        }
    };
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[2], wn[2];
    if (nchunks > 0) {
        stage_x(0, smem);
        wf[0] = *reinterpret_cast<const bf16x8*>(bp);
        wf[1] = *reinterpret_cast<const bf16x8*>(bp + 32);
    }
    for (int c = 0; c < nchunks; ++c) {
        char* cur = smem + (c & 1) * XBYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nchunks) {
            stage_x(c + 1, smem + ((c + 1) & 1) * XBYTES);
            wn[0] = *reinterpret_cast<const bf16x8*>(bp + (c + 1) * 64);
            wn[1] = *reinterpret_cast<const bf16x8*>(bp + (c + 1) * 64 + 32);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 16 + fr;
                const int kc = s * 4 + fk;
                const bf16x8 xa = *reinterpret_cast<const bf16x8*>(cur + r * 128 + ((kc ^ ((r >> 1) & 7)) << 4));
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], xa, acc[mt], 0, 0, 0);      // D[n][m]
            }
        }
        wf[0] = wn[0]; wf[1] = wn[1];
    }
    const int n = n0 + fk * 4;
    if (n >= N) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + fr;
        if (m >= M) continue;
        float v[4] = {acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]};
        if (DIRECT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < N) {
                    if (HAS_BIAS) v[r] += bf2f(bias[n + r]);
                    if (HAS_RES) v[r] += bf2f(res[(int64_t)m * ldr + n + r]);
                }
            }
            uint16_t* cp = C + (int64_t)m * ldc + n;
            if (n + 3 < N && (ldc & 3) == 0) {
                uint2 o;
                o.x = f2bf2(v[0], v[1]);
                o.y = f2bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(cp) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) cp[r] = f2bf(v[r]);
            }
        } else {
            float* sp = scratch + ((int64_t)blockIdx.y * M + m) * N + n;          // partial slab [split][M][N]
            if (n + 3 < N && (N & 3) == 0) *reinterpret_cast<float4*>(sp) = make_float4(v[0], v[1], v[2], v[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) sp[r] = v[r];
            }
        }
    }
}

template <bool HAS_BIAS, bool HAS_RES>
__global__ void gemm_skinny_finish(const float* __restrict__ scratch, int splits, const uint16_t* __restrict__ bias,
                                   const uint16_t* __restrict__ res, int64_t ldr, uint16_t* __restrict__ C, int64_t ldc, int M, int N) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += scratch[(int64_t)s * M * N + idx];      // fixed order: deterministic
    if (HAS_BIAS) v += bf2f(bias[n]);
    if (HAS_RES) v += bf2f(res[(int64_t)m * ldr + n]);
    C[(int64_t)m * ldc + n] = f2bf(v);
}

template <int MT>
static int launch_skinny(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias, const uint16_t* res,
                         int64_t ldr, uint16_t* C, int64_t ldc, float* scratch, int64_t scratch_elems, int M, int N, int K, hipStream_t s) {
    const int nblk = st_cdiv(N, 64);
    int sk = 1;
    if (scratch && scratch_elems >= (int64_t)2 * M * N) {
        sk = st_cdiv(512, nblk);
        if ((int64_t)sk * M * N > scratch_elems) sk = (int)(scratch_elems / ((int64_t)M * N));
        const int max_sk = K / 128;                          // at least two 64-chunks per slice
        if (sk > max_sk) sk = max_sk;
        if (sk > 16) sk = 16;
        if (sk < 1) sk = 1;
    }
    int kps = st_cdiv(st_cdiv(K, sk), 64) * 64;
    sk = st_cdiv(K, kps);
    const dim3 grid(nblk, sk), block(256);
    const bool hb = bias != nullptr, hr = res != nullptr;
    if (sk == 1) {
#define SKD(HB, HR) hipLaunchKernelGGL((gemm_skinny_kernel<MT, true, HB, HR>), grid, block, 0, s, A, lda, B, ldb, bias, res, ldr, C, ldc, scratch, M, N, K, kps)
        if (hb && hr) SKD(true, true); else if (hb) SKD(true, false); else if (hr) SKD(false, true); else SKD(false, false);
#undef SKD
    } else {
        hipLaunchKernelGGL((gemm_skinny_kernel<MT, false, false, false>), grid, block, 0, s, A, lda, B, ldb, bias, res, ldr, C, ldc, scratch, M, N, K, kps);
        const dim3 fg(st_cdiv((int64_t)M * N, 256));
#define SKF(HB, HR) hipLaunchKernelGGL((gemm_skinny_finish<HB, HR>), fg, dim3(256), 0, s, scratch, sk, bias, res, ldr, C, ldc, M, N)
        if (hb && hr) SKF(true, true); else if (hb) SKF(true, false); else if (hr) SKF(false, true); else SKF(false, false);
#undef SKF
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <bool HB, bool HR, bool OB, bool OF, bool AC>
static int launch_gemm(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                       const uint16_t* res, int64_t ldr, uint16_t* Cb, float* Cf, int64_t ldc, int M, int N, int K,
                       hipStream_t s) {
    const int tiles_m = st_cdiv(M, BM), tiles_n = st_cdiv(N, BN);
    hipLaunchKernelGGL((gemm_nt_kernel<HB, HR, OB, OF, AC>), dim3(tiles_m * tiles_n), dim3(256), 0, s, A, lda, B, ldb, bias, res,
                       ldr, Cb, Cf, ldc, M, N, K, tiles_m, tiles_n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int st_gemm_tile_dispatch(int variant, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                          const uint16_t* res, int64_t ldr, uint16_t* Cb, float* Cf, int64_t ldc, int accumulate, int M, int N, int K,
                          hipStream_t s);

// production tile of the training-shape GEMMs: 40 = 4 waves x 128x128 with the hand-scheduled K loop (gemm_asm4.hip; round 3: in the
// bench 1248 vs 1185 TF/s for the GEMM class, 16.1 vs 15.7 samples/s), 23 = 8 waves x 64x128 (mid-tile barrier, LDS-staged epilogue).
// ST_GEMM_VARIANT / st_gemm_select override it (A/B runs).
static bool train_variant_ok(int v) { return v == 23 || v == 40 || v == 6 || v == 8 || v == 31; }
// the environment override goes through the same whitelist as st_gemm_select: anything else (in particular the debug variants 41-47,
// timing experiments with WRONG results, reachable only through st_gemm_nt_variant) falls back to the production tile with a notice
int g_train_variant = [] {
    const char* e = getenv("ST_GEMM_VARIANT");
    if (!e) return 40;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || !train_variant_ok((int)v)) {
        fprintf(stderr, "[st_hip] ST_GEMM_VARIANT=%s is not a production tile (23, 40, 6, 8, 31): using 40\n", e);
        return 40;
    }
    return (int)v;
}();
extern "C" int st_gemm_select(int variant) {
    if (!train_variant_ok(variant)) return ST_EINVAL;
    g_train_variant = variant;
    return 0;
}

extern "C" int st_gemm_nt(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
                          const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, float* out_f32, int64_t ldc,
                          int accumulate, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || M < 0 || N < 0 || K <= 0 || (K % BK) || (lda & 7) || (ldb & 7) || lda < K || ldb < K || ldc < N) return ST_EINVAL;
    if ((out_bf16 == nullptr) == (out_f32 == nullptr)) return ST_EINVAL;      // exactly one output
    if (out_bf16 && accumulate) return ST_EINVAL;
    if ((((uintptr_t)A) & 15) || (((uintptr_t)B) & 15)) return ST_EINVAL;
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool hb = bias != nullptr, hr = residual != nullptr;
    StProfScope ps(ST_K_GEMM, s, 2.0 * (double)M * (double)N * (double)K,     // MFMA-bound class only (roofline.achieved)
                   st_prof_tag(0, (hb ? 1 : 0) | (hr ? 2 : 0) | (out_f32 ? 4 : 0) | (accumulate ? 8 : 0), M, N, K));
    // tile choice (tools/gemm_shapes.py on MI355X, all 12 fwd/dX/dW shapes of a 7B layer): the 256x256 tile (8 waves, 128 KiB LDS,
    // 128 flop/B of L2 traffic) wins or ties from ~0.5 workgroups per CU upwards (dW of the 3584x3584 projection, 196 tiles:
    // 1098 vs 926 TF); only smaller problems fill the chip better with 128x128 tiles at 2 workgroups/CU.
    if ((int64_t)st_cdiv(M, 256) * st_cdiv(N, 256) >= 128) {
        // the 4-wave tile addresses an operand tile through 32-bit buffer offsets: row pitches of 2^22 elements and more take the 8-wave tile
        const int v = (g_train_variant == 40 && (lda >= (1 << 22) || ldb >= (1 << 22))) ? 23 : g_train_variant;
        return st_gemm_tile_dispatch(v, A, lda, B, ldb, bias, residual, ldr, out_bf16, out_f32, ldc, accumulate, M, N, K, s);
    }
#define GO(HB, HR, OB, OF, AC) return launch_gemm<HB, HR, OB, OF, AC>(A, lda, B, ldb, bias, residual, ldr, out_bf16, out_f32, ldc, M, N, K, s)
    if (out_bf16) {
        if (hb && hr) GO(true, true, true, false, false);
        if (hb) GO(true, false, true, false, false);
        if (hr) GO(false, true, true, false, false);
        GO(false, false, true, false, false);
    } else {
        if (hb || hr) return ST_EINVAL;
        if (accumulate) GO(false, false, false, true, true);
        GO(false, false, false, true, false);
    }
#undef GO
}

int st_gemm_tile_decode(int variant, int splits, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* bias,
                        const uint16_t* res, int64_t ldr, uint16_t* Cb, float* slabs, int M, int N, int K, int64_t ldc, hipStream_t s);

static int launch_decode_tiles(int variant, int splits, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                               const uint16_t* bias, const uint16_t* res, int64_t ldr, uint16_t* C, int64_t ldc, float* scratch,
                               int64_t scratch_elems, int M, int N, int K, hipStream_t s) {
    if (splits > K / 64) splits = K / 64;
    if (splits > 1 && (!scratch || (int64_t)splits * M * N > scratch_elems)) return ST_EINVAL;
    if (splits < 1) splits = 1;
    const int kt_per = st_cdiv(K / 64, splits);
    splits = st_cdiv(K / 64, kt_per);                       // slices actually launched
    int rc = st_gemm_tile_decode(variant, splits, A, lda, B, ldb, bias, res, ldr, C, scratch, M, N, K, ldc, s);
    if (rc || splits == 1) return rc;
    const bool hb = bias != nullptr, hr = res != nullptr;
    const dim3 fg(st_cdiv((int64_t)M * N, 256));
#define SKF(HB, HR) hipLaunchKernelGGL((gemm_skinny_finish<HB, HR>), fg, dim3(256), 0, s, scratch, splits, bias, res, ldr, C, ldc, M, N)
    if (hb && hr) SKF(true, true); else if (hb) SKF(true, false); else if (hr) SKF(false, true); else SKF(false, false);
#undef SKF
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// Default (variant, splits) of a decode-shaped GEMM, from tools/decode_gemm_tune.py sweeps on MI355X (7B shapes, M = 64 / 256):
// wide outputs (gate/up, lm_head) stream W with one tile per workgroup and no split; narrow outputs (qkv, o, down) have too
// few column tiles to fill 256 CUs and split K into ~256-512 workgroups.  A host-side autotuner may override the choice
// through st_gemm_nt_decode_variant.
// Split count of a one-workgroup-per-CU decode tile: whole rounds over the CUs, each costing its K-tiles plus ~20 K-tile steps of
// prologue / fp32 slab epilogue (fitted on MI355X: 256x128 tile, 7B down projection at 256 / 512 rows: 78 / 86 us at 4 slices,
// 53 / 93 at 9 — one full round beats 1.1 rounds by a third).
static int decode_fill_split(int tiles, int kt, int max_sp) {
    const int ncu = st_num_cus();
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= max_sp && kt / sp >= 4; ++sp) {
        const double cost = (double)st_cdiv(tiles * sp, ncu) * (st_cdiv(kt, sp) + 20.0);
        if (cost < best_cost - 1e-9) { best = sp; best_cost = cost; }
    }
    return best;
}

static void decode_plan(int M, int N, int K, int64_t scratch_elems, int* variant, int* splits) {
    const int bm = M <= 64 ? 64 : (M <= 128 ? 128 : 256);
    const bool wide = N >= 16384;
    const int mt = st_cdiv(M, 256);                          // 256-row tiles (2 for the 257..512-row decode batches)
    int v, bn, row_tiles = mt;
    if (bm == 64) { v = (wide || K >= 8192) ? 11 : 10; bn = v == 11 ? 128 : 64; }
    else if (bm == 128) { v = (wide || K >= 8192) ? 14 : 13; bn = v == 14 ? 128 : 64; }     // down at 128 rows: 128x128 x 9 slices 39 us vs 62
    else if (wide) {
        // 256x256 tiles are ~10 % faster per flop than 256x128 but quantise worse on 256 CUs: compare the last-round fill
        const int t18 = mt * st_cdiv(N, 256), t16 = mt * st_cdiv(N, 128);
        const double e18 = 1.1 * t18 / (double)(st_cdiv(t18, 256) * 256), e16 = t16 / (double)(st_cdiv(t16, 256) * 256);
        v = e18 >= e16 ? 18 : 16; bn = v == 18 ? 256 : 128;
    } else if (K >= 8192) { v = 16; bn = 128; }
    else if (N >= 4096 && mt > 1) { v = 16; bn = 128; }                  // qkv at 257..512 rows: 36 us vs 48 (128x64 tiles), tools/decode_gemm_tune.py
    else if (N >= 4096) { v = 14; bn = 128; row_tiles = st_cdiv(M, 128); }   // qkv at <= 256 rows: 25 us vs 34
    else { v = 13; bn = 64; row_tiles = st_cdiv(M, 128); }
    int sp = 1;
    if (!wide) {
        const int tiles = row_tiles * st_cdiv(N, bn);
        sp = (256 + tiles / 2) / tiles;
        if (bm == 64 && sp < 4 && tiles <= 128) sp = 4;
        if (bm == 64 && K >= 8192) sp = 8;
        if (sp > 8) sp = 8;
        // round 5: <= 64 rows too (ST_DECODE_SPLIT64=0: the older rule above).  The launch takes what the BUSIEST CU stages: 7B qkv at 64 rows
        // was 72 tiles x 4 slices = 288 workgroups (32 CUs ran two), now 72 x 3 = 216; o 56 x 5 = 280 -> 56 x 4 = 224; down 28 x 8 -> 28 x 9
        static const bool split64 = [] { const char* e = getenv("ST_DECODE_SPLIT64"); return !e || e[0] != '0'; }();
        if (bm >= 128 || split64) sp = decode_fill_split(tiles, K / 64, 12);       // fill whole rounds of CUs (tools/decode_gemm_tune.py, M = 128 / 256 / 512)
        while (sp > 1 && (K / 64) / sp < 4) --sp;                       // keep >= 4 K-tiles per slice
        while (sp > 1 && (int64_t)sp * M * N > scratch_elems) --sp;
        if (sp < 1) sp = 1;
    }
    *variant = v; *splits = sp;
}

extern "C" int st_gemm_decode_plan(int M, int N, int K, int64_t scratch_elems, int* variant_out, int* splits_out) {
    if (M <= 0 || M > ST_DECODE_MAX_ROWS || N <= 0 || K <= 0 || (K % BK) || !variant_out || !splits_out) return ST_EINVAL;
    int v, sp;
    decode_plan(M, N, K, scratch_elems, &v, &sp);
    const int kt_per = st_cdiv(K / 64, sp);
    *variant_out = v; *splits_out = st_cdiv(K / 64, kt_per);             // slices actually launched
    return 0;
}

/* tuning entry: explicit tile variant (st_gemm_tile_decode ids) and split count for a decode-shaped GEMM */
extern "C" int st_gemm_nt_decode_variant(int variant, int splits, const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb,
                                         const st_bf16* bias, const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, int64_t ldc,
                                         float* scratch, int64_t scratch_elems, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || !out_bf16 || M <= 0 || M > ST_DECODE_MAX_ROWS || N <= 0 || K <= 0 || (K % BK) || (lda & 7) || (ldb & 7) || lda < K || ldb < K ||
        ldc < N || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0) {                                     // the register-streaming kernel (kept for comparison; picks its own split)
        if (M <= 32) return launch_skinny<2>(A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K, s);
        if (M <= 64) return launch_skinny<4>(A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K, s);
        if (M <= 128) return launch_skinny<8>(A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K, s);
        return launch_skinny<16>(A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K, s);
    }
    return launch_decode_tiles(variant, splits, A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K, s);
}

/* decode-shaped GEMM that stops at the fp32 split-K slabs [splits][M][N] (no finish): the caller's fused epilogue
 * (st_decode_finish_norm / st_decode_finish_qkv) sums them.  *splits_out = number of slabs written (>= 1). */
extern "C" int st_gemm_nt_decode_slabs(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, float* scratch,
                                       int64_t scratch_elems, int M, int N, int K, int* splits_out, st_stream_t stream) {
    if (!A || !B || !scratch || !splits_out || M <= 0 || M > ST_DECODE_MAX_ROWS || N <= 0 || K <= 0 || (K % BK) || (lda & 7) || (ldb & 7) || lda < K ||
        ldb < K || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15) || scratch_elems < (int64_t)M * N)
        return ST_EINVAL;
    int variant, splits;
    decode_plan(M, N, K, scratch_elems, &variant, &splits);
    const int kt_per = st_cdiv(K / 64, splits);
    splits = st_cdiv(K / 64, kt_per);
    *splits_out = splits;
    // splits == 1 also goes through the slab output (one fp32 slab) so the fused epilogue has a single input format
    return st_gemm_tile_decode(variant, splits > 1 ? splits : -1, A, lda, B, ldb, nullptr, nullptr, 0, nullptr, scratch, M, N, K, N,
                               (hipStream_t)stream);
}

/* decode-shaped GEMM (M <= 256): out_bf16 = A B^T (+bias)(+residual).  scratch (scratch_elems floats, contents
 * irrelevant) holds the split-K partial slabs [split][M][N]; NULL disables split-K. */
extern "C" int st_gemm_nt_skinny(const st_bf16* A, int64_t lda, const st_bf16* B, int64_t ldb, const st_bf16* bias,
                                 const st_bf16* residual, int64_t ldr, st_bf16* out_bf16, int64_t ldc, float* scratch,
                                 int64_t scratch_elems, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || !out_bf16 || M <= 0 || M > ST_DECODE_MAX_ROWS || N <= 0 || K <= 0 || (K % BK) || (lda & 7) || (ldb & 7) || lda < K || ldb < K ||
        ldc < N || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))
        return ST_EINVAL;
    int variant, splits;
    decode_plan(M, N, K, scratch ? scratch_elems : 0, &variant, &splits);
    return launch_decode_tiles(variant, splits, A, lda, B, ldb, bias, residual, ldr, out_bf16, ldc, scratch, scratch_elems, M, N, K,
                               (hipStream_t)stream);
}
