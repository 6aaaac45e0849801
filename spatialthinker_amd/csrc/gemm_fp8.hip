// gemm_fp8.hip — block-scaled fp8 ("MX-fp8": OCP e4m3fn elements, one e8m0 power-of-two scale per 32 consecutive k) GEMM for
// gfx950, BASELINE.json config #5 ("fp8 MFMA"):   C[M,N] (bf16) = dequant(Aq)[M,K] * dequant(Bq)[N,K]^T  (+bias)(+residual).
//
// Why the scaled form: on CDNA4 the plain fp8 MFMA (16x16x32 / 32x32x16) runs at the bf16 rate; only
// v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 per instruction, hardware dequantisation by the per-lane e8m0 scale) reaches the
// ~5 PF dense fp8 peak (MI355X_MICROARCH.md, MFMA table).  One lane of that instruction holds 32 consecutive k-bytes of one row
// — exactly one MX block — and supplies that block's scale byte, so the operand layout is the natural one.
//
// Structure = the bf16 256x256 tile (gemm_tiles.hip) with k counted in BYTES: 8 waves (4 x 2, 64 x 128 per wave), K-tile = 128
// fp8 = 128-byte operand rows, so the LDS image, its XOR swizzle (16-byte chunk ^ (row >> 1) & 7, applied on the LDS-DMA source
// address and on the fragment read) and the 64 KiB/stage budget are unchanged while every byte carries twice the flops.  Two LDS
// stages; the LDS-DMA of K-tile t+1 flies under the 32 MFMAs of tile t; one barrier per tile.
// Scales travel beside the operands: stored K-tile-major, S[kt][row] = one dword holding the 4 block scales of that row inside
// K-tile kt, so a tile's 256 row-dwords are ONE 1-KiB LDS-DMA instruction; a lane reads its row's dword and shifts its own block's
// byte down (the instruction's op_sel picks a byte per INSTRUCTION, the block index is per lane group).
// Roofline: MFMA-bound, 2*M*N*K flop per launch against the 5 PF dense fp8 peak.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ void glds16q(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// ------------------------------------------------------------------------------ quantiser
// x (R, K) bf16 row-major -> q (R, K) e4m3 bytes + scales S[K/128][Rs] dwords (byte j of S[kt][r] = e8m0 scale of k-block 4*kt+j of
// row r).  OCP MX rule: shared exponent = floor(log2(amax)) - 8 (e4m3's largest binade), elements = RNE(x * 2^-shared), saturated to
// +-448.  One wave per (row, 512 k): lane = 8 consecutive k (16-byte load), 4 lanes = one block, 16 lanes = one K-tile.
__global__ __launch_bounds__(256) void mxfp8_quantize_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                            int64_t ldq, uint32_t* __restrict__ scales, int64_t scale_rows, int R, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunks = K / 512 + ((K % 512) ? 1 : 0);
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= (int64_t)R * chunks) return;
    const int r = (int)(item / chunks), k0 = (int)(item % chunks) * 512 + lane * 8;
    const bool live = k0 < K;                                 // K is a multiple of 128: a lane is all-in or all-out
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) unpack8(*reinterpret_cast<const uint4*>(x + (int64_t)r * ldx + k0), f);
    uint32_t w[2];
    const int e = mx_quant8(f, w);
    if (live) *reinterpret_cast<uint2*>(q + (int64_t)r * ldq + k0) = make_uint2(w[0], w[1]);
    const uint32_t sd = mx_scale_dword(e, lane);
    if (live && (lane & 15) == 0) scales[(int64_t)(k0 >> 7) * scale_rows + r] = sd;
}

// Transposing quantiser (weight-gradient GEMMs, dW = dY^T X: the contraction runs over the TOKEN index, so both operands are needed
// token-minor with MX blocks of 32 consecutive tokens): x (R, C) bf16 row-major -> qT (C, R) e4m3 bytes + scales S[R/128][Cs] dwords (byte j of
// S[kt][c] = scale of rows 128 kt + 32 j .. + 31 of column c) = st_mxfp8_quantize of the transposed matrix, without materialising it.
// R % 128 == 0 (packed token counts are padded to 128; the padding rows are zero).  One workgroup per 128 rows x 64 columns: the tile passes
// through LDS, then thread (column c = t >> 2, block j = t & 3) owns one whole MX block — no cross-lane reduction — and writes its 32 bytes.
// (A 128 x 128 tile with two columns per thread — one LDS dword per row — measured 20 % SLOWER in the bench: 33 KiB of LDS and 64 live
// values per thread cost more occupancy than the halved LDS instruction count buys.)
// BOTH: the same pass also writes the ordinary (row-wise) quantisation q / srow of x — a gradient that feeds an fp8 input-gradient GEMM
// (MX blocks along its features) AND an fp8 weight-gradient GEMM (blocks along the tokens) is read once; a thread quantises the 32
// columns it has just loaded (one whole MX block) from its registers.  Needs C % 32 == 0 for that part (C % 128 == 0 at the call sites).
template <bool BOTH>
__global__ __launch_bounds__(256) void mxfp8_quantize_t_kernel(const uint16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ qT, int64_t ldq,
                                                              uint32_t* __restrict__ scales, int64_t scale_rows, int R, int C,
                                                              uint8_t* __restrict__ qrow, int64_t ldqr, uint32_t* __restrict__ srow, int64_t srow_rows) {
    __shared__ uint16_t tile[128][64 + 2];                    // +2: rows 132 bytes apart; a wave's column reads (64 consecutive columns of one row) are conflict-free
    const int t = threadIdx.x;
    const int r0 = blockIdx.y * 128, c0 = blockIdx.x * 64;
    {
        const int row = t >> 1, cb = (t & 1) * 32;
        float fr[32];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + cb + k * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < C) v = *reinterpret_cast<const uint4*>(x + (int64_t)(r0 + row) * ldx + c);       // C % 8 == 0: all-in or all-out
            uint32_t* d = reinterpret_cast<uint32_t*>(&tile[row][(cb + k * 8 + 32 * (row >> 6)) & 63]);     // rows 64.. rotated by 32 columns (see the reads)
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            if constexpr (BOTH) {
                float f8[8];
                unpack8(v, f8);
#pragma unroll
                for (int i = 0; i < 8; ++i) fr[k * 8 + i] = f8[i];
            }
        }
        if constexpr (BOTH) {
            if (c0 + cb < C) {                                    // C % 32 == 0: the thread's 32 columns are one whole MX block
                float amax = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(fr[i]));
                int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
                e = e < 0 ? 0 : (e > 254 ? 254 : e);
                const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
                uint32_t w[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    float v4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v4[i] = fminf(fmaxf(fr[4 * h + i] * inv, -448.f), 448.f);
                    int packed = __builtin_amdgcn_cvt_pk_fp8_f32(v4[0], v4[1], 0, false);
                    packed = __builtin_amdgcn_cvt_pk_fp8_f32(v4[2], v4[3], packed, true);
                    w[h] = (uint32_t)packed;
                }
                uint4* dst = reinterpret_cast<uint4*>(qrow + (int64_t)(r0 + row) * ldqr + c0 + cb);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
                const int col = c0 + cb;
                reinterpret_cast<uint8_t*>(srow + (int64_t)(col >> 7) * srow_rows + r0 + row)[(col >> 5) & 3] = (uint8_t)e;
            }
        }
    }
    __syncthreads();
    // thread (column c = t >> 2, block j = t & 3): the four blocks of a column sit in four consecutive lanes, so a column's 128 output bytes
    // leave as one contiguous 128-byte segment (with c = t & 63 a wave wrote 64 scattered 32-byte pieces: 2.8 TB/s).  LDS: rows 32 apart
    // are 32 banks apart (33-dword pitch), rows 64 apart would collide — they are stored rotated by 32 columns (16 dwords)
    const int c = t >> 2, j = t & 3;
    const int cr = (c + 32 * (j >> 1)) & 63;
    float f[32];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        f[i] = bf2f(tile[j * 32 + i][cr]);
        amax = fmaxf(amax, fabsf(f[i]));
    }
    int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
    e = e < 0 ? 0 : (e > 254 ? 254 : e);
    const float inv = __uint_as_float((uint32_t)(254 - e) << 23);
    uint32_t w[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(f[4 * h + i] * inv, -448.f), 448.f);
        int packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], packed, true);
        w[h] = (uint32_t)packed;
    }
    if (c0 + c < C) {
        uint4* dst = reinterpret_cast<uint4*>(qT + (int64_t)(c0 + c) * ldq + r0 + j * 32);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        reinterpret_cast<uint8_t*>(scales + (int64_t)blockIdx.y * scale_rows + c0 + c)[j] = (uint8_t)e;
    }
}

// ------------------------------------------------------------------------------ GEMM
#define Q_BM 256
#define Q_BN 256
#define Q_OP_BYTES (256 * 128)                  // one operand tile: 256 rows x 128 k-bytes
#define Q_STAGE (2 * Q_OP_BYTES + 2048)         // A, B, A scales (1 KiB), B scales (1 KiB)

template <bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(512) void gemm_mxfp8_kernel(const uint8_t* __restrict__ A, int64_t lda, const uint32_t* __restrict__ SA,
                                                        int64_t sa_rows, const uint8_t* __restrict__ B, int64_t ldb,
                                                        const uint32_t* __restrict__ SB, int64_t sb_rows,
                                                        const uint16_t* __restrict__ bias, const uint16_t* __restrict__ res, int64_t ldr,
                                                        uint16_t* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int TM = 4, TN = 8;                 // 16x16 MFMA tiles per wave: 64 rows x 128 columns
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {                                             // XCD-aware bijective remap + groups of 8 tile rows (as gemm_tiles.hip)
        const int xcd = bid & 7, idx = bid >> 3, qq = nb >> 3, r = nb & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int per_group = 8 * tiles_n, group = bid / per_group, in_g = bid % per_group;
    const int first_m = group * 8, gsz = min(tiles_m - first_m, 8);
    const int m0 = (first_m + in_g % gsz) * Q_BM, n0 = (in_g / gsz) * Q_BN;

    auto stage = [&](int kt, char* dst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int inst = wave * 4 + j, p = inst * 64 + lane, r = p >> 3, kc = (p & 7) ^ ((r >> 1) & 7);
            int ga = m0 + r; ga = ga < M ? ga : M - 1;
            glds16q(A + (int64_t)ga * lda + (int64_t)kt * 128 + kc * 16, dst + inst * 1024);
            int gb = n0 + r; gb = gb < N ? gb : N - 1;
            glds16q(B + (int64_t)gb * ldb + (int64_t)kt * 128 + kc * 16, dst + Q_OP_BYTES + inst * 1024);
        }
        if (wave < 2) {                           // the 256 row-scales of the tile: one instruction per operand
            const uint32_t* S = wave == 0 ? SA : SB;
            const int64_t rows = wave == 0 ? sa_rows : sb_rows;
            int r4 = (wave == 0 ? m0 : n0) + lane * 4;
            r4 = r4 + 4 <= rows ? r4 : (int)rows - 4;
            glds16q(S + (int64_t)kt * rows + r4, dst + 2 * Q_OP_BYTES + wave * 1024);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / 128;
    stage(0, smem);
    // Operand layout of the K = 128 instruction (probed on MI355X, tools/probes/mx_layout_probe.hip): lane group g = lane >> 4 holds
    // k = 16g .. 16g+15 in its first 16 bytes and k = 64+16g .. 64+16g+15 in the second 16 (two K = 64 halves), while the scale of
    // MX block b (k = 32b .. 32b+31) is taken from lane row + 16b.  So a lane reads 16-byte chunks g and 4+g of its 128-byte row and
    // supplies the scale byte of block g.
    const int frow = lane & 15, kb = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile kt has landed (this wave's pieces; the barrier publishes all)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * Q_STAGE);
        const char* la = smem + (kt & 1) * Q_STAGE;
        const char* lb = la + Q_OP_BYTES;
        const uint32_t* sa = reinterpret_cast<const uint32_t*>(la + 2 * Q_OP_BYTES);
        const uint32_t* sb = sa + 256;
        i32x8 af[TM];
        int sca[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ra = wm * 64 + i * 16 + frow, sw = (ra >> 1) & 7;
            const int4 lo = *reinterpret_cast<const int4*>(la + ra * 128 + ((kb ^ sw) << 4));
            const int4 hi = *reinterpret_cast<const int4*>(la + ra * 128 + (((4 + kb) ^ sw) << 4));
            af[i] = (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            sca[i] = (int)((sa[ra] >> (8 * kb)) & 0xffu);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int rb = wn * 128 + ni * 16 + frow, sw = (rb >> 1) & 7;
            const int4 lo = *reinterpret_cast<const int4*>(lb + rb * 128 + ((kb ^ sw) << 4));
            const int4 hi = *reinterpret_cast<const int4*>(lb + rb * 128 + (((4 + kb) ^ sw) << 4));
            const i32x8 bf = (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const int scb = (int)((sb[rb] >> (8 * kb)) & 0xffu);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)   // swapped operands (D rows = n, columns = m): a lane ends up with 4 consecutive output columns
                acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf, af[mi], acc[ni][mi], 0, 0, 0, scb, 0, sca[mi]);
        }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n0 + wn * 128 + ni * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= N) break;
                if (HAS_BIAS) v[r] += bf2f(bias[n + r]);
                if (HAS_RES) v[r] += bf2f(res[(int64_t)m * ldr + n + r]);
            }
            uint16_t* cp = C + (int64_t)m * ldc + n;
            if (n + 3 < N && ((ldc & 3) == 0)) {
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

// gemm_mx4.hip: the 4-wave hand-scheduled tile (default); ST_FP8_TILE=8 keeps the 8-wave tile above for A/B runs
int st_launch_gemm_mx4(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                       int64_t sb_rows, const uint16_t* bias, const uint16_t* residual, int64_t ldr, uint16_t* out, int64_t ldc, int M, int N,
                       int K, int dbg, hipStream_t s);
int st_launch_gemm_mx4_swiglu(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                              int64_t sb_rows, void* out, int64_t ldc, uint32_t* sq, int64_t sq_rows, int M, int N, int K, hipStream_t s);
int st_launch_gemm_mx4_f32(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb, const uint32_t* SB,
                           int64_t sb_rows, float* out, int64_t ldc, int accumulate, int M, int N, int K, hipStream_t s);
static int g_fp8_tile = -1;
static int fp8_tile_waves() {
    if (g_fp8_tile < 0) { const char* e = getenv("ST_FP8_TILE"); g_fp8_tile = (e && atoi(e) == 8) ? 8 : 4; }
    return g_fp8_tile;
}

extern "C" {

int st_gemm_mxfp8_select(int waves) {
    // 41..44 are timing experiments with WRONG results: only a process that asks for them by name (tools/mx4_ksweep.py) may select them
    static const bool experiments = [] { const char* e = getenv("ST_FP8_TIMING_EXPERIMENTS"); return e && e[0] == '1'; }();
    if (waves >= 41 && waves <= 44 && !experiments) return ST_EINVAL;
    if (waves != 4 && waves != 8 && !(waves >= 41 && waves <= 44) && waves != 50 && waves != 52 && waves != 56) return ST_EINVAL;
    g_fp8_tile = waves;
    return 0;
}

int st_mxfp8_quantize(const st_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows, int R, int K,
                      st_stream_t stream) {
    if (!x || !q || !scales || R <= 0 || K <= 0 || (K % 128) || (ldx & 7) || (ldq & 7) || ldx < K || ldq < K || scale_rows < R ||
        (scale_rows & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 7))
        return ST_EINVAL;
    const int chunks = (K + 511) / 512;
    hipLaunchKernelGGL(mxfp8_quantize_kernel, dim3(st_cdiv((int64_t)R * chunks, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, ldq,
                       scales, scale_rows, R, K);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_mxfp8_quantize_t(const st_bf16* x, int64_t ldx, uint8_t* qT, int64_t ldq, uint32_t* scales, int64_t scale_rows, int R, int C,
                        st_stream_t stream) {
    if (!x || !qT || !scales || R <= 0 || C <= 0 || (R % 128) || (C & 7) || (ldx & 7) || (ldq & 15) || ldx < C || ldq < R || scale_rows < C ||
        (((uintptr_t)x) & 15) || (((uintptr_t)qT) & 15))
        return ST_EINVAL;
    hipLaunchKernelGGL(mxfp8_quantize_t_kernel<false>, dim3(st_cdiv(C, 64), R / 128), dim3(256), 0, (hipStream_t)stream, x, ldx, qT, ldq, scales,
                       scale_rows, R, C, (uint8_t*)nullptr, (int64_t)0, (uint32_t*)nullptr, (int64_t)0);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_mxfp8_quantize_both(const st_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, uint32_t* scales, int64_t scale_rows, uint8_t* qT, int64_t ldqT,
                           uint32_t* scales_t, int64_t scale_t_rows, int R, int C, st_stream_t stream) {
    if (!x || !q || !scales || !qT || !scales_t || R <= 0 || C <= 0 || (R % 128) || (C % 128) || (ldx & 7) || (ldq & 15) || (ldqT & 15) || ldx < C ||
        ldq < C || ldqT < R || scale_rows < R || scale_t_rows < C || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 15) || (((uintptr_t)qT) & 15))
        return ST_EINVAL;
    hipLaunchKernelGGL(mxfp8_quantize_t_kernel<true>, dim3(st_cdiv(C, 64), R / 128), dim3(256), 0, (hipStream_t)stream, x, ldx, qT, ldqT, scales_t,
                       scale_t_rows, R, C, q, ldq, scales, scale_rows);
    ST_CHECK_LAUNCH();
    return 0;
}

int st_gemm_mxfp8_nt_f32(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                         const uint32_t* SB, int64_t sb_rows, float* out, int64_t ldc, int accumulate, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || !SA || !SB || !out || M <= 0 || N <= 0 || K <= 0 || (K % 128) || (lda & 15) || (ldb & 15) || lda < K || ldb < K ||
        ldc < N || sa_rows < M || sb_rows < N || (sa_rows & 3) || (sb_rows & 3) || sa_rows < 4 || sb_rows < 4 || (((uintptr_t)A) & 15) ||
        (((uintptr_t)B) & 15) || (((uintptr_t)SA) & 15) || (((uintptr_t)SB) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM_FP8, s, 2.0 * (double)M * (double)N * (double)K);
    return st_launch_gemm_mx4_f32(A, lda, SA, sa_rows, B, ldb, SB, sb_rows, out, ldc, accumulate, M, N, K, s);
}

int st_gemm_mxfp8_nt(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                     const uint32_t* SB, int64_t sb_rows, const st_bf16* bias, const st_bf16* residual, int64_t ldr, st_bf16* out,
                     int64_t ldc, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || !SA || !SB || !out || M <= 0 || N <= 0 || K <= 0 || (K % 128) || (lda & 15) || (ldb & 15) || lda < K || ldb < K ||
        ldc < N || sa_rows < M || sb_rows < N || (sa_rows & 3) || (sb_rows & 3) || sa_rows < 4 || sb_rows < 4 || (((uintptr_t)A) & 15) ||
        (((uintptr_t)B) & 15) || (((uintptr_t)SA) & 15) || (((uintptr_t)SB) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int tiles_m = st_cdiv(M, Q_BM), tiles_n = st_cdiv(N, Q_BN);
    constexpr int smem = 2 * Q_STAGE;
    StProfScope ps(ST_K_GEMM_FP8, s, 2.0 * (double)M * (double)N * (double)K);
    if (fp8_tile_waves() != 8)        // 4, or 41..44 = the 4-wave tile's timing experiments (tools/mx4_ksweep.py; wrong results), 50 / 52 / 56 = its K-tile schedules 0 / 2 / 6 (correct)
        return st_launch_gemm_mx4(A, lda, SA, sa_rows, B, ldb, SB, sb_rows, bias, residual, ldr, out, ldc, M, N, K,
                                  fp8_tile_waves() > 40 ? fp8_tile_waves() - 40 : 0, s);
#define QGO(HB, HR)                                                                                                              \
    do {                                                                                                                         \
        auto kern = gemm_mxfp8_kernel<HB, HR>;                                                                                   \
        static bool configured = false;                                                                                          \
        if (!configured) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); configured = true; } \
        hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), smem, s, A, lda, SA, sa_rows, B, ldb, SB, sb_rows, bias, residual, ldr,  \
                           out, ldc, M, N, K, tiles_m, tiles_n);                                                                 \
    } while (0)
    if (bias && residual) QGO(true, true); else if (bias) QGO(true, false); else if (residual) QGO(false, true); else QGO(false, false);
#undef QGO
    ST_CHECK_LAUNCH();
    return 0;
}

int st_gemm_mxfp8_swiglu(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                         const uint32_t* SB, int64_t sb_rows, st_bf16* out, int64_t ldc, int M, int N, int K, st_stream_t stream) {
    if (!A || !B || !SA || !SB || !out || M <= 0 || N <= 0 || K <= 0 || (K % 128) || (lda & 15) || (ldb & 15) || lda < K || ldb < K ||
        ldc < N || sa_rows < M || sb_rows < 2 * (int64_t)N || (sa_rows & 3) || (sb_rows & 3) || sa_rows < 4 || (((uintptr_t)A) & 15) ||
        (((uintptr_t)B) & 15) || (((uintptr_t)SA) & 15) || (((uintptr_t)SB) & 15))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM_FP8, s, 4.0 * (double)M * (double)N * (double)K);
    return st_launch_gemm_mx4_swiglu(A, lda, SA, sa_rows, B, ldb, SB, sb_rows, out, ldc, nullptr, 0, M, N, K, s);
}

int st_gemm_mxfp8_swiglu_q(const uint8_t* A, int64_t lda, const uint32_t* SA, int64_t sa_rows, const uint8_t* B, int64_t ldb,
                           const uint32_t* SB, int64_t sb_rows, uint8_t* q, int64_t ldq, uint32_t* sq, int64_t sq_rows, int M, int N, int K,
                           st_stream_t stream) {
    if (!A || !B || !SA || !SB || !q || !sq || M <= 0 || N <= 0 || K <= 0 || (K % 128) || (N % 128) || (lda & 15) || (ldb & 15) || lda < K ||
        ldb < K || ldq < N || (ldq & 7) || sa_rows < M || sb_rows < 2 * (int64_t)N || sq_rows < M || (sa_rows & 3) || (sb_rows & 3) ||
        (sq_rows & 3) || sa_rows < 4 || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15) || (((uintptr_t)SA) & 15) || (((uintptr_t)SB) & 15) ||
        (((uintptr_t)q) & 7))
        return ST_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    StProfScope ps(ST_K_GEMM_FP8, s, 4.0 * (double)M * (double)N * (double)K);
    return st_launch_gemm_mx4_swiglu(A, lda, SA, sa_rows, B, ldb, SB, sb_rows, q, ldq, sq, sq_rows, M, N, K, s);
}

}  // extern "C"
