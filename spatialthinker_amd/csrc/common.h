// common.h — shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/st_hip.h"

#define ST_WAVE 64

#define ST_DECODE_MAX_ROWS 512   /* rows of a decode-shaped GEMM (sequences decoded together): one or two 256-row tiles */
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (one VALU op per PAIR).  The integer formulation
// (add 0x7fff + lsb, NaN test) costs ~20 instructions and an exec-mask branch per element — it was 10 us of an 87-us GEMM tile.
typedef __bf16 st_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float st_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {        // two values -> one packed dword (lo in bits 0..15)
    const st_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, st_bf16x2_t));
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

struct alignas(16) bf8 { uint16_t v[8]; };                      // one 16-byte global/LDS access

__device__ __forceinline__ void unpack8(const uint4& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 r;
    r.x = f2bf2(f[0], f[1]); r.y = f2bf2(f[2], f[3]); r.z = f2bf2(f[4], f[5]); r.w = f2bf2(f[6], f[7]);
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- MX-fp8 (OCP e4m3 elements, one e8m0 scale per 32 consecutive k): the quantiser's step for 8 consecutive k held by one lane; the 4
// lanes of a block must be 4 consecutive lanes starting at a multiple of 4.  Shared exponent = floor(log2(amax)) - 8 (e4m3's largest
// binade), elements = RNE(x * 2^-shared), saturated to +-448.  Returns the biased e8m0 scale byte; w = the 8 e4m3 bytes.
__device__ __forceinline__ int mx_quant8(const float (&f)[8], uint32_t (&w)[2]) {
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;  // biased exponent of the scale 2^(floor(log2 amax) - 8)
    e = e < 0 ? 0 : (e > 254 ? 254 : e);                      // amax == 0 (or subnormal) -> smallest scale, all elements quantise to 0
    const float inv = __uint_as_float((uint32_t)(254 - e) << 23);   // 2^-(e - 127), exact
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(f[4 * h + j] * inv, -448.f), 448.f);
        int packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], packed, true);
        w[h] = (uint32_t)packed;
    }
    return e;
}
// the 4 block scales of a 128-k tile sit in lanes 0, 4, 8, 12 of its 16-lane group: the tile's scale dword (valid in every lane of the group)
__device__ __forceinline__ uint32_t mx_scale_dword(int e, int lane) {
    const int base = lane & 48;
    const uint32_t s0 = (uint32_t)__shfl(e, base, 64), s1 = (uint32_t)__shfl(e, base + 4, 64);
    const uint32_t s2 = (uint32_t)__shfl(e, base + 8, 64), s3 = (uint32_t)__shfl(e, base + 12, 64);
    return s0 | (s1 << 8) | (s2 << 16) | (s3 << 24);
}

// ---- gfx950 data-movement helpers ------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

// LDS-DMA: every lane copies 16 B global -> LDS; the LDS destination is wave-uniform base + lane*16 (lane-linear image).
__device__ __forceinline__ void st_glds16(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}
// the same copy with the non-temporal policy (aux = 2): for streams ONE workgroup reads once (decode K/V, decode weights) — MI355X_MICROARCH.md
// row nt-weights: issued -> landed -18 %; never for data other workgroups re-read from L2
__device__ __forceinline__ void st_glds16_nt(const void* gsrc, char* lds_dst_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 2);
}
// Hardware transpose read: within a 16-lane group whose lane s points at row (s>>2), column 4*(s&3) of a [4][16] bf16 block,
// lane i receives column i (the 4 rows) — i.e. 4 contiguous k-values of a k-strided MFMA operand.
__device__ __forceinline__ s16x4_t st_lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
// two fp32 -> packed bf16 pair with the hardware RNE convert (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t st_pk_bf16(float a, float b) {
    hw_f32x2_t f = {a, b};
    hw_bf16x2_t h = __builtin_convertvector(f, hw_bf16x2_t);
    return *reinterpret_cast<uint32_t*>(&h);
}
// max / sum across the two 32-lane halves without touching the LDS crossbar (v_permlane32_swap)
__device__ __forceinline__ float st_half_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float st_half_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- launch / profiling plumbing (host side) -------------------------------------------------
struct StProf {
    bool on = false;
    int cap = 0, n = 0, stride = 1;
    long long seen = 0;            // eligible (non-captured) launches since st_prof_enable, sampled or not
    hipEvent_t* ev = nullptr;      // 2*cap events
    double units = 0.0;            // algorithmic flops / bytes of the SAMPLED launches
    double pending = 0.0;          // st_prof_hint_units: units of the next launch when the launcher cannot know them (attention)
    double* unit_each = nullptr;   // cap entries: units of sampled launch i (st_prof_read_events)
    unsigned long long* tags = nullptr;   // cap entries: the launcher's shape tag of sampled launch i (st_prof_tag), 0 when it gives none
};
// shape tag of a GEMM launch: form (0 = NT, 1 = NN, 2 = TN, 3 = SwiGLU) | epilogue flags (1 bias, 2 residual, 4 fp32 out, 8 accumulate) | M, N, K
static inline unsigned long long st_prof_tag(int form, int flags, int M, int N, int K) {
    return ((unsigned long long)(form & 15) << 60) | ((unsigned long long)(flags & 15) << 56) | ((unsigned long long)(K & 0x3ffff) << 36) |
           ((unsigned long long)(N & 0x3ffff) << 18) | (unsigned long long)(M & 0x3ffff);
}
extern StProf g_prof[ST_K_COUNT];

struct StProfScope {
    int k; hipStream_t s; bool live;
    StProfScope(int klass, hipStream_t st, double units, unsigned long long tag = 0) : k(klass), s(st), live(false) {
        StProf& p = g_prof[k];
        if (p.on) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            hipStreamIsCapturing(s, &cs);                       // launches captured into a hipGraph are not timed
            if (cs == hipStreamCaptureStatusNone) {
                const bool pick = (p.seen % p.stride) == 0 && p.n < p.cap;      // every stride-th launch: the cap spans the whole run
                p.seen++;
                if (pick) {
                    live = true; hipEventRecord(p.ev[2 * p.n], s);
                    const double u = units > 0.0 ? units : p.pending;
                    p.units += u; p.unit_each[p.n] = u; p.tags[p.n] = tag;
                }
            }
            p.pending = 0.0;
        }
    }
    ~StProfScope() {
        if (live) { StProf& p = g_prof[k]; hipEventRecord(p.ev[2 * p.n + 1], s); p.n++; }
    }
};

#define ST_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
static inline int st_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int st_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

