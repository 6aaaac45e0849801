// common.h — shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/st_hip.h"

#define ST_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {            // round-to-nearest-even, NaN-preserving
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

struct alignas(16) bf8 { uint16_t v[8]; };                      // one 16-byte global/LDS access

__device__ __forceinline__ void unpack8(const uint4& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 r;
    r.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    r.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    r.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    r.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
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

// ---- launch / profiling plumbing (host side) -------------------------------------------------
struct StProf {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;      // 2*cap events
    double units = 0.0;
};
extern StProf g_prof[ST_K_COUNT];

struct StProfScope {
    int k; hipStream_t s; bool live;
    StProfScope(int klass, hipStream_t st, double units) : k(klass), s(st), live(false) {
        StProf& p = g_prof[k];
        if (p.on && p.n < p.cap) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            hipStreamIsCapturing(s, &cs);                       // launches captured into a hipGraph are not timed
            if (cs == hipStreamCaptureStatusNone) { live = true; hipEventRecord(p.ev[2 * p.n], s); p.units += units; }
        }
    }
    ~StProfScope() {
        if (live) { StProf& p = g_prof[k]; hipEventRecord(p.ev[2 * p.n + 1], s); p.n++; }
    }
};

#define ST_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
static inline int st_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
