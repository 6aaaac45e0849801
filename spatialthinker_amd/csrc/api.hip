// api.hip — version + profiling hooks of the C ABI.
#include "common.h"

StProf g_prof[ST_K_COUNT];

// shader-clock probe: s_memtime counts shader cycles, s_memrealtime the constant 100-MHz reference — their ratio over a short spin is the
// clock the chip's power management grants AT THAT MOMENT (MI355X_MICROARCH.md "DVFS give-back": 1.9-2.3 GHz by load, 2.4 GHz max)
__global__ void clock_probe_kernel(unsigned long long* out, int spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < (unsigned long long)spin_ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
}

extern "C" {
int st_version(void) { return 1; }
const char* st_arch(void) { return "gfx950"; }

int st_prof_enable(int klass, int max_events) {
    if (klass < 0 || klass >= ST_K_COUNT || max_events <= 0) return ST_EINVAL;
    StProf& p = g_prof[klass];
    if (p.ev) { for (int i = 0; i < 2 * p.cap; ++i) hipEventDestroy(p.ev[i]); delete[] p.ev; delete[] p.unit_each; delete[] p.tags; }
    p.ev = new hipEvent_t[2 * max_events];
    p.unit_each = new double[max_events];
    p.tags = new unsigned long long[max_events];
    for (int i = 0; i < 2 * max_events; ++i) hipEventCreate(&p.ev[i]);
    p.cap = max_events; p.n = 0; p.units = 0.0; p.pending = 0.0; p.seen = 0; p.stride = 1; p.on = true;
    return 0;
}
int st_prof_set_stride(int klass, int stride) {
    if (klass < 0 || klass >= ST_K_COUNT || stride <= 0) return ST_EINVAL;
    g_prof[klass].stride = stride;
    return 0;
}
int st_prof_hint_units(int klass, double units) {
    if (klass < 0 || klass >= ST_K_COUNT) return ST_EINVAL;
    if (g_prof[klass].on) g_prof[klass].pending = units;
    return 0;
}
int64_t st_prof_seen(int klass) {
    if (klass < 0 || klass >= ST_K_COUNT) return -1;
    return (int64_t)g_prof[klass].seen;
}
int st_prof_read(int klass, int* launches, double* total_ms, double* total_units) {
    if (klass < 0 || klass >= ST_K_COUNT) return ST_EINVAL;
    StProf& p = g_prof[klass];
    double ms = 0.0;
    for (int i = 0; i < p.n; ++i) {
        float t = 0.f;
        hipEventSynchronize(p.ev[2 * i + 1]);
        if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) == hipSuccess) ms += t;
    }
    if (launches) *launches = p.n;
    if (total_ms) *total_ms = ms;
    if (total_units) *total_units = p.units;
    p.n = 0; p.units = 0.0; p.seen = 0;
    return 0;
}
/* per-launch view of the sampled launches since st_prof_enable / the last read: duration (ms), algorithmic units and the launcher's
 * shape tag (GEMM class: st_prof_tag = form | epilogue flags | M, N, K; 0 for classes that give none).  Synchronises the events, fills
 * up to `max` entries, returns the number of sampled launches in *n_out and resets the class like st_prof_read. */
int st_prof_read_events(int klass, int max, float* ms_out, double* units_out, unsigned long long* tags_out, int* n_out) {
    if (klass < 0 || klass >= ST_K_COUNT || max < 0 || !n_out) return ST_EINVAL;
    StProf& p = g_prof[klass];
    for (int i = 0; i < p.n; ++i) {
        float t = 0.f;
        hipEventSynchronize(p.ev[2 * i + 1]);
        if (hipEventElapsedTime(&t, p.ev[2 * i], p.ev[2 * i + 1]) != hipSuccess) t = 0.f;
        if (i < max) {
            if (ms_out) ms_out[i] = t;
            if (units_out) units_out[i] = p.unit_each[i];
            if (tags_out) tags_out[i] = p.tags[i];
        }
    }
    *n_out = p.n;
    p.n = 0; p.units = 0.0; p.seen = 0;
    return 0;
}
int st_stream_create_cu_range(int first_cu, int n_cus, st_stream_t* stream_out) {
    const int total = st_num_cus();
    if (!stream_out || first_cu < 0 || n_cus <= 0 || first_cu + n_cus > total) return ST_EINVAL;
    uint32_t mask[16] = {0};
    if (total > 512) return ST_EINVAL;
    for (int c = first_cu; c < first_cu + n_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (st_stream_t)s;
    return 0;
}
int st_stream_destroy(st_stream_t stream) {
    if (!stream) return ST_EINVAL;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}
int st_clock_probe(uint64_t* out, int n_blocks, int spin_us, st_stream_t stream) {
    if (!out || n_blocks <= 0 || n_blocks > 64 || spin_us <= 0 || spin_us > 1000) return ST_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, spin_us * 100);
    ST_CHECK_LAUNCH();
    return 0;
}
int st_prof_disable(int klass) {
    if (klass < 0 || klass >= ST_K_COUNT) return ST_EINVAL;
    g_prof[klass].on = false;
    return 0;
}
/* the kernel-choice switches in force (they change speed, never results: a wrong default passes every parity test — tests/test_layout.py
 * pins the defaults in a fresh process): 0 = training GEMM tile (st_gemm_select / ST_GEMM_VARIANT), 1 = non-temporal decode weight stream
 * (ST_DECODE_NT), 2 = decode attention kernel (st_decode_attn_select / ST_DECODE_ATTN), 3 = non-temporal K/V copies in the decode attention (ST_DECODE_ATTN_NT) */
int64_t st_switch_value(int which) {
    extern int g_train_variant, g_decode_nt, g_decode_attn_nt;
    switch (which) {
        case 0: return g_train_variant;
        case 1: return g_decode_nt;
        case 2: return st_decode_attn_selected();
        case 3: return g_decode_attn_nt;
        default: return -1;
    }
}
}
