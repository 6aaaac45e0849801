// api.hip — version + profiling hooks of the C ABI.
#include "common.h"

StProf g_prof[ST_K_COUNT];

extern "C" {
int st_version(void) { return 1; }
const char* st_arch(void) { return "gfx950"; }

int st_prof_enable(int klass, int max_events) {
    if (klass < 0 || klass >= ST_K_COUNT || max_events <= 0) return ST_EINVAL;
    StProf& p = g_prof[klass];
    if (p.ev) { for (int i = 0; i < 2 * p.cap; ++i) hipEventDestroy(p.ev[i]); delete[] p.ev; }
    p.ev = new hipEvent_t[2 * max_events];
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
int st_prof_disable(int klass) {
    if (klass < 0 || klass >= ST_K_COUNT) return ST_EINVAL;
    g_prof[klass].on = false;
    return 0;
}
}
