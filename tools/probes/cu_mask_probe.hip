// CU-masked streams on MI355X (gfx950): (1) which CUs of which XCD does bit i of hipExtStreamCreateWithCUMask select, (2) do two kernels on
// DISJOINT masks run side by side — a weight-streaming (HBM-bound) kernel on a few CUs next to an MFMA-bound kernel on the rest — and at what
// rates.  Decides whether the decode tail of the rollout can overlap the old / reference log-prob passes.
//   hipcc --offload-arch=gfx950 -O3 cu_mask_probe.hip -o cu_mask_probe && ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

// HBM-bound: stream `n16` uint4 through the CUs (grid-stride), xor-reduce so the loads are live
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// MFMA-bound: every wave runs `iters` x 16 independent 16x16x32 bf16 MFMAs (registers only)
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345f) out[0] = s;
}

static hipStream_t masked_stream(const std::vector<int>& cus) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c : cus) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    return s;
}

static void report(const char* name, const std::vector<int>& cus, unsigned* d, unsigned* h) {
    hipStream_t s = masked_stream(cus);
    const int blocks = 2048;
    CK(hipMemsetAsync(d, 0xff, blocks * 8, s));
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, s, d, 40000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> per_xcc[16];
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 15;
        per_xcc[xcc].insert(((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15));     // (se, sh, cu)
    }
    printf("%-34s bits set %3zu -> CUs used per XCC:", name, cus.size());
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", per_xcc[x].size()); tot += (int)per_xcc[x].size(); }
    printf("  (total %d)\n", tot);
    CK(hipStreamDestroy(s));
}

int main() {
    unsigned *d, *h = (unsigned*)malloc(2048 * 8);
    CK(hipMalloc(&d, 2048 * 8));
    std::vector<int> all, first32, first64, mod8_0, mod8_01, last224, rest_first64;
    for (int i = 0; i < 256; ++i) {
        all.push_back(i);
        if (i < 32) first32.push_back(i);
        if (i < 64) first64.push_back(i); else rest_first64.push_back(i);
        if (i % 8 == 0) mod8_0.push_back(i);
        if (i % 8 < 2) mod8_01.push_back(i);
        if (i >= 32) last224.push_back(i);
    }
    report("all 256 bits", all, d, h);
    report("bits 0..31", first32, d, h);
    report("bits 0..63", first64, d, h);
    report("bits i % 8 == 0", mod8_0, d, h);
    report("bits i % 8 < 2", mod8_01, d, h);
    report("bits 32..255", last224, d, h);

    {   // does a hipGraph LAUNCHED into a CU-masked stream keep the mask?  (the decode iteration is a replayed graph)
        hipStream_t cap; CK(hipStreamCreate(&cap));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, cap, d, 40000);
        CK(hipStreamEndCapture(cap, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipStream_t ms = masked_stream(first64);
        CK(hipMemsetAsync(d, 0xff, 2048 * 8, ms));
        CK(hipGraphLaunch(ge, ms));
        CK(hipStreamSynchronize(ms));
        CK(hipMemcpy(h, d, 2048 * 8, hipMemcpyDeviceToHost));
        std::set<unsigned> cus;
        for (int b = 0; b < 2048; ++b) cus.insert((h[b * 2 + 1] & 15) * 1024 + ((h[b * 2] >> 13) & 7) * 64 + ((h[b * 2] >> 12) & 1) * 16 + ((h[b * 2] >> 8) & 15));
        printf("graph captured on a plain stream, launched into the bits-0..63 stream: %zu distinct CUs used\n", cus.size());
        // and captured ON the masked stream itself
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(ms, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, ms, d, 40000);
        CK(hipStreamEndCapture(ms, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge2, ms));
        CK(hipStreamSynchronize(ms));
        CK(hipMemcpy(h, d, 2048 * 8, hipMemcpyDeviceToHost));
        cus.clear();
        for (int b = 0; b < 2048; ++b) cus.insert((h[b * 2 + 1] & 15) * 1024 + ((h[b * 2] >> 13) & 7) * 64 + ((h[b * 2] >> 12) & 1) * 16 + ((h[b * 2] >> 8) & 15));
        printf("graph captured ON the masked stream and launched into it: %zu distinct CUs used\n", cus.size());
        hipStream_t plain; CK(hipStreamCreate(&plain));
        CK(hipGraphLaunch(ge2, plain));
        CK(hipStreamSynchronize(plain));
        CK(hipMemcpy(h, d, 2048 * 8, hipMemcpyDeviceToHost));
        cus.clear();
        for (int b = 0; b < 2048; ++b) cus.insert((h[b * 2 + 1] & 15) * 1024 + ((h[b * 2] >> 13) & 7) * 64 + ((h[b * 2] >> 12) & 1) * 16 + ((h[b * 2] >> 8) & 15));
        printf("the same graph launched into a plain stream: %zu distinct CUs used\n", cus.size());
    }
    // ---- concurrency: stream kernel on a small mask, MFMA kernel on the complement
    const size_t bytes = (size_t)6 << 30;
    uint4* src; CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
    float* fo; CK(hipMalloc(&fo, 64));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    struct Split { const char* name; std::vector<int> small, big; };
    std::vector<Split> splits;
    splits.push_back({"decode = bits 0..63 | rest", first64, rest_first64});
    { Split s; s.name = "decode = bits i%8<2 (64) | rest"; for (int i = 0; i < 256; ++i) (i % 8 < 2 ? s.small : s.big).push_back(i); splits.push_back(s); }
    { Split s; s.name = "decode = bits 0..31 | rest"; for (int i = 0; i < 256; ++i) (i < 32 ? s.small : s.big).push_back(i); splits.push_back(s); }
    { Split s; s.name = "decode = bits i%8==0 (32) | rest"; for (int i = 0; i < 256; ++i) (i % 8 == 0 ? s.small : s.big).push_back(i); splits.push_back(s); }
    { Split s; s.name = "decode = bits 0..127 | rest"; for (int i = 0; i < 256; ++i) (i < 128 ? s.small : s.big).push_back(i); splits.push_back(s); }
    { Split s; s.name = "decode = bits 0..95 | rest"; for (int i = 0; i < 256; ++i) (i < 96 ? s.small : s.big).push_back(i); splits.push_back(s); }
    const int mf_iters = 60000, mf_blocks = 256 * 8;
    auto run_stream = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, s, src, bytes / 16, (unsigned*)d); };
    auto run_mfma = [&](hipStream_t s) { hipLaunchKernelGGL(mfma_kernel, dim3(mf_blocks), dim3(256), 0, s, fo, mf_iters); };
    const double mf_flops = (double)mf_blocks * 4 * mf_iters * 16 * 2.0 * 16 * 16 * 32;
    {   // baselines on unmasked streams
        hipStream_t s; CK(hipStreamCreate(&s));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, s)); run_stream(s, 2048); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("alone, all CUs: stream %.2f ms = %.2f TB/s", ms, bytes / ms * 1e-9);
            CK(hipEventRecord(e0, s)); run_mfma(s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf(" | mfma %.2f ms = %.0f TF/s\n", ms, mf_flops / ms * 1e-9);
        }
        // two UNMASKED streams together
        hipStream_t s2; CK(hipStreamCreate(&s2));
        CK(hipEventRecord(e0, s)); run_mfma(s); CK(hipEventRecord(e1, s));
        CK(hipEventRecord(f0, s2)); run_stream(s2, 2048); CK(hipEventRecord(f1, s2));
        CK(hipDeviceSynchronize());
        float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, f0, f1));
        printf("two unmasked streams together: mfma %.2f ms (%.0f TF/s), stream %.2f ms (%.2f TB/s)\n", a, mf_flops / a * 1e-9, b, bytes / b * 1e-9);
        CK(hipStreamDestroy(s)); CK(hipStreamDestroy(s2));
    }
    for (auto& sp : splits) {
        hipStream_t ss = masked_stream(sp.small), sb = masked_stream(sp.big);
        float a1, b1, a2, b2;
        const int sblocks = (int)sp.small.size() * 8;
        CK(hipEventRecord(f0, ss)); run_stream(ss, sblocks); CK(hipEventRecord(f1, ss)); CK(hipStreamSynchronize(ss)); CK(hipEventElapsedTime(&b1, f0, f1));
        CK(hipEventRecord(e0, sb)); run_mfma(sb); CK(hipEventRecord(e1, sb)); CK(hipStreamSynchronize(sb)); CK(hipEventElapsedTime(&a1, e0, e1));
        CK(hipEventRecord(e0, sb)); run_mfma(sb); CK(hipEventRecord(e1, sb));
        CK(hipEventRecord(f0, ss)); run_stream(ss, sblocks); CK(hipEventRecord(f1, ss));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&a2, e0, e1)); CK(hipEventElapsedTime(&b2, f0, f1));
        printf("%-36s alone: stream %.2f TB/s, mfma %.0f TF/s | together: stream %.2f TB/s (%.2f ms), mfma %.0f TF/s (%.2f ms)\n", sp.name,
               bytes / b1 * 1e-9, mf_flops / a1 * 1e-9, bytes / b2 * 1e-9, b2, mf_flops / a2 * 1e-9, a2);
        CK(hipStreamDestroy(ss)); CK(hipStreamDestroy(sb));
    }
    {   // only the MFMA side masked (bits 64..255); the streaming kernel on a PLAIN stream may use any CU that is free
        hipStream_t sb = masked_stream(rest_first64), ss; CK(hipStreamCreate(&ss));
        float a2, b2;
        for (int blocks : {512, 2048}) {
            CK(hipEventRecord(e0, sb)); run_mfma(sb); CK(hipEventRecord(e1, sb));
            CK(hipEventRecord(f0, ss)); run_stream(ss, blocks); CK(hipEventRecord(f1, ss));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&a2, e0, e1)); CK(hipEventElapsedTime(&b2, f0, f1));
            printf("mfma masked to bits 64..255, stream kernel UNMASKED (%d blocks): stream %.2f TB/s (%.2f ms), mfma %.0f TF/s (%.2f ms)\n", blocks,
                   bytes / b2 * 1e-9, b2, mf_flops / a2 * 1e-9, a2);
        }
    }
    return 0;
}
