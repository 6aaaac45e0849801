// Which lane's scale covers which operand bytes of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)?  One wave; A has a single 1.0 at
// (lane L, byte J), B is all ones, every scale is 1 except lane L2's A-scale = 2^10: D[row][0] = 1024 iff lane L2's scale
// covers byte (L, J), else 1.   hipcc --offload-arch=gfx950 mx_layout_probe.hip -o mx_layout_probe && ./mx_layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(float* out, int L, int J, int L2, int swap) {
    const int lane = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int i = 0; i < 8; ++i) b[i] = 0x38383838;                 // e4m3 1.0 in every byte
    if (lane == L) a[J >> 2] = 0x38 << (8 * (J & 3));
    const int sa = lane == L2 ? 137 : 127, sb = 127;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (swap) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, sb, 0, sa);   // a as the SECOND operand (the kernel's order)
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); float h[256];
    for (int swap = 0; swap < 2; ++swap)
        for (int L = 0; L < 64; L += 16)
            for (int J = 0; J < 32; J += 8) {
                printf("swap=%d data lane %2d byte %2d: covered by scale of lane(s):", swap, L, J);
                for (int L2 = 0; L2 < 64; ++L2) {
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, L, J, L2, swap);
                    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                    float mx = 0; int at = -1;
                    for (int i = 0; i < 256; ++i) if (h[i] > mx) { mx = h[i]; at = i; }
                    if (mx > 100.f) printf(" %d(max %.0f at lane %d reg %d)", L2, mx, at / 4, at % 4);
                }
                printf("\n");
            }
    return 0;
}
