// Operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, e8m0 block scales) on gfx950 — the K = 64 instruction the
// 4-wave fp8 tile uses (two k-steps per 128-byte operand row, fragments double-buffered like the bf16 tile).
//   part 1: which lane's scale covers which operand byte (a single 1.0 in A, one lane's A-scale = 2^10);
//   part 2: which (lane group, byte) of B meets (lane 0, byte J) of A in the contraction;
//   part 3: a random 32x32x64 product under two packing hypotheses against a host reference:
//     H1: lane (r = lane & 31, g = lane >> 5) holds k = 16g..16g+15 in bytes 0..15 and k = 32+16g..32+16g+15 in bytes 16..31, scale of block g
//     H2: lane holds k = 32g..32g+31, scale of block g
//   hipcc --offload-arch=gfx950 -O2 mx_layout_probe32.hip -o mx_layout_probe32 && ./mx_layout_probe32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void cover(float* out, int L, int J, int L2, int swap) {
    const int lane = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int i = 0; i < 8; ++i) b[i] = 0x38383838;
    if (lane == L) a[J >> 2] = 0x38 << (8 * (J & 3));
    const int sa = lane == L2 ? 137 : 127, sb = 127;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    if (swap) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c, 0, 0, 0, sb, 0, sa);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
__global__ void meet(float* out, int J, int L2, int J2) {
    const int lane = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane == 0) a[J >> 2] = 0x38 << (8 * (J & 3));
    if (lane == L2) b[J2 >> 2] = 0x38 << (8 * (J2 & 3));
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
// opsel: byte `sel` of the scale registers
template <int SEL>
__global__ void product(const i32x8* A, const i32x8* B, const int* sa, const int* sb, float* out) {
    const int lane = threadIdx.x;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], c, 0, 0, SEL, sa[lane], SEL, sb[lane]);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
static float e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 4); float h[1024];
    for (int swap = 0; swap < 2; ++swap)
        for (int L = 0; L < 64; L += 32)
            for (int J = 0; J < 32; J += 8) {
                printf("cover swap=%d data lane %2d byte %2d: scale lane(s):", swap, L, J);
                for (int L2 = 0; L2 < 64; ++L2) {
                    hipLaunchKernelGGL(cover, dim3(1), dim3(64), 0, 0, d, L, J, L2, swap);
                    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                    float mx = 0; int at = -1;
                    for (int i = 0; i < 1024; ++i) if (h[i] > mx) { mx = h[i]; at = i; }
                    if (mx > 100.f) printf(" %d(at lane %d reg %d)", L2, at / 16, at % 16);
                }
                printf("\n");
            }
    for (int J = 0; J < 32; J += 4) {
        printf("meet A(lane 0, byte %2d) <-> B:", J);
        for (int L2 = 0; L2 < 64; L2 += 32)
            for (int J2 = 0; J2 < 32; ++J2) {
                hipLaunchKernelGGL(meet, dim3(1), dim3(64), 0, 0, d, J, L2, J2);
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                float mx = 0;
                for (int i = 0; i < 1024; ++i) if (h[i] > mx) mx = h[i];
                if (mx > 0.5f) printf(" (lane %d, byte %d)", L2, J2);
            }
        printf("\n");
    }
    // part 3
    srand(7);
    static uint8_t Am[32][64], Bm[32][64]; static int SA[32][2], SB[32][2];
    for (int r = 0; r < 32; ++r) {
        for (int k = 0; k < 64; ++k) {
            do { Am[r][k] = rand() & 0xff; } while (((Am[r][k] >> 3) & 15) == 15 && (Am[r][k] & 7) == 7);
            do { Bm[r][k] = rand() & 0xff; } while (((Bm[r][k] >> 3) & 15) == 15 && (Bm[r][k] & 7) == 7);
        }
        for (int b = 0; b < 2; ++b) { SA[r][b] = 120 + rand() % 12; SB[r][b] = 120 + rand() % 12; }
    }
    static double ref[32][32];       // ref[m][n] = sum_k A[m][k] B[n][k] 2^(sa-127) 2^(sb-127)
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)e4m3(Am[m][k]) * ldexp(1.0, SA[m][k >> 5] - 127) * (double)e4m3(Bm[n][k]) * ldexp(1.0, SB[n][k >> 5] - 127);
            ref[m][n] = s;
        }
    i32x8 *dA, *dB; int *dsa, *dsb;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    for (int hyp = 1; hyp <= 2; ++hyp)
        for (int sel = 0; sel < 4; sel += 3) {
            uint8_t pa[64][32], pb[64][32]; int sa[64], sb[64];
            for (int l = 0; l < 64; ++l) {
                const int r = l & 31, g = l >> 5;
                for (int j = 0; j < 32; ++j) {
                    const int k = hyp == 1 ? (j < 16 ? 16 * g + j : 32 + 16 * g + (j - 16)) : 32 * g + j;
                    pa[l][j] = Am[r][k]; pb[l][j] = Bm[r][k];
                }
                sa[l] = (SA[r][g] << (8 * sel)) | (sel ? 0x55 : 0x55000000); sb[l] = (SB[r][g] << (8 * sel)) | (sel ? 0x33 : 0x33000000);
            }
            hipMemcpy(dA, pa, sizeof(pa), hipMemcpyHostToDevice); hipMemcpy(dB, pb, sizeof(pb), hipMemcpyHostToDevice);
            hipMemcpy(dsa, sa, sizeof(sa), hipMemcpyHostToDevice); hipMemcpy(dsb, sb, sizeof(sb), hipMemcpyHostToDevice);
            if (sel == 0) hipLaunchKernelGGL(product<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, d);
            else hipLaunchKernelGGL(product<3>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            // D[i][j]: lane l holds j = l & 31, i = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5); with (A, B) operand order D = A B^T: i = m (A row), j = n
            double e1 = 0, e2 = 0, mag = 0;
            for (int l = 0; l < 64; ++l)
                for (int reg = 0; reg < 16; ++reg) {
                    const int j = l & 31, i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                    e1 = fmax(e1, fabs(h[l * 16 + reg] - ref[i][j]));
                    e2 = fmax(e2, fabs(h[l * 16 + reg] - ref[j][i]));
                    mag = fmax(mag, fabs(ref[i][j]));
                }
            printf("product hypothesis H%d, scale byte %d: max |D - ref| with D[i][j] = A row i . B row j: %.3g; transposed: %.3g (max |ref| %.3g)\n", hyp, sel, e1, e2, mag);
        }
    return 0;
}
