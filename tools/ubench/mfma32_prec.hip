// Numerical check of the split-f16 three-product scheme on v_mfma_f32_32x32x16_f16 against 16x16x32 and fp64 (do both
// shapes treat small / subnormal f16 operands alike?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ void split(float v, _Float16 &h, _Float16 &l) { h = (_Float16)v; l = (_Float16)(v - (float)h); }
__global__ void k32(const float *A, const float *B, float *D) {       // A [32][16], B [16][32]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) { _Float16 x, y; split(A[i * 16 + 8 * h + j], x, y); ah[j] = x; al[j] = y; split(B[(8 * h + j) * 32 + i], x, y); bh[j] = x; bl[j] = y; }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = c[r];
}
__global__ void k16(const float *A, const float *B, float *D) {       // A [16][32] (rows 0..15 of a 32 x 16? no: own shapes) -> D [16][16]
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    h8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) { _Float16 x, y; split(A[i * 32 + 8 * q + j], x, y); ah[j] = x; al[j] = y; split(B[(8 * q + j) * 16 + i], x, y); bh[j] = x; bl[j] = y; }
    f4 c = {};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}
int main() {
    float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dD, sizeof(hD));
    for (float scale : {1.0f, 0.05f, 0.002f}) {
        srand(2);
        for (auto &v : hA) v = scale * (float)(rand() % 20001 - 10000) / 10000.0f;
        for (auto &v : hB) v = (float)(rand() % 20001 - 10000) / 10000.0f;
        (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        double e32 = 0, mag = 0;
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double r = 0; for (int k = 0; k < 16; ++k) r += (double)hA[m * 16 + k] * hB[k * 32 + n]; e32 = fmax(e32, fabs(r - hD[m * 32 + n])); mag = fmax(mag, fabs(r)); }
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        double e16 = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { double r = 0; for (int k = 0; k < 32; ++k) r += (double)hA[m * 32 + k] * hB[k * 16 + n]; e16 = fmax(e16, fabs(r - hD[m * 16 + n])); }
        printf("scale %g: |D| ~ %.3g, max abs error 32x32x16: %.3e   16x16x32: %.3e\n", scale, mag, e32, e16);
    }
    return 0;
}
