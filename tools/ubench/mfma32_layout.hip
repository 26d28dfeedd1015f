// Checks the operand / result lane layouts of v_mfma_f32_32x32x16_f16 and v_mfma_f32_32x32x2_f32 assumed by the 32x32
// kernels (stage1_f16.h, stage_cs_f16.h):
//   A[m][k]: lane (m = l & 31, h = l >> 5) holds k = 8 h + j, j = 0..7      (x2_f32: k = h)
//   B[k][n]: lane (n = l & 31, h)          holds k = 8 h + j                (x2_f32: k = h)
//   D[m][n]: lane (n = l & 31, h), register r holds m = 8 (r >> 2) + 4 h + (r & 3)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma32_layout.hip -o tools/ubench/mfma32_layout && tools/ubench/mfma32_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k16(const float *A, const float *B, float *D) {       // A [32][16], B [16][32], D [32][32]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[i * 16 + 8 * h + j]; b[j] = (_Float16)B[(8 * h + j) * 32 + i]; }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = c[r];
}
__global__ void k2(const float *A, const float *B, float *D) {        // A [32][2], B [2][32]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 + h], B[h * 32 + i], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + i] = c[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], *dA, *dB, *dD;
    srand(1);
    for (auto &v : hA) v = (float)(rand() % 9 - 4);
    for (auto &v : hB) v = (float)(rand() % 9 - 4);
    (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dD, sizeof(hD));
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    int bad = 0;
    for (int which = 0; which < 2; ++which) {
        const int K = which ? 2 : 16;
        if (which) hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        (void)hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        int wrong = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                float ref = 0;
                for (int k = 0; k < K; ++k) ref += (which ? hA[m * 2 + k] : hA[m * 16 + k]) * hB[k * 32 + n];
                wrong += ref != hD[m * 32 + n];
            }
        printf("%s: %d of 1024 results differ from the assumed layout\n", which ? "32x32x2_f32" : "32x32x16_f16", wrong);
        bad += wrong;
    }
    return bad != 0;
}
