// Micro-benchmark (development aid): does a v_mfma_f32_16x16x32_f16 cost the SIMD 16 issue cycles beside saturated VALU
// work, or fewer (the matrix pipe running beside the vector ALU)?  Loop bodies of NV independent v_fma_f32 and NM
// independent MFMAs (inline asm: no compiler reshuffling), 1/2/3/4 waves per SIMD, cycles per iteration per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_mix.hip -o tools/ubench/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4x __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int NV, int NM, int ORDER>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    f4x acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f4x{0.f, 0.f, 0.f, 0.f};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.01f); hb[i] = (_Float16)(i * 0.1f); }
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {            // all MFMAs, then all VALU
#pragma unroll
            for (int j = 0; j < NM; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 15]) : "v"(m), "v"(c));
        } else {                     // interleaved: NV / NM VALU after each MFMA
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
#pragma unroll
                for (int i = 0; i < NV / (NM ? NM : 1); ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(j * 4 + i) & 15]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NM, int ORDER>
void run(float *out) {
    const int iters = 20000;
    for (int w : {1, 2, 3, 4}) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL((k<NV, NM, ORDER>), dim3(blocks), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NM, ORDER>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns_per_iter = ms * 1e6 / iters;        // per SIMD: w waves run concurrently
        printf("NV=%2d NM=%d %s waves/SIMD %d: %.1f ns per iteration of all %d waves -> %.1f ns per wave-iteration "
               "(at 2.0 GHz: %.0f cycles; pure sum VALU*3.2+MFMA*16 = %.0f, VALU*3.2+MFMA*8 = %.0f)\n",
               NV, NM, ORDER ? "interleaved" : "blocked    ", w, ns_per_iter, w, ns_per_iter / w, ns_per_iter / w * 2.0,
               NV * 3.2 + NM * 16.0, NV * 3.2 + NM * 8.0);
    }
}

int main() {
    float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
    run<32, 0, 0>(out);
    run<0, 8, 0>(out);
    run<32, 8, 0>(out);
    run<32, 8, 1>(out);
    run<64, 8, 1>(out);
    run<32, 4, 1>(out);
    // the shape of a Linear + epilogue in the stage kernels: 24 MFMAs and 3 / 6 vector instructions per MFMA
    run<72, 24, 0>(out);
    run<72, 24, 1>(out);
    run<144, 24, 0>(out);
    run<144, 24, 1>(out);
    return 0;
}
