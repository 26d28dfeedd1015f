// Micro-benchmark (development aid): how much vector-ALU work fits beside the matrix pipe, by MFMA shape and by who
// issues it.  DESIGN 4.3c found that a v_mfma_f32_16x16x32_f16 stream leaves the vector ALU of its SIMD only ~1/4 of
// its cycles.  Questions here: (1) is that the same for v_mfma_f32_32x32x16_f16 (half the A/B register reads per MAC)?
// (2) does it matter whether the SAME wave issues both streams or one wave of the SIMD issues only MFMAs and another
// only vector instructions (workgroup of 8 waves: waves 0-3 and 4-7 share SIMDs 0-3)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shapes.hip -o tools/ubench/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4x __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 16x16x32 (NM per iteration), 1: 32x32x16 (NM/2 per iteration: same MACs).  ROLE 0: every wave issues both
// streams (MFMA then NV fmas, interleaved), 1: waves 0-3 only MFMAs, waves 4-7 only fmas (workgroup of 512).
template <int NV, int NM, int SHAPE, int ROLE>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    f4x acc[8];
    f16x big[4];
    for (int j = 0; j < 8; ++j) acc[j] = f4x{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) big[j][i] = 0.f;
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.01f); hb[i] = (_Float16)(i * 0.1f); }
    const float m = 0.999f, c = 0.001f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
    constexpr int NMM = SHAPE == 0 ? NM : NM / 2;
    constexpr int VPER = NMM ? NV / NMM : NV;
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 0) {
#pragma unroll
            for (int j = 0; j < NMM; ++j) {
                if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[j & 3]) : "v"(ha), "v"(hb));
#pragma unroll
                for (int i = 0; i < VPER; ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(j * 4 + i) & 15]) : "v"(m), "v"(c));
            }
            if (NMM == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 15]) : "v"(m), "v"(c));
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int j = 0; j < NMM; ++j) {
                    if (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[j & 3]) : "v"(ha), "v"(hb));
                }
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 15]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += big[j][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int NM, int SHAPE, int ROLE>
void run(float *out) {
    const int iters = 20000;
    // one workgroup of 8 waves per CU: two waves per SIMD
    hipLaunchKernelGGL((k<NV, NM, SHAPE, ROLE>), dim3(256), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NM, SHAPE, ROLE>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e6 / iters * 2.4;             // wall cycles at 2.4 GHz per iteration (both waves of a SIMD)
    // work per SIMD per iteration: ROLE 0 -> 2 waves x (NM x 8192 MACs, NV fmas); ROLE 1 -> 1 wave of each
    const int mult = ROLE == 0 ? 2 : 1;
    printf("%s NV=%3d MFMA-equivalents=%2d %s: %.0f cycles per iteration per SIMD; work = %d x16-cycle MFMA units (%d cyc) + %d fmas "
           "(%.0f cyc at 2.8) -> busy sum / wall = %.2f\n",
           SHAPE ? "32x32x16" : "16x16x32", NV, NM, ROLE ? "split roles " : "same wave   ", cyc, mult * NM, mult * NM * 16, mult * NV,
           mult * NV * 2.8, (mult * NM * 16 + mult * NV * 2.8) / cyc);
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 24, 0, 0>(out);
    run<0, 24, 1, 0>(out);
    run<72, 0, 0, 0>(out);
    run<72, 24, 0, 0>(out);
    run<72, 24, 1, 0>(out);
    run<144, 24, 0, 0>(out);
    run<144, 24, 1, 0>(out);
    run<72, 24, 0, 1>(out);
    run<72, 24, 1, 1>(out);
    run<144, 24, 0, 1>(out);
    run<144, 24, 1, 1>(out);
    run<288, 24, 0, 1>(out);
    run<288, 24, 1, 1>(out);
    return 0;
}
