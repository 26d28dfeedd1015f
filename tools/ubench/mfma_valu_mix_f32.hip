// Micro-benchmark (development aid): does vector work run in the shadow of the fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles;
// v_mfma_f32_16x16x4_f32, 32) -- inside one wave's instruction stream (interleaved), or from another wave of the SIMD (blocked,
// 2 waves)?  The exact-fp32 kernels measured matrix busy + vector busy = their run time (no co-execution counter ticks);
// this tool says whether that is the hardware or the schedule.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_mix_f32.hip -o tools/ubench/mfma_valu_mix_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4x __attribute__((ext_vector_type(4)));

template <int NV, int NM, int ORDER, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    f16v acc[4];
    f4x acc4[8];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int j = 0; j < 8; ++j) acc4[j] = f4x{0.f, 0.f, 0.f, 0.f};
    float fa = threadIdx.x * 0.01f, fb = 0.5f;
    const float m = 0.999f, c = 0.001f;
    auto mf = [&](int j) {
        if (SHAPE == 32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[j & 7]) : "v"(fa), "v"(fb));
    };
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int j = 0; j < NM; ++j) mf(j);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 15]) : "v"(m), "v"(c));
        } else {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                mf(j);
#pragma unroll
                for (int i = 0; i < NV / (NM ? NM : 1); ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(j * 4 + i) & 15]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][5];
    for (int j = 0; j < 8; ++j) s += acc4[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NM, int ORDER, int SHAPE>
void run(float *out) {
    const int iters = 5000;
    for (int w : {1, 2, 3}) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL((k<NV, NM, ORDER, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NM, ORDER, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / iters / w;
        const int mc = SHAPE == 32 ? 64 : 32;
        printf("%dx%d NV=%3d NM=%2d %s waves/SIMD %d: %.1f ns per wave-iteration = %.0f cycles at 2.38 GHz (MFMA alone %d, VALU alone at 4 cycles %d)\n",
               SHAPE, SHAPE, NV, NM, ORDER ? "interleaved" : "blocked    ", w, ns, ns * 2.38, NM * mc, NV * 4);
    }
}

int main() {
    float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0, 8, 0, 32>(out);
    run<64, 0, 0, 32>(out);
    run<64, 8, 0, 32>(out);      // 8 MFMAs (512 cycles) then 64 fmas (256)
    run<64, 8, 1, 32>(out);      // 8 fmas behind each MFMA
    run<96, 8, 1, 32>(out);      // 12 behind each
    run<0, 16, 0, 16>(out);
    run<64, 16, 0, 16>(out);
    run<64, 16, 1, 16>(out);     // 4 behind each 16x16x4
    return 0;
}
