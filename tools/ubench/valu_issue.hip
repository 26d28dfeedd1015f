// Micro-benchmark (development aid): VALU issue rate per SIMD as a function of resident waves per SIMD, for plain,
// packed and transcendental fp32 instructions, alone and beside f16 MFMAs.  Answers: does a second / fourth wave on
// a SIMD raise vector throughput (2 cycles per wave64 instruction) or is the pipe 4 cycles per instruction?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4x __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    float a[16];
    f2 pk[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) pk[i] = f2{a[2 * i], a[2 * i + 1]};
    f4x acc[4] = {};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.01f); hb[i] = (_Float16)(i * 0.1f); }
    const float m = 0.999f, c = 0.001f;
    const f2 m2 = {m, m}, c2 = {c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {           // 32 independent v_fma_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == 1) {    // 32 v_pk_fma_f32 (64 fmas)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i]) : "v"(m2), "v"(c2));
        } else if (MODE == 2) {    // 32 v_exp_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == 3) {    // 4 MFMA 16x16x32 f16 + 32 v_fma_f32 interleaved (8 per MFMA)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(8 * j + i) & 15]) : "v"(m), "v"(c));
            }
        } else if (MODE == 4) {    // 4 MFMA only
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j], 0, 0, 0);
        } else if (MODE == 5) {    // 4 MFMA + 16 v_pk_fma_f32 (32 fmas) interleaved
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[(4 * j + i) & 7]) : "v"(m2), "v"(c2));
            }
        } else if (MODE == 6) {    // 32 dependent-free v_fma_f32 in 4 chains of dependent ops (latency)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter_valu, int per_iter_mfma) {
    const int iters = 4000;
    float *out; unsigned long long *cyc;
    const int maxb = 256 * 8;
    hipMalloc(&out, maxb * 256 * 4); hipMalloc(&cyc, maxb * 4 * 8);
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        // s_memtime ticks at 100 MHz on this part? report both raw and wall-derived
        const double wall_cycles = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD %d: %.3f ms, memtime/iter %.1f, wall-cycles(2.4GHz)/iter %.1f -> per-SIMD cycles per VALU instr %.2f (per MFMA %.2f)\n",
               name, w, ms, avg / iters, wall_cycles / iters,
               per_iter_valu ? wall_cycles / iters / (per_iter_valu * w) : 0.0,
               per_iter_mfma ? wall_cycles / iters / (per_iter_mfma * w) : 0.0);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32 x32", 32, 0);
    run<1>("v_pk_fma_f32 x32", 32, 0);
    run<2>("v_exp_f32 x32", 32, 0);
    run<6>("v_fma_f32 dep chains x32", 32, 0);
    run<4>("mfma16x16x32 x4", 0, 4);
    run<3>("mfma x4 + v_fma x32", 32, 4);
    run<5>("mfma x4 + v_pk_fma x16", 16, 4);
    return 0;
}
