// Micro-benchmark (development aid): what the matrix and vector pipes cost in POWER.  The forward runs at the package power
// cap (tools/clock_probe.sh: ~1380 W, ~2.0 GHz instead of 2.4), so its speed is set by energy per image, not by issue slots
// alone.  Each mode runs one instruction stream on every SIMD (2 waves per SIMD, as the stage kernels) for a few seconds
// while tools/power_probe.sh samples rocm-smi; the program prints the achieved instruction rate.
//   mode 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_32x32x16_f16   2: v_fma_f32   3: v_exp_f32   4: 1 MFMA(16x16x32) : 6 fma
//   5: v_pk_fma_f32   6: the operand split of a pair
//   (round 4: data movement)  7: ds_read_b128 from LDS (1 KB per wave-instruction)   8: global_load_dwordx4 from an L2-resident
//   1 MB window (L1 thrashed: every CU sweeps 64 KB per iteration)   9: the same from a 4 GB buffer (HBM)
//   (round 5: what an int8 correction product would cost)  10: v_mfma_i32_16x16x64_i8   11: v_mfma_i32_32x32x32_i8 (twice the MACs of
//   the f16 instruction of the same shape in the same cycles)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/power_probe.hip -o tools/ubench/power_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f4x __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void kmem(float *out, const char *buf, unsigned long long span, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f4x acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 7) {
            f4x v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f4x *>(lds + ((j * 8 + wave) * 1024 + lane * 16));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        } else {
            // 8 loads of 1 KB per wave and iteration; the workgroup sweeps 64 KB per iteration, `span` bytes in all
            const unsigned long long base = ((unsigned long long)blockIdx.x * 65536ull * 977ull + (unsigned long long)it * 65536ull) % span;
            f4x v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f4x *>(buf + (base + (j * 8 + wave) * 1024 + lane * 16) % span);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

typedef int i4x __attribute__((ext_vector_type(4)));
typedef int i16x __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void ki8(float *out, int iters) {
    i4x a[2], b[2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int v = 0; v < 2; ++v)
        for (int i = 0; i < 4; ++i) {
            s = s * 1664525u + 1013904223u; a[v][i] = (int)s;
            s = s * 1664525u + 1013904223u; b[v][i] = (int)s;
        }
    i4x acc[8];
    i16x big[4];
    for (int j = 0; j < 8; ++j) acc[j] = i4x{0, 0, 0, 0};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) big[j][i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 10) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[j & 1]), "v"(b[j & 1]));
            if (MODE == 11 && (j & 1) == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(big[j >> 1]) : "v"(a[(j >> 1) & 1]), "v"(b[(j >> 1) & 1]));
        }
    }
    int r = 0;
    for (int j = 0; j < 8; ++j) r += acc[j][0];
    for (int j = 0; j < 4; ++j) r += big[j][0];
    out[blockIdx.x * 512 + threadIdx.x] = (float)r;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i * 0.37f;
    f4x acc[8];
    f16x big[4];
    for (int j = 0; j < 8; ++j) acc[j] = f4x{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) big[j][i] = 0.f;
    h8 ha[2], hb[2];
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int v = 0; v < 2; ++v)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; ha[v][i] = (_Float16)(((int)(s >> 20) - 2048) * 0.0007f);
            s = s * 1664525u + 1013904223u; hb[v][i] = (_Float16)(((int)(s >> 20) - 2048) * 0.0007f);
        }
    const float m = 0.9991f, c = 0.0013f;
    const float pm[2] = {m, m}, pc[2] = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0 || MODE == 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(ha[j & 1]), "v"(hb[j & 1]));
            if (MODE == 1 && (j & 1) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[j >> 1]) : "v"(ha[(j >> 1) & 1]), "v"(hb[(j >> 1) & 1]));
            if (MODE == 2)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == 3)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 5)          // 16 fp32 FMAs as eight packed instructions on register pairs
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double *>(&a[i])) : "v"(*reinterpret_cast<const double *>(&pm[0])), "v"(*reinterpret_cast<const double *>(&pc[0])));
            if (MODE == 6)          // the operand split of one pair: v_cvt_pkrtz + 2 v_fma_mix_f32 + v_cvt_pkrtz, four pairs
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    unsigned hh;
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hh) : "v"(a[i]), "v"(a[i + 1]));
                    float r0 = a[i], r1 = a[i + 1];
                    asm volatile("v_fma_mix_f32 %0, %2, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r0), "+v"(r1) : "v"(hh));
                    unsigned ll;
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(ll) : "v"(r0), "v"(r1));
                    a[i] += __builtin_bit_cast(float, (hh & 0x3fff3fffu)) * 1e-30f; a[i + 1] += __builtin_bit_cast(float, (ll & 0x3fff3fffu)) * 1e-30f;
                }
            if (MODE == 4)
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(j * 6 + i) & 15]) : "v"(m), "v"(c));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int j = 0; j < 8; ++j) r += acc[j][0];
    for (int j = 0; j < 4; ++j) r += big[j][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = mode >= 7 ? 4000 : 20000;
    char *buf = nullptr;
    const unsigned long long span = argc > 3 ? ((unsigned long long)atoll(argv[3]) << 20)      // (round 5) MB: 64 = Infinity-Cache resident
                                             : mode == 9 ? (4ull << 30) : (1ull << 20);
    if (mode >= 8) {
        (void)hipMalloc(&buf, span);
        (void)hipMemset(buf, 1, span);
    }
    auto launch = [&]() {
        switch (mode) {
            case 7: hipLaunchKernelGGL(kmem<7>, dim3(256), dim3(512), 0, 0, out, buf, span, iters); break;
            case 8: hipLaunchKernelGGL(kmem<8>, dim3(256), dim3(512), 0, 0, out, buf, span, iters); break;
            case 9: hipLaunchKernelGGL(kmem<9>, dim3(256), dim3(512), 0, 0, out, buf, span, iters); break;
            case 10: hipLaunchKernelGGL(ki8<10>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 11: hipLaunchKernelGGL(ki8<11>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, iters); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, out, iters); break;
            default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, iters); break;
        }
    };
    launch();
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        for (int i = 0; i < 4; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    // per iteration and wave: 8 MFMA(16x16x32) | 4 MFMA(32x32x16) | 128 fma | 32 exp | 8 MFMA + 48 fma;  2048 waves
    const double per_iter[12] = {8, 4, 128, 32, 8, 64, 32, 8, 8, 8, 8, 4};
    const double rate = n * (double)iters * per_iter[mode] * 2048 / el;
    const char *what[12] = {"MFMA16x16x32", "MFMA32x32x16", "v_fma_f32", "v_exp_f32", "MFMA16x16x32 (+6 fma each)", "v_pk_fma_f32", "operand-split pairs (4 instr + 2 adds each)",
                            "ds_read_b128 (1 KB each)", "global_load_dwordx4 from L2 (1 KB each)", "global_load_dwordx4 from HBM (1 KB each)",
                            "MFMA_i32_16x16x64_i8", "MFMA_i32_32x32x32_i8"};
    printf("mode %d: %.3e %s wave-instructions/s over %.1f s (%.2f per SIMD per us)", mode, rate, what[mode], el, rate / 1024 / 1e6);
    if (mode >= 7 && mode <= 9) printf("  = %.2f TB/s", rate * 1024 / 1e12);
    printf("\n");
    return 0;
}
