// Micro-benchmark (development aid): the stage-1 chain step of chain_shapes.hip (layout B: 32x32x16, two pixel
// tiles per wave) with the two tiles SOFTWARE-PIPELINED inside the wave: tile 1's MFMAs are issued between tile 0's GELU /
// LayerNorm / split instructions and vice versa (the tiles run half a step apart), the interleave fixed at compile time with
// __builtin_amdgcn_sched_group_barrier (one MFMA, then NV vector instructions).  The question: how much of the matrix time hides
// under the vector work when both sit in ONE wave's stream (two waves of different phases on a SIMD do not overlap:
// tools/ubench/mfma_shapes.hip, "split roles").
//   hipcc --offload-arch=gfx950 -O3 -I balf_amd/csrc tools/ubench/chain_pipe.hip -o tools/ubench/chain_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "split16.h"
using namespace balf;
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float max0(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}
__device__ __forceinline__ float gelu1(float x) {
    constexpr float kG0 = -1.000037633e+00f, kG1 = -1.150787766e+00f, kG2 = -4.599926517e-01f, kG3 = -5.182716455e-02f,
                    kG4 = 7.084460191e-03f, kG5 = -4.732939498e-04f;
    const float ax = fabsf(x);
    float p = fmaf(kG5, ax, kG4);
    p = fmaf(p, ax, kG3); p = fmaf(p, ax, kG2); p = fmaf(p, ax, kG1); p = fmaf(p, ax, kG0);
    float e = __builtin_amdgcn_exp2f(p);
    asm("" : "+v"(e));
    return fmaf(-ax, e, max0(x));
}
__device__ __forceinline__ HL split8(const float (&v)[8]) {
    HL r;
    h2 h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
    r.hi = h8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
    r.lo = h8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
    return r;
}
__device__ __forceinline__ void swap32(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, r[0]); b = __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float opaque_copy(float v) { asm("" : "+v"(v)); return v; }
constexpr float kEps = 1e-5f;

__device__ __forceinline__ void ln_split(const float (&x)[16], HL (&b)[2]) {
    float s = x[0], ss = x[0] * x[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) { s += x[r]; ss = fmaf(x[r], x[r], ss); }
    swap32(s, ss);
    float c = s + ss, c1 = opaque_copy(c);
    swap32(c, c1);
    const float mean = c * (1.0f / 32), var = fmaf(c1, 1.0f / 32, -mean * mean);
    const float rstd = __builtin_amdgcn_rsqf(max0(var) + kEps), shift = -mean * rstd;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fmaf(x[8 * ks + j], rstd, shift);
        b[ks] = split8(y);
    }
}
__device__ __forceinline__ f16v mfma6(const HL (&a)[2], const HL (&b)[2]) {
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.1f * (r & 3) + 0.1f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].lo, b[ks].hi, acc, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[ks].lo, acc, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[ks].hi, acc, 0, 0, 0);
    return acc;
}

// MODE 0: serial (chain_shapes' layout B: both tiles LN+split, 12 MFMAs, both tiles GELU)
// MODE 1: tiles half a step apart, the compiler's own schedule inside each half step
// MODE 2: the same with the interleave pinned: 1 MFMA, NV vector instructions, six times, then the rest
template <int MODE, int NV>
__global__ __launch_bounds__(512, 1) void chain(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char w[2 * 2048];
    for (int i = threadIdx.x; i < 2 * 2048 / 4; i += 512) reinterpret_cast<unsigned *>(w)[i] = 0x2c002c00u + (i & 0xff);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float x[2][16];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) x[t][r] = 0.01f * (lane + 3 * t + r) - 0.3f;
    auto wfrag = [&](HL (&a)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a[ks].hi = *reinterpret_cast<const h8 *>(w + ks * 2048 + lane * 16);
            a[ks].lo = *reinterpret_cast<const h8 *>(w + ks * 2048 + 1024 + lane * 16);
        }
    };
    if constexpr (MODE == 3) {               // the vector work alone: the accumulators come from the fragments' bits
        for (int it = 0; it < iters; ++it) {
            HL b[2][2];
            ln_split(x[0], b[0]);
            ln_split(x[1], b[1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[0][r] = gelu1((float)b[0][r >> 3].hi[r & 7] + (float)b[0][r >> 3].lo[r & 7]);
                x[1][r] = gelu1((float)b[1][r >> 3].hi[r & 7] + (float)b[1][r >> 3].lo[r & 7]);
            }
        }
    } else if constexpr (MODE == 4) {        // the matrix work alone
        HL b[2][2], a[2];
        ln_split(x[0], b[0]);
        ln_split(x[1], b[1]);
        f16v acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = x[0][r]; acc1[r] = x[1][r]; }
        for (int it = 0; it < iters; ++it) {
            wfrag(a);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].lo, b[0][ks].hi, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].lo, b[1][ks].hi, acc1, 0, 0, 0); }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[0][ks].lo, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[1][ks].lo, acc1, 0, 0, 0); }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[0][ks].hi, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[1][ks].hi, acc1, 0, 0, 0); }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= 0.001f; acc1[r] *= 0.001f; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[0][r] = acc0[r]; x[1][r] = acc1[r]; }
    } else if constexpr (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            HL b[2][2], a[2];
            ln_split(x[0], b[0]);
            ln_split(x[1], b[1]);
            wfrag(a);
            const f16v acc0 = mfma6(a, b[0]), acc1 = mfma6(a, b[1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) { x[0][r] = gelu1(acc0[r]); x[1][r] = gelu1(acc1[r]); }
        }
    } else {
        HL b0[2], b1[2];
        ln_split(x[0], b0);
        f16v acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = x[1][r];
        auto pin = [] {
            if constexpr (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
                }
            }
        };
        for (int it = 0; it < iters; ++it) {
            HL a[2];
            // half step X: tile 0's MFMAs beside tile 1's GELU + LayerNorm + split
            wfrag(a);
            const f16v acc0 = mfma6(a, b0);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[1][r] = gelu1(acc1[r]);
            ln_split(x[1], b1);
            pin();
            __builtin_amdgcn_sched_barrier(0);
            // half step Y: tile 1's MFMAs beside tile 0's GELU + LayerNorm + split
            wfrag(a);
            acc1 = mfma6(a, b1);
#pragma unroll
            for (int r = 0; r < 16; ++r) x[0][r] = gelu1(acc0[r]);
            ln_split(x[0], b0);
            pin();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) x[1][r] = acc1[r];
    }
    float s = 0;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += x[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_threads = 512;
template <typename K>
void run(const char *name, K kern, float *out) {
    const int iters = 4000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(g_threads), 0, 0, out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(g_threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.1f ns per iteration (%d waves per SIMD)\n", name, ms * 1e6 / iters, g_threads / 256);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 4; ++rep) {
        g_threads = (rep & 1) ? 256 : 512;
        run("vector work alone", chain<3, 0>, out);
        run("matrix work alone (12 MFMA + 32 multiplies)", chain<4, 0>, out);
        run("serial (LN+split x2, 12 MFMA, GELU x2)", chain<0, 0>, out);
        run("tiles half a step apart, compiler's schedule", chain<1, 0>, out);
        run("pinned: 1 MFMA + 3 vector instructions x6", chain<2, 3>, out);
        run("pinned: 1 MFMA + 5 vector instructions x6", chain<2, 5>, out);
        run("pinned: 1 MFMA + 8 vector instructions x6", chain<2, 8>, out);
        run("pinned: 1 MFMA + 16 vector instructions x6", chain<2, 16>, out);
    }
    return 0;
}
