// Micro-benchmark (development aid): one step of the stage-1 chain -- LayerNorm -> operand split -> Linear 32 -> 32 (three
// f16 products, weights in LDS) -> 2^P GELU -- on a wave's 64 tokens x 32 channels, carried in registers from step to step,
// in the two accumulator layouts:
//   A  v_mfma_f32_16x16x32_f16: 2 row tiles x 4 pixel tiles, a pixel's 32 channels in 4 lanes x 8 registers
//      (what stage1_f16.h does: LayerNorm statistics cross the four lane quarters with three row swaps per pixel tile)
//   B  v_mfma_f32_32x32x16_f16: 1 row tile x 2 pixel tiles, a pixel's 32 channels in 2 lanes x 16 registers
//      (statistics cross the two lane halves: v_permlane32_swap only; half as many finalisations per lane; an accumulator's
//      registers 8s .. 8s+7 are the K-slots of K-step s of the next Linear when the weight fragments are packed to match)
// Both do the same arithmetic per token; results are not compared with anything (timing only, values stay finite).
//   hipcc --offload-arch=gfx950 -O3 -I balf_amd/csrc tools/ubench/chain_shapes.hip -o tools/ubench/chain_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include "split16.h"
using namespace balf;
typedef float f4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ void swap16(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, r[0]); b = __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ void swap32(float &a, float &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, r[0]); b = __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float opaque_copy(float v) { asm("" : "+v"(v)); return v; }

constexpr float kEps = 1e-5f;

// ---- layout A ----
__global__ __launch_bounds__(512, 1) void chain_a(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char w[2 * 2048];     // two row tiles: [hi 64 x 16 B][lo 64 x 16 B]
    for (int i = threadIdx.x; i < 2 * 2048 / 4; i += 512) reinterpret_cast<unsigned *>(w)[i] = 0x2c002c00u + (i & 0xff);   // ~0.06
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f4 x[2][4];
    for (int nt = 0; nt < 2; ++nt) for (int p = 0; p < 4; ++p) for (int r = 0; r < 4; ++r) x[nt][p][r] = 0.01f * (lane + 3 * nt + 5 * p + r) - 0.3f;
    for (int it = 0; it < iters; ++it) {
        HL b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float s = x[0][p][0], ss = x[0][p][0] * x[0][p][0];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = (nt == 0 ? 1 : 0); r < 4; ++r) { s += x[nt][p][r]; ss = fmaf(x[nt][p][r], x[nt][p][r], ss); }
            swap16(s, ss);
            float c = s + ss, c1 = opaque_copy(c);
            swap32(c, c1);
            float d = c + c1, d1 = opaque_copy(d);
            swap16(d, d1);
            const float mean = d * (1.0f / 32), var = fmaf(d1, 1.0f / 32, -mean * mean);
            const float rstd = __builtin_amdgcn_rsqf(max0(var) + kEps), shift = -mean * rstd;
            float y[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { y[r] = fmaf(x[0][p][r], rstd, shift); y[4 + r] = fmaf(x[1][p][r], rstd, shift); }
            b[p] = split8(y);
        }
        HL a[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            a[nt].hi = *reinterpret_cast<const h8 *>(w + nt * 2048 + lane * 16);
            a[nt].lo = *reinterpret_cast<const h8 *>(w + nt * 2048 + 1024 + lane * 16);
        }
        f4 acc[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[nt][p] = f4{0.1f, 0.2f, 0.3f, 0.4f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[nt][p] = mfma16(a[nt].lo, b[p].hi, acc[nt][p]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[nt][p] = mfma16(a[nt].hi, b[p].lo, acc[nt][p]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[nt][p] = mfma16(a[nt].hi, b[p].hi, acc[nt][p]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) x[nt][p][r] = gelu1(acc[nt][p][r]);
    }
    float s = 0;
    for (int nt = 0; nt < 2; ++nt) for (int p = 0; p < 4; ++p) for (int r = 0; r < 4; ++r) s += x[nt][p][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// ---- layout B ----
__global__ __launch_bounds__(512, 1) void chain_b(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char w[2 * 2048];     // two K-steps: [hi 64 x 16 B][lo 64 x 16 B]
    for (int i = threadIdx.x; i < 2 * 2048 / 4; i += 512) reinterpret_cast<unsigned *>(w)[i] = 0x2c002c00u + (i & 0xff);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float x[2][16];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) x[t][r] = 0.01f * (lane + 3 * t + r) - 0.3f;
    for (int it = 0; it < iters; ++it) {
        HL b[2][2];                                    // [pixel tile][K-step]
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s = x[t][0], ss = x[t][0] * x[t][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) { s += x[t][r]; ss = fmaf(x[t][r], x[t][r], ss); }
            swap32(s, ss);                             // [s.lo ss.lo], [s.hi ss.hi]
            float c = s + ss, c1 = opaque_copy(c);     // lower lanes: S, upper lanes: SS
            swap32(c, c1);                             // c = S everywhere, c1 = SS everywhere
            const float mean = c * (1.0f / 32), var = fmaf(c1, 1.0f / 32, -mean * mean);
            const float rstd = __builtin_amdgcn_rsqf(max0(var) + kEps), shift = -mean * rstd;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = fmaf(x[t][8 * ks + j], rstd, shift);
                b[t][ks] = split8(y);
            }
        }
        HL a[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a[ks].hi = *reinterpret_cast<const h8 *>(w + ks * 2048 + lane * 16);
            a[ks].lo = *reinterpret_cast<const h8 *>(w + ks * 2048 + 1024 + lane * 16);
        }
        f16v acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.1f * (r & 3) + 0.1f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].lo, b[t][ks].hi, acc[t], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[t][ks].lo, acc[t], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks].hi, b[t][ks].hi, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[t][r] = gelu1(acc[t][r]);
    }
    float s = 0;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += x[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
void run(const char *name, K kern, float *out) {
    const int iters = 4000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // two waves per SIMD: one step of both waves takes ms / iters
    printf("%s: %.1f ns per step per wave (two waves per SIMD; %.0f cycles per step per SIMD-pair at a nominal 2.4 GHz)\n",
           name, ms * 1e6 / iters / 2, ms * 1e6 / iters * 2.4);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run("A 16x16x32: LN + split + Linear(32->32, 3 products) + GELU on 64 tokens", chain_a, out);
        run("B 32x32x16: LN + split + Linear(32->32, 3 products) + GELU on 64 tokens", chain_b, out);
    }
    return 0;
}
