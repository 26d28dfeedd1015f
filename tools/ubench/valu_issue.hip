// Micro-benchmark (development aid): VALU issue rate per SIMD as a function of resident waves per SIMD, for plain,
// packed and transcendental fp32 instructions, alone and beside f16 MFMAs.  Answers: does a second / fourth wave on
// a SIMD raise vector throughput (2 cycles per wave64 instruction) or is the pipe 4 cycles per instruction?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
        } else if (MODE >= 7) {    // 32 independent instructions of another class (which of them run at the fma's rate?)
            unsigned *u = reinterpret_cast<unsigned *>(a);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (MODE == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(7));
                    if (MODE == 8) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                    if (MODE == 9) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                    if (MODE == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
                    if (MODE == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(0x7fffffffu));
                    if (MODE == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    if (MODE == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                    if (MODE == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    if (MODE == 15) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                    if (MODE == 16) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));
                    if (MODE == 17) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                    if (MODE == 18) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                    if (MODE == 19) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                    if (MODE == 20) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                    if (MODE == 21) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
                    if (MODE == 22) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                    if (MODE == 23) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(0x07060302u));
                    if (MODE == 24) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(0xffff0000u), "v"(m));
                    if (MODE == 25) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
                    if (MODE == 26) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(m));
                    if (MODE == 27) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,0]" : "+v"(a[i]) : "v"(m), "v"(c));
                    if (MODE == 28) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                    if (MODE == 29) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
                    if (MODE == 30) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
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
    for (int w : {2, 4}) {
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
    if (getenv("VALU_CLASSES")) {
        run<0>("v_fma_f32 x32", 32, 0);
        run<7>("v_max_i32 x32", 32, 0);
        run<8>("v_cvt_pkrtz_f16_f32 x32", 32, 0);
        run<9>("v_fma_mixlo_f16 x32", 32, 0);
        run<10>("v_mov_b32 x32", 32, 0);
        run<11>("v_and_b32 x32", 32, 0);
        run<12>("v_add_f32 x32", 32, 0);
        run<13>("v_mul_f32 x32", 32, 0);
        run<14>("v_max_f32 x32", 32, 0);
        run<15>("v_add_f32 dpp row_ror x32", 32, 0);
        run<16>("s_nop 1 + v_permlane16_swap x32", 32, 0);
        run<17>("v_cndmask_b32 x32", 32, 0);
        run<18>("v_fmac_f32 x32", 32, 0);
        run<19>("v_fma_f32 |src| x32", 32, 0);
        run<20>("v_cvt_pk_f16_f32 x32", 32, 0);
        run<21>("v_cvt_f16_f32 x32", 32, 0);
        run<22>("v_pack_b32_f16 x32", 32, 0);
        run<23>("v_perm_b32 x32", 32, 0);
        run<24>("v_bfi_b32 x32", 32, 0);
        run<25>("v_cvt_f32_f16 x32", 32, 0);
        run<26>("v_med3_f32 x32", 32, 0);
        run<27>("v_fma_mix_f32 x32", 32, 0);
        run<28>("v_sub_f32 x32", 32, 0);
        run<29>("v_lshlrev_b32 x32", 32, 0);
        run<30>("v_cvt_pk_bf16_f32 x32", 32, 0);
        return 0;
    }
    run<0>("v_fma_f32 x32", 32, 0);
    run<1>("v_pk_fma_f32 x32", 32, 0);
    run<2>("v_exp_f32 x32", 32, 0);
    run<6>("v_fma_f32 dep chains x32", 32, 0);
    run<4>("mfma16x16x32 x4", 0, 4);
    run<3>("mfma x4 + v_fma x32", 32, 4);
    run<5>("mfma x4 + v_pk_fma x16", 16, 4);
    return 0;
}
