// Micro-benchmark (development aid): chip-wide rate of global_store_dwordx4 for the access patterns of the stage-1
// epilogue (NHWC fp32 rows of 128 B written from the MFMA accumulator layout), 8 waves per CU, persistent waves.
//   pattern 0: 1 KiB contiguous per wave-instruction
//   pattern 1: 16 pixel rows x 64 B (half a 128-B line each), rows contiguous (two instructions cover 2 KiB)
//   pattern 2: as 1, rows placed like an 8x8 block of a 1920-wide image (8 image rows, 2 x 4 adjacent pixels)
//   pattern 3: 8 pixel rows x 128 B (whole lines), rows placed like pattern 2 (what a lane exchange would give)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float *buf, long n_groups, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    const long gw = (long)blockIdx.x * 8 + wave, nw = (long)gridDim.x * 8;
    f4 v = {1.0f * lane, 2.0f, 3.0f, 4.0f};
    for (long g = gw; g < n_groups; g += nw) {
        // one "group" = 64 pixels x 128 B = 8 KiB written by 8 store instructions
        if (PAT == 0) {
            char *base = (char *)buf + g * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i) *(f4 *)(base + i * 1024 + lane * 16) = v;
        } else if (PAT == 1) {
            char *base = (char *)buf + g * 8192;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) *(f4 *)(base + (p * 16 + li) * 128 + nt * 64 + q * 16) = v;
        } else {
            const long per_row = W / 8;
            const long by = g / per_row, bx = g % per_row;
            if (PAT == 2) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const long pix = (by * 8 + (li >> 1)) * W + bx * 8 + 4 * (li & 1) + p;
                        *(f4 *)((char *)buf + pix * 128 + nt * 64 + q * 16) = v;
                    }
            } else {
                // instruction (p, h): lanes li < 8 / >= 8 -> pixel of lane (li & 7) + 8 h, half (li >> 3): 8 whole rows
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int l2 = (li & 7) + 8 * h;
                        const long pix = (by * 8 + (l2 >> 1)) * W + bx * 8 + 4 * (l2 & 1) + p;
                        *(f4 *)((char *)buf + pix * 128 + (li >> 3) * 64 + q * 16) = v;
                    }
            }
        }
    }
}

template <int PAT>
void run(float *buf, long bytes) {
    const int W = 1920;
    // patterns 2/3 address the buffer as an image of width W: keep whole block rows inside it
    const long block_rows = bytes / ((long)W * 128 * 8);
    const long groups = PAT >= 2 ? block_rows * (W / 8) : bytes / 8192;
    if (groups * 8192 > bytes) { printf("bad size\n"); return; }
    hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, buf, groups, W);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, buf, groups, W);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double wb = groups * 8192.0;
    printf("pattern %d: %.3f ms for %.2f GB -> %.2f TB/s, %.1f cycles(2.1GHz) per store instruction per CU\n", PAT, ms, wb / 1e9,
           wb / 1e9 / ms, ms * 1e-3 * 2.1e9 / (wb / 1024.0 / 256.0));
}

int main() {
    const long bytes = 4L << 30;     // 4 GiB: 8 images x 1088 x 1920 x 128 B = 2.1 GB is the real tensor
    float *buf; hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    run<0>(buf, bytes); run<1>(buf, bytes); run<2>(buf, bytes); run<3>(buf, bytes);
    run<0>(buf, bytes); run<2>(buf, bytes); run<3>(buf, bytes);
    return 0;
}
