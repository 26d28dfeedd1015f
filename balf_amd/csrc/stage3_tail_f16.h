// Tail of stage 3 (C = 128, Cin = 64) of the split-f16 detector forward as a persistent kernel in which every WAVE owns 32
// tokens with all 128 channels and the three weight matrices it needs are resident in LDS.  Included by detector_f16.hip
// inside balf::{anonymous}.
//
// Reference: the end of Down.forward of stage 3 -- x2 = t * s + x1 + x0 with t = conv2(lrelu(conv1(LN(x1)))) (RCAB,
// /root/reference/balf/model/mlp_ma_decoder.py:185-199) and MaxPool2d(2) (:236) -- recomputed from x1 as in the other
// stages' tail kernels.
//
// Why (round 4).  The channel-split tail (stage_cs_kernel16<128, 64, 2>) gave four waves 32 CHANNELS each of a token group:
// every Linear's input is published through LDS behind barriers, and conv0 + conv1 + conv2 -- 160 KB of split-f16 fragments --
// stream from L2 for every group, which is what bounded it (0.35 ms per 8 images at 1088x1920 as it was, 0.23 with the weight
// tiles served from L1 in an ablation build, against 0.22 for its HBM bytes).  A tail has no token mix, so nothing forces the
// waves of a group together: with the TOKENS split every wave is independent -- no LDS exchange, no barrier, no counters --
// and exactly the three matrices fit the CU's LDS (64 + 64 + 32 KB = 160 KB; the bias vectors come from L2).  It stays on
// v_mfma_f32_16x16x32_f16, so x1 (NHWC fp32, as the channel-split block kernel leaves it), the stage input X3 and the output
// X4 (16x16 fragment format) keep their formats and the weights are the blob's tiles, copied verbatim.
//   lane (li = lane & 15, q = lane >> 4); pixel tile p in {0, 1}; the wave's tokens are rows 4 w .. 4 w + 3 of the 8x8 block:
//   ty = 4 w + 2 ((li >> 2) & 1) + (li >> 3), tx = 2 (li & 3) + p -- a lane's two tiles are horizontal neighbours, the rows of a
//   2x2 pooling window sit in lanes li, li ^ 8 (one DPP row_ror:8);
//   accumulator tile t register r holds channel 16 t + 4 q + r; K-step s of the next Linear = tiles 2 s, 2 s + 1 (split8).
#pragma once

constexpr int kT3C = 128, kT3Cin = 64, kT3NT = kT3C / 16, kT3KS = kT3C / 32, kT3KI = kT3Cin / 32;
constexpr int kT3R1 = 0, kT3R2 = kT3NT * kT3KS * 2048, kT3Conv0 = 2 * kT3NT * kT3KS * 2048;
constexpr int kT3LdsBytes = kT3Conv0 + kT3NT * kT3KI * 2048;       // 64 + 64 + 32 KB: the whole LDS of a CU
constexpr int kT3Waves = 8;
static_assert(kT3LdsBytes == 160 * 1024, "the three matrices fill the LDS exactly");

// sums over the four lanes l, l ^ 16, l ^ 32, l ^ 48 of TWO values at once, by row swaps (pure vector ALU).  (Opaque copies:
// the swap builtins mis-fold when both operands are one SSA value, stage1_f16.h.)
__device__ __forceinline__ void quad_allreduce2(float &s, float &ss) {
    auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
    auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
    const auto r0 = __builtin_amdgcn_permlane16_swap(u(s), u(ss), false, false);      // rows [s0 ss0 s2 ss2], [s1 ss1 s3 ss3]
    unsigned a0 = r0[0], a1 = r0[1];
    asm("" : "+v"(a0), "+v"(a1));
    const float c = f(a0) + f(a1);                                                      // [S01 SS01 S23 SS23]
    unsigned c1 = u(c);
    asm("" : "+v"(c1));
    const auto r1 = __builtin_amdgcn_permlane32_swap(u(c), c1, false, false);         // [S01 SS01 S01 SS01], [S23 SS23 S23 SS23]
    unsigned b0 = r1[0], b1 = r1[1];
    asm("" : "+v"(b0), "+v"(b1));
    const float d = f(b0) + f(b1);                                                      // [S SS S SS]
    unsigned d1 = u(d);
    asm("" : "+v"(d1));
    const auto r2 = __builtin_amdgcn_permlane16_swap(u(d), d1, false, false);         // [S S S S], [SS SS SS SS]
    s = f(r2[0]);
    ss = f(r2[1]);
}

// acc[nt][p] += W(row tile nt) . B[.][p] over KS K-steps of 32; weight tile (nt, ks) at wl + (nt * KS + ks) * 2048 (wl already
// + lane * 16), products in the order of the other kernels (low halves first)
template <int KS>
__device__ __forceinline__ void t3_linear(f4 (&acc)[kT3NT][2], const unsigned char *wl, const HL (&b)[KS][2]) {
#pragma unroll
    for (int nt = 0; nt < kT3NT; ++nt) {
        HL a[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a[ks].hi = *reinterpret_cast<const h8 *>(wl + (nt * KS + ks) * 2048);
            a[ks].lo = *reinterpret_cast<const h8 *>(wl + (nt * KS + ks) * 2048 + 1024);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) if (!BALF_DROP_WLO) acc[nt][p] = mfma16(a[ks].lo, b[ks][p].hi, acc[nt][p]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[nt][p] = mfma16(a[ks].hi, b[ks][p].lo, acc[nt][p]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[nt][p] = mfma16(a[ks].hi, b[ks][p].hi, acc[nt][p]);
        // one row tile's fragments at a time: left alone, hipcc gathers the LDS reads of several row tiles in front of the first
        // MFMA (8 x 32 registers for a Linear) and spills 118 registers
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void t3_bias(f4 (&t)[kT3NT][2], const float *bias, int q) {
#pragma unroll
    for (int nt = 0; nt < kT3NT; ++nt) {
        const f4 b = ldg4(bias + 16 * nt + 4 * q);
        t[nt][0] = b;
        t[nt][1] = b;
    }
}

__device__ __forceinline__ void t3_split(const f4 (&x)[kT3NT][2], HL (&b)[kT3KS][2]) {
#pragma unroll
    for (int s = 0; s < kT3KS; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) b[s][p] = split8(x[2 * s][p], x[2 * s + 1][p]);
}

__global__ __launch_bounds__(kT3Waves * 64, 1) void stage3_tail_kernel16(StageArgs A) {
    constexpr int C = kT3C, NW = kT3Waves, NTHR = NW * 64, NT = kT3NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    {   // the three matrices, once per workgroup (the blob's 16x16 fragment tiles, verbatim)
        auto copy = [&](int dst, int src_floats, int bytes) {
            const char *s = reinterpret_cast<const char *>(blob + src_floats);
            for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16)
                *reinterpret_cast<uint4 *>(smem_raw + dst + i) = *reinterpret_cast<const uint4 *>(s + i);
        };
        copy(kT3R1, S.r1_w, NT * kT3KS * 2048);
        copy(kT3R2, S.r2_w, NT * kT3KS * 2048);
        copy(kT3Conv0, S.conv0_w, NT * kT3KI * 2048);
        __syncthreads();                                         // the only barrier of the kernel
    }
    const unsigned char *wl = smem_raw + lane * 16;

    const int H = A.H, W = A.W, fw = W / 8;
    const int per_img = (H / 8) * fw;
    const int total = A.B * per_img * 2;                         // work items: half token groups (32 tokens)
    // XCD-aware persistent schedule, one item per wave and round (stage1_f16.h)
    const int nx = (gridDim.x >> 3) * NW;
    const int wx = (blockIdx.x >> 3) * NW + wave, xcd = blockIdx.x & 7;
    const int ty_l = 2 * ((li >> 2) & 1) + (li >> 3), tx0 = 2 * (li & 3);

    float rng = 0.0f;                                            // max |v| over the stage's output (status block)
    for (int item = xcd * nx + wx; item < total; item += 8 * nx) {
        const int grp = item >> 1, w = item & 1;
        const int n = grp / per_img, rem = grp - n * per_img;
        const int gy = rem / fw, gx = rem - gy * fw;
        const int y = 8 * gy + 4 * w + ty_l, x0p = 8 * gx + tx0;
        const long pix0 = ((long)n * H + y) * W + x0p;           // the lane's two pixels: pix0, pix0 + 1

        // x1 as the block kernel left it (NHWC fp32) and the stage input's fragments
        f4 x1t[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) x1t[nt][p] = *reinterpret_cast<const f4 *>(A.R + (pix0 + p) * C + 16 * nt + 4 * q);
        HL xin[kT3KI][2];
#pragma unroll
        for (int kk = 0; kk < kT3KI; ++kk)
#pragma unroll
            for (int p = 0; p < 2; ++p) xin[kk][p] = load_frag_px(A.X, pix0 + p, kT3Cin, kk, q);

        // x0 = relu(conv0(X)) first (its operands die here), then r = x1 + x0 once LN(x1) has been taken
        f4 x0[NT][2];
        t3_bias(x0, blob + S.conv0_b, q);
        t3_linear<kT3KI>(x0, wl + kT3Conv0, xin);
        HL b[kT3KS][2];
        {   // (x1 - mean) * rstd over the pixel's 128 channels (affine part folded into conv1's weights), split K-step by K-step
            float rstd[2], shift[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float s = x1t[0][p][0], ss = x1t[0][p][0] * x1t[0][p][0];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = (nt == 0 ? 1 : 0); r < 4; ++r) { s += x1t[nt][p][r]; ss = fmaf(x1t[nt][p][r], x1t[nt][p][r], ss); }
                quad_allreduce2(s, ss);
                const float mean = s * (1.0f / C);
                const float var = fmaf(ss, 1.0f / C, -mean * mean);
                rstd[p] = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
                shift[p] = -mean * rstd[p];
            }
#pragma unroll
            for (int s = 0; s < kT3KS; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f4 y0, y1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        y0[r] = fmaf(x1t[2 * s][p][r], rstd[p], shift[p]);
                        y1[r] = fmaf(x1t[2 * s + 1][p][r], rstd[p], shift[p]);
                    }
                    b[s][p] = split8(y0, y1);
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) x1t[nt][p][r] += max0(x0[nt][p][r]);       // r = x1 + relu(conv0)
        f4 m1[NT][2];
        t3_bias(m1, blob + S.r1_b, q);
        t3_linear<kT3KS>(m1, wl + kT3R1, b);
        lrelu(m1);
        t3_split(m1, b);
        f4 t[NT][2];
        t3_bias(t, blob + S.r2_b, q);
        t3_linear<kT3KS>(t, wl + kT3R2, b);
        // v = r + s t, max over the 2x2 window: the lane's two tiles are horizontal neighbours, the rows in lanes li, li ^ 8
        f4 mx[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f4 sc = ldg4(A.scale + (long)n * C + 16 * nt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = fmaf(t[nt][0][r], sc[r], x1t[nt][0][r]);
                const float v1 = fmaf(t[nt][1][r], sc[r], x1t[nt][1][r]);
                const float m = __builtin_fmaxf(v0, v1);
                const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                mx[nt][r] = __builtin_fmaxf(m, o);
            }
        }
        // lanes li and li ^ 8 hold the same pooled pixel: the one with li < 8 stores K-steps 0, 1 (tiles 0-3), its partner 2, 3
        {
            const int sel = li >> 3;
            const unsigned sm = 0u - (unsigned)sel;
            f4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[k][r] = lane_select(sm, mx[4 + k][r], mx[k][r]);
#pragma unroll
            for (int k = 0; k < 4; ++k) rng = range_max(range_max(rng, o[k][0], o[k][1]), o[k][2], o[k][3]);   // about to be split (status block)
            const long opix = ((long)n * (H / 2) + (y >> 1)) * (W / 2) + (x0p >> 1);
            store_frag_px(A.out, opix, C, 2 * sel, q, split8(o[0], o[1]));
            store_frag_px(A.out, opix, C, 2 * sel + 1, q, split8(o[2], o[3]));
        }
    }
    if (rng >= kF16Max) status_raise(A.status, 1 /* BALF_STATUS_RANGE */);
}
