// Stage 1 (C = 32, Cin = 3) of the EXACT-fp32 detector forward: persistent, barrier-free kernels in which ONE WAVE OWNS A
// WHOLE TOKEN GROUP, on v_mfma_f32_32x32x2_f32.  Included by detector.hip inside balf::{anonymous}.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 32 (31 % of the fp32 forward's time before this file).
//
// The generic fp32 kernel (stage_branch_kernel, detector.hip) streams every weight fragment from L2 for every 64 x P pixels,
// parks every Linear's input in LDS and crosses three workgroup barriers per group; its matrix pipe is busy 0.52-0.57 of the
// time at C = 32 (profiles/r5_pmc.json) and deeper weight prefetch does not move it (profiles/r5_f32.txt).  This file is the
// structure of the split-f16 stage-1 kernels (stage1_f16.h) in plain fp32.  What it can gain is bounded: the fp32 MFMAs do NOT
// run beside vector work -- tools/ubench/mfma_valu_mix_f32: 8 x v_mfma_f32_32x32x2_f32 562 cycles, 64 x v_fma_f32 185, both 700,
// interleaved in one wave or from two waves alike -- so a group costs its 324 / 196 MFMAs x 64 cycles PLUS its ~1900 / ~1300
// vector instructions (measured with the memory traffic ablated: exactly that sum), where the f16 MFMAs hide a part:
//   * lane (n = lane & 31, h = lane >> 5); pixel tile p in {0, 1}; token t = 2 n + p (ty = n >> 2, tx = 2 (n & 3) + p);
//     register i of a tile's accumulator holds channel 8 (i >> 2) + 4 h + (i & 3) of the lane's pixel;
//   * "pixel on the lane": D[out_ch][pixel] = W . act^T, so accumulator register i IS the B operand of K-step i (K = 2: lane
//     half 0 carries channel 8 (i >> 2) + (i & 3), lane half 1 the same + 4) of the next Linear -- no conversion, no LDS;
//   * all weights of a branch resident in LDS as A operands in that K order (28.5 / 48.5 KB), staged once per persistent
//     workgroup (one per CU, 8 waves = 2 per SIMD: one wave's LayerNorm / GELU under the other's MFMAs);
//   * the 64 x 64 token mix through a wave-private XOR-swizzled transposed tile (8 KB), no s_barrier in the loop;
//   * LayerNorm statistics cross ONE lane pair (v_permlane32_swap).
// Tensors in HBM are the generic kernel's (U, T, R as NHWC fp32, one row of channel sums per group), so the SE, pool and
// stage-2 kernels are unchanged.  Same arithmetic per output as the generic kernel (bias first, K ascending, two-pass
// LayerNorm, the 2^P GELU); the K order inside a Linear and the order of the channel sums differ, so results agree to
// rounding (1e-7 relative), not bit for bit.
#pragma once

typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma32f(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

constexpr int kF1C = 32;
constexpr int kF1NW = 8;                                           // waves per workgroup (2 per SIMD)
// LDS image, bytes.  A 32 x 32 Linear is 4 KB: float4[(g * 64 + lane)] = W[row m][8 g + 4 h + (0..3)], g = 0..3
constexpr int kF1Conv0 = 0;                                        // 64 lanes x 2 floats
constexpr int kF1Q1 = 512;                                         // this branch's half of RSHMAG.dense1
constexpr int kF1D1 = kF1Q1 + 4096;                                // a half, b half
constexpr int kF1Mix = kF1D1 + 8192;                               // B operands of the token mix: float4[(pt * 8 + j) * 64 + lane]
constexpr int kF1D2 = kF1Mix + 16384;
constexpr int kF1Q2 = kF1D2 + 4096;                                // block only: u' half, v' half
constexpr int kF1R1 = kF1Q2 + 8192;
constexpr int kF1R2 = kF1R1 + 4096;
template <int MODE> constexpr int f1_weight_bytes() { return MODE == 0 ? kF1Q2 : kF1R2 + 4096; }
enum F1Par { kF1pConv0B = 0, kF1pQ1B = 32, kF1pD1B = 64, kF1pGlnG = 128, kF1pGlnB = 160, kF1pMixB1 = 192, kF1pD2B = 256,
             kF1pQ2B = 288, kF1pR1B = 320, kF1pR2B = 352, kF1ParFloats = 384 };
constexpr int kF1BtBytes = kF1C * 256;                             // transposed token tile of a wave: 32 channel rows x 64 tokens
// ... then the GELU chord table of the split-f16 kernels (layout.h: kGeluLutN intervals over [-6, 6), weights.hip builds it; chord
// error 7.6e-7 against 8.6e-7 for the 2^P polynomial of the generic kernel), then the token tiles.  fp32 MFMAs and vector
// instructions share one pipe (see above), so a vector instruction saved is time saved: 4 + 1 LDS read per value instead of 8.
constexpr int kF1LutBytes = ((kGeluLutN + 1) * 8 + 15) / 16 * 16;
template <int MODE> constexpr int f1_lut_offset() { return f1_weight_bytes<MODE>() + kF1ParFloats * 4; }
template <int MODE> constexpr int f1_lds_bytes() { return f1_lut_offset<MODE>() + kF1LutBytes + kF1NW * kF1BtBytes; }
static_assert(f1_lut_offset<1>() < 0x10000, "the table's base must fit the 16-bit offset field of ds_read (the index is the address register)");
static_assert(f1_lds_bytes<1>() <= 160 * 1024, "stage-1 fp32 LDS image");

// float index of W[n][k] inside an A-fragment-ordered [N, K] matrix of the blob (layout.h, weights.hip: pack_frags)
__device__ __forceinline__ int f1_frag_index(int n, int k, int K) {
    return ((((n >> 4) * (K >> 4) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) << 2) + (k & 3);
}

// sum over the two lanes l, l ^ 32 (pure VALU row swap).  The builtin mis-folds a swap whose two operands are the same SSA
// value (stage1_f16.h: half_allreduce2), hence the opaque copy of the operand.
__device__ __forceinline__ float f1_pair_sum(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);      // [v.lo v.lo], [v.hi v.hi]
    // (hipcc folds r[1] into r[0] when both results feed one add -- the sum came out as 2 r[0], seen in the assembly; the
    // opaque pass-through keeps them apart, as at the permlane16_swap of stage1_f16.h)
    unsigned w0 = r[0], w1 = r[1];
    asm("" : "+v"(w0), "+v"(w1));
    return __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
}

template <int N>
__device__ __forceinline__ float f1_row_ror_add(float v) {    // + the value of lane (l + N) mod 16 of the same 16-lane row
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, r);
}

// (x - mean) * rstd over the pixel's 32 channels (two-pass, as ln_stats<false> of det_common.h); affine part folded into
// the consuming Linear (weights.hip: fold_ln) unless g / b are given
__device__ __forceinline__ void f1_ln_stats(const f16v &x, float &rstd, float &shift) {
    constexpr float inv_c = 1.0f / 32;
    float s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    s += ((x[8] + x[9]) + (x[10] + x[11])) + ((x[12] + x[13]) + (x[14] + x[15]));
    const float mean = f1_pair_sum(s) * inv_c;
    float v = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float d = x[r] - mean;
        v = fmaf(d, d, v);
    }
    rstd = __builtin_amdgcn_rsqf(f1_pair_sum(v) * inv_c + kLnEps);
    shift = -mean * rstd;
}

__device__ __forceinline__ void f1_ln_plain(const f16v (&x)[2], f16v (&y)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float rstd, shift;
        f1_ln_stats(x[p], rstd, shift);
#pragma unroll
        for (int r = 0; r < 16; ++r) y[p][r] = fmaf(x[p][r], rstd, shift);
    }
}

// acc[p] += W(32 output rows, 32 inputs) . b[p]: A operands from the LDS image at `wl` (already + lane * 16)
__device__ __forceinline__ void f1_linear(f16v (&acc)[2], const unsigned char *wl, const f16v (&b)[2]) {
    f4 a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = *reinterpret_cast<const f4 *>(wl + g * 1024);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[p] = mfma32f(a[g][e], b[p][4 * g + e], acc[p]);
}

// the bias of the lane's 16 channels (8 g + 4 h + r), four LDS reads: the C operand of a Linear's FIRST MFMA of both tiles
// (an MFMA's C and D may be different registers: no v_mov per accumulator register -- 32 vector instructions per Linear)
__device__ __forceinline__ f16v f1_bias_vec(const float *par, int h) {
    f16v t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4 b = *reinterpret_cast<const f4 *>(par + 8 * g + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; ++r) t[4 * g + r] = b[r];
    }
    return t;
}

// acc[p] = bias + W . b[p]
__device__ __forceinline__ void f1_linear_b(f16v (&acc)[2], const unsigned char *wl, const f16v (&b)[2], const float *bias, int h) {
    f4 a[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) a[g] = *reinterpret_cast<const f4 *>(wl + g * 1024);
    const f16v bv = f1_bias_vec(bias, h);
#pragma unroll
    for (int p = 0; p < 2; ++p) acc[p] = mfma32f(a[0][0], b[p][0], bv);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (g + e)
#pragma unroll
                for (int p = 0; p < 2; ++p) acc[p] = mfma32f(a[g][e], b[p][4 * g + e], acc[p]);
}

// exact (erf) GELU from the chord table: y = clamp01(x / 12 + 1/2), t = 8 N y + 1.5 * 2^23 (the rounded 8 N y lands in the low
// mantissa bits), (a, b) = table[bits(t) & 0x7FF8], gelu = a + b x  (stage1_f16.h: gelu_lut_off; the table's LDS offset is a
// constant, it goes into the read's offset field)
#ifndef BALF_F32_GELU_LUT
#define BALF_F32_GELU_LUT 1
#endif
template <int LUT_OFF>
__device__ __forceinline__ void f1_gelu(f16v (&t)[2]) {
    // (raw LDS addresses: the dynamic LDS block of these kernels starts at LDS address 0 -- they have no static __shared__ --, so
    // masked bits + a constant ARE the address and the constant goes into the read's offset field; through a generic pointer hipcc
    // spends a v_add_u32 per read on adding the LDS symbol's zero.  The last fma is asm with the result in x's own register: left
    // to itself hipcc pairs two of them into a v_pk_fma_f32 behind three v_mov_b32 -- both seen in this kernel's first build, both
    // as in stage1_f16.h: gelu_lut_pipe)
    typedef const f2 __attribute__((address_space(3))) *lds_f2_ptr;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = t[p][r];
            if (BALF_F32_GELU_LUT) {
                const float y = __builtin_amdgcn_fmed3f(fmaf(x, 0.5f / kGeluLutL, 0.5f), 0.0f, 1.0f);
                const float u = fmaf(y, 8.0f * kGeluLutN, 12582912.0f);
                const f2 ab = *reinterpret_cast<lds_f2_ptr>((__builtin_bit_cast(unsigned, u) & 0x7FF8u) + (unsigned)LUT_OFF);
                asm("v_fma_f32 %0, %1, %0, %2" : "+v"(x) : "v"(ab[1]), "v"(ab[0]));
                t[p][r] = x;
            } else {
                t[p][r] = gelu1<false>(x);
            }
        }
}

// BALF_F32_DBG = k (diag.h; tests/experiments/f32_s1_debug.py): the grid kernel stores intermediate tensor k into U instead of u'
#define F1_DBG(K, TENSOR)                                                                                       \
    if (BALF_F32_DBG == K && MODE == 0) {                                                                       \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)       \
            *reinterpret_cast<f4 *>(A.U + (pix0 + p_ * pstep) * C + 8 * g_ + 4 * h) =                           \
                f4{TENSOR[p_][4 * g_], TENSOR[p_][4 * g_ + 1], TENSOR[p_][4 * g_ + 2], TENSOR[p_][4 * g_ + 3]};  \
        continue;                                                                                               \
    }

template <int MODE>
__global__ __launch_bounds__(kF1NW * 64, 1) void stage1_kernel32(StageArgs A) {
    constexpr int C = kF1C, P = 2, NW = kF1NW, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *par = reinterpret_cast<float *>(smem_raw + f1_weight_bytes<MODE>());
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    // ---- stage the weights and parameters once per workgroup ----
    {
        // a 32 x 32 block W[r0 + m][k0 + k] of a fragment-ordered [N, K] matrix -> A operands in the register chain's K order
        auto stage_lin = [&](int dst, int src, int K, int r0, int k0) {
            float *d = reinterpret_cast<float *>(smem_raw + dst);
            for (int i = threadIdx.x; i < 1024; i += NTHR) {
                const int e = i & 3, ln = (i >> 2) & 63, g = i >> 8;
                d[i] = blob[src + f1_frag_index(r0 + (ln & 31), k0 + 8 * g + 4 * (ln >> 5) + e, K)];
            }
        };
        stage_lin(kF1Q1, S.q1_w, C, MODE * C, 0);
        stage_lin(kF1D1, Br.d1_w, C, 0, 0);
        stage_lin(kF1D1 + 4096, Br.d1_w, C, C, 0);
        stage_lin(kF1D2, Br.d2_w, C, 0, 0);
        if (MODE == 1) {
            stage_lin(kF1Q2, S.q2_w, 2 * C, 0, 0);
            stage_lin(kF1Q2 + 4096, S.q2_w, 2 * C, 0, C);
            stage_lin(kF1R1, S.r1_w, C, 0, 0);
            stage_lin(kF1R2, S.r2_w, C, 0, 0);
        }
        // token mix, B operands: lane (n, h) of pixel tile pt, K-step 4 j + e: Wmix[t' = 2 n + pt][t = 8 j + 4 h + e]
        {
            float *d = reinterpret_cast<float *>(smem_raw + kF1Mix);
            for (int i = threadIdx.x; i < 4096; i += NTHR) {
                const int e = i & 3, ln = (i >> 2) & 63, j = (i >> 8) & 7, pt = i >> 11;
                d[i] = blob[Br.mix_w + f1_frag_index(2 * (ln & 31) + pt, 8 * j + 4 * (ln >> 5) + e, kTokens)];
            }
        }
        if (BALF_F32_GELU_LUT) {
            const uint4 *src = reinterpret_cast<const uint4 *>(blob + kLayout.gelu_lut);
            for (int i = threadIdx.x; i < kF1LutBytes / 16; i += NTHR)
                reinterpret_cast<uint4 *>(smem_raw + f1_lut_offset<MODE>())[i] = src[i];
        }
        // conv0 [32, 3] as A operands of two MFMAs (K = 3 padded to 4): lane (m, h) holds W[m][h] and W[m][2] (h = 0) / 0 (h = 1)
        for (int i = threadIdx.x; i < 64; i += NTHR) {
            const int m = i & 31, hh = i >> 5;
            *reinterpret_cast<float2 *>(smem_raw + kF1Conv0 + i * 8) =
                make_float2(blob[S.conv0_w + m * 3 + hh], hh == 0 ? blob[S.conv0_w + m * 3 + 2] : 0.0f);
        }
        for (int i = threadIdx.x; i < kF1ParFloats; i += NTHR) {
            float v;
            if (i < kF1pQ1B) v = blob[S.conv0_b + i];
            else if (i < kF1pD1B) v = blob[S.q1_b + MODE * C + (i - kF1pQ1B)];
            else if (i < kF1pGlnG) v = blob[Br.d1_b + (i - kF1pD1B)];
            else if (i < kF1pGlnB) v = blob[Br.gln_g + (i - kF1pGlnG)];
            else if (i < kF1pMixB1) v = blob[Br.gln_b + (i - kF1pGlnB)];
            else if (i < kF1pD2B) v = blob[Br.mix_b + (i - kF1pMixB1)] + 1.0f;
            else if (i < kF1pQ2B) v = blob[Br.d2_b + (i - kF1pD2B)];
            else if (i < kF1pR1B) v = blob[S.q2_b + (i - kF1pQ2B)];
            else if (i < kF1pR2B) v = blob[S.r1_b + (i - kF1pR1B)];
            else v = blob[S.r2_b + (i - kF1pR2B)];
            par[i] = v;
        }
        __syncthreads();                                         // the only barrier of the kernel
    }

    unsigned char *bT = smem_raw + f1_lut_offset<MODE>() + kF1LutBytes + wave * kF1BtBytes;
    const unsigned char *wl = smem_raw + lane * 16;
    const float2 a0 = *reinterpret_cast<const float2 *>(smem_raw + kF1Conv0 + lane * 8);

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    const int total = A.B * per_img;
    // XCD-aware persistent schedule (speed only; as stage1_f16.h): workgroups b and b + 8 share an XCD, every XCD walks a
    // contiguous run of groups, one group per wave per round
    const int nx = (gridDim.x >> 3) * NW;
    const int wx = (blockIdx.x >> 3) * NW + wave, xcd = blockIdx.x & 7;
    const int stride = 8 * nx;
    const int ty = n >> 2, tx0 = 2 * (n & 3);
    const int pstep = (MODE == 0) ? fw : 1;

    // a group's lane geometry and its raw input (conv0's B operands: bx[p] = input channel h, bx[2 + p] = channel 2 of pixel tile p)
    struct Geo { int img, y, x0; long pix0; };
    auto geo = [&](int item) {
        Geo g;
        g.img = item / per_img;
        const int rem = item - g.img * per_img;
        const int gy = rem / fw, gx = rem - gy * fw;
        if (MODE == 0) { g.y = ty * fh + gy; g.x0 = tx0 * fw + gx; }
        else           { g.y = 8 * gy + ty;  g.x0 = 8 * gx + tx0; }
        g.pix0 = ((long)g.img * H + g.y) * W + g.x0;
        return g;
    };
    auto load_bx = [&](const Geo &g, float (&bx)[4]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float in3[3];
            load_input3(A, blob + kLayout.u8_lut, g.img, g.y, g.x0 + p * pstep, in3);
            bx[p] = h ? in3[1] : in3[0];
            bx[2 + p] = in3[2];
        }
    };
    for (int item = xcd * nx + wx; item < total; item += stride) {
        const Geo gg_ = geo(item);
        const long pix0 = gg_.pix0;

        // ---- x0 = relu(conv0(X)) on the matrix pipe ----
        float bx[4];
        load_bx(gg_, bx);
        // (block) the u' rows of the lane's pixels (NHWC fp32) are requested behind the token mix, where the gate's registers have
        // died, pinned there by a scheduling fence (the compiler otherwise sinks them to their use); requesting the next group's
        // pixels a group ahead as well measured nothing (profiles/r5_f32.txt)
        f16v ub[(MODE == 1) ? 2 : 1];
        auto load_u = [&]() {
            if constexpr (MODE == 1) {
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f4 v = ldg4(A.U + (pix0 + p * pstep) * C + 8 * g + 4 * h);
#pragma unroll
                        for (int r = 0; r < 4; ++r) ub[p][4 * g + r] = v[r];
                    }
            }
        };
        f16v x0v[2];
        const f16v c0b = f1_bias_vec(par + kF1pConv0B, h);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            x0v[p] = mfma32f(a0.x, bx[p], c0b);
            x0v[p] = mfma32f(a0.y, bx[2 + p], x0v[p]);
#pragma unroll
            for (int r = 0; r < 16; ++r) x0v[p][r] = max0(x0v[p][r]);
        }

        F1_DBG(1, x0v)
        // ---- z = GELU(dense1[MODE half](LN(x0))) : u (grid) or v (block) ----
        f16v hb[2];
        f1_ln_plain(x0v, hb);
        F1_DBG(2, hb)
        f16v z[2];
        f1_linear_b(z, wl + kF1Q1, hb, par + kF1pQ1B, h);
        f1_gelu<f1_lut_offset<MODE>()>(z);
        F1_DBG(3, z)

        // ---- gMLP branch on z ----
        f1_ln_plain(z, hb);
        f16v ga[2];
        f1_linear_b(ga, wl + kF1D1, hb, par + kF1pD1B, h);
        f1_gelu<f1_lut_offset<MODE>()>(ga);
        F1_DBG(4, ga)
        {
            f16v gb[2];
            f1_linear_b(gb, wl + kF1D1 + 4096, hb, par + kF1pD1B + C, h);
            f1_gelu<f1_lut_offset<MODE>()>(gb);
            // gating LayerNorm (affine) -> transposed token tile bT[c][t], t = 2 n + p: 8-byte stores, the row's sixteen
            // 16-byte chunks XOR-swizzled by the row so that the 16-byte A-operand reads below spread over the banks
            float rstd[P], shift[P];
#pragma unroll
            for (int p = 0; p < P; ++p) f1_ln_stats(gb[p], rstd[p], shift[p]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 gg = *reinterpret_cast<const f4 *>(par + kF1pGlnG + 8 * g + 4 * h);
                const f4 bb = *reinterpret_cast<const f4 *>(par + kF1pGlnB + 8 * g + 4 * h);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 8 * g + 4 * h + r;
                    float v0 = fmaf(fmaf(gb[0][4 * g + r], rstd[0], shift[0]), gg[r], bb[r]);
                    float v1 = fmaf(fmaf(gb[1][4 * g + r], rstd[1], shift[1]), gg[r], bb[r]);
                    asm("" : "+v"(v0), "+v"(v1));      // (opaque: hipcc otherwise pairs the two tiles' fmas into v_pk_fma_f32 behind three v_mov_b32)
                    *reinterpret_cast<float2 *>(bT + c * 256 + (((n >> 1) ^ (c & 15)) << 4) + (n & 1) * 8) = make_float2(v0, v1);
                }
            }
        }
        {
            // mix^T[c][t'] = sum_t bT[c][t] Wmix[t'][t] (+ bias[t'] + 1 as the accumulator's start value), then the gate
            f4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const f4 *>(bT + n * 256 + (((2 * j + h) ^ (n & 15)) << 4));
            const float2 mb1 = *reinterpret_cast<const float2 *>(par + kF1pMixB1 + 2 * n);
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const float mb = pt ? mb1.y : mb1.x;
                f16v m;
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = mb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f4 w = *reinterpret_cast<const f4 *>(wl + kF1Mix + (pt * 8 + j) * 1024);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m = mfma32f(a[j][e], w[e], m);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ga[pt][r] *= m[r];
            }
        }
        load_u();
        __builtin_amdgcn_sched_barrier(0);
        F1_DBG(5, ga)
        f16v o[2];
        f1_linear_b(o, wl + kF1D2, ga, par + kF1pD2B, h);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[p][r] += z[p][r];

        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f4 *>(A.U + (pix0 + p * pstep) * C + 8 * g + 4 * h) =
                        f4{o[p][4 * g], o[p][4 * g + 1], o[p][4 * g + 2], o[p][4 * g + 3]};
        } else {
            // ---- x1 = dense2(cat[u', v']) + x0 ----
            f16v x1[2];
            f1_linear_b(x1, wl + kF1Q2 + 4096, o, par + kF1pQ2B, h);      // v' half
            f1_linear(x1, wl + kF1Q2, ub);                       // u' half
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x1[p][r] += x0v[p][r];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f4 *>(A.R + (pix0 + p * pstep) * C + 8 * g + 4 * h) =
                        f4{x1[p][4 * g] + x0v[p][4 * g], x1[p][4 * g + 1] + x0v[p][4 * g + 1],
                           x1[p][4 * g + 2] + x0v[p][4 * g + 2], x1[p][4 * g + 3] + x0v[p][4 * g + 3]};
            }
            // ---- t = conv2(lrelu(conv1(LN(x1)))) ----
            f1_ln_plain(x1, hb);
            f16v m1[2];
            f1_linear_b(m1, wl + kF1R1, hb, par + kF1pR1B, h);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) m1[p][r] = __builtin_fmaxf(m1[p][r], 0.2f * m1[p][r]);
            f16v t[2];
            f1_linear_b(t, wl + kF1R2, m1, par + kF1pR2B, h);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f4 *>(A.T + (pix0 + p * pstep) * C + 8 * g + 4 * h) =
                        f4{t[p][4 * g], t[p][4 * g + 1], t[p][4 * g + 2], t[p][4 * g + 3]};
            // channel sums of t over the group's 64 pixels, fixed order: the two tiles, the 16 lanes of a row (DPP), the two
            // rows of the lane half; lanes 0 and 32 store 16 channels each: one row of partial sums per group
            float cs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = t[0][r] + t[1][r];
                s = f1_row_ror_add<1>(f1_row_ror_add<2>(f1_row_ror_add<4>(f1_row_ror_add<8>(s))));
                cs[r] = s + __shfl_xor(s, 16, 64);
            }
            if (n == 0) {
                float *pp = A.partial + (long)item * C + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f4 *>(pp + 8 * g) = f4{cs[4 * g], cs[4 * g + 1], cs[4 * g + 2], cs[4 * g + 3]};
            }
        }
    }
}
