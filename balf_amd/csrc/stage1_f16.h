// Stage 1 (C = 32, Cin = 3) of the split-f16 detector forward: persistent, barrier-free kernels in which ONE WAVE
// OWNS A WHOLE TOKEN GROUP.  Included by detector_f16.hip inside balf::{anonymous}.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 32 (the stage that holds 40 % of the forward).
//
// Why a dedicated kernel.  The generic stage kernel spreads the 64 tokens of a group over the four waves of a
// workgroup (16 each), so the 64x64 token mix needs workgroup barriers, and it parks every B operand in an LDS slot
// so that the K loop can be a runtime loop.  At C = 32 neither is necessary:
//   * a wave's four 16-pixel MFMA tiles ARE the 64 tokens of one 8x8 block (block branch) or of the 64 grid cells at
//     one in-cell offset (grid branch): the token mix becomes wave-local (transpose through a wave-private LDS tile,
//     no s_barrier anywhere in the main loop);
//   * every Linear is a single K-step, so the B fragments go from the accumulator registers straight into the next
//     MFMA (split8), no LDS round trip;
//   * all weights of the stage (36 KB grid / 52 KB block, split-f16 fragments) fit in LDS next to the token tiles:
//     they are staged ONCE per workgroup and the workgroup is persistent (one per CU, waves loop over groups), so
//     no weight ever comes from L2 inside the loop; conv0 (3 -> 32) runs on the matrix pipe too (K padded to 32).
// Token t = 8 ty + tx of a group sits in MFMA column li of pixel tile p with t = 4 li + p: a lane's four tiles are
// four ADJACENT tokens, so the transposed tile is written with 8-byte LDS stores and (block branch) the NCHW input
// is read with one 16-byte load per colour plane.  The mixing matrix is re-ordered to that column order when it is
// staged.  Results are independent of the grid size and of which wave processes which group.
#pragma once

constexpr int kS1C = 32;
constexpr int kS1Pitch = 72;                                   // halves per channel row of the transposed token tile
constexpr int kS1BtBytes = 2 * kS1C * kS1Pitch * 2;            // hi + lo planes per wave: 9216 B
// LDS image: weight tiles (2 KiB each: [hi 64 x 16 B][lo 64 x 16 B]) ...
constexpr int kS1Conv0 = 0;                                    // 2 row tiles (built in the kernel from the plain [32,3] matrix)
constexpr int kS1Q1 = kS1Conv0 + 2 * 2048;                     // 2 row tiles (this branch's half of RSHMAG.dense1)
constexpr int kS1D1 = kS1Q1 + 2 * 2048;                        // 4 row tiles
constexpr int kS1Mix = kS1D1 + 4 * 2048;                       // 4 token tiles x 2 K-steps, columns re-ordered
constexpr int kS1D2 = kS1Mix + 8 * 2048;                       // 2 row tiles
constexpr int kS1Q2 = kS1D2 + 2 * 2048;                        // block only: 2 row tiles x 2 K-steps
constexpr int kS1R1 = kS1Q2 + 4 * 2048;
constexpr int kS1R2 = kS1R1 + 2 * 2048;
template <int MODE> constexpr int s1_weight_bytes() { return MODE == 0 ? kS1Q2 : kS1R2 + 2 * 2048; }
// ... then per-channel parameters (floats) ...
enum S1Par { kS1pConv0B = 0, kS1pQ1B = 32, kS1pD1B = 64, kS1pGlnG = 128, kS1pGlnB = 160, kS1pMixB1 = 192, kS1pD2B = 256,
             kS1pQ2B = 288, kS1pR1B = 320, kS1pR2B = 352, kS1pLut = 384, kS1ParFloats = 640 };
// ... then one transposed token tile per wave.
template <int MODE> constexpr int s1_waves() { return MODE == 0 ? 12 : 8; }      // 3 / 2 waves per SIMD
template <int MODE> constexpr int s1_lds_bytes() {
    return s1_weight_bytes<MODE>() + kS1ParFloats * 4 + s1_waves<MODE>() * kS1BtBytes;
}

// sum over the four lanes l, l^16, l^32, l^48 of TWO values at once with the gfx950 row swaps (pure VALU; the
// ds_bpermute form of __shfl_xor goes through the LDS crossbar and its latency).  Rows = 16-lane groups r0..r3.
__device__ __forceinline__ void quarter_allreduce2(float &s, float &ss) {
    // (the builtin pads the VALU-write -> permlane-read hazard itself; it mis-folds a swap whose two operands are the
    // same SSA value -- both results come out as one register -- hence the opaque copies)
    auto swap16 = [](unsigned a, unsigned b) { return __builtin_amdgcn_permlane16_swap(a, b, false, false); };
    auto swap32 = [](unsigned a, unsigned b) { return __builtin_amdgcn_permlane32_swap(a, b, false, false); };
    auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
    auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
    const auto r0 = swap16(u(s), u(ss));                  // [s0 ss0 s2 ss2], [s1 ss1 s3 ss3]
    const float c = f(r0[0]) + f(r0[1]);                  // [S01 SS01 S23 SS23]
    unsigned c1 = u(c);
    asm("" : "+v"(c1));
    const auto r1 = swap32(u(c), c1);                     // [S01 SS01 S01 SS01], [S23 SS23 S23 SS23]
    const float d = f(r1[0]) + f(r1[1]);                  // [S SS S SS]
    unsigned d1 = u(d);
    asm("" : "+v"(d1));
    const auto r2 = swap16(u(d), d1);                     // [S S S S], [SS SS SS SS]
    s = f(r2[0]);
    ss = f(r2[1]);
}

// LayerNorm statistics of pixel tile p in one pass (sum and sum of squares; var = E[x^2] - mean^2: the inputs here are
// O(1) activations with |mean| of the order of the deviation, so the cancellation costs ~1e-6 relative on the variance)
template <int NT, int P>
__device__ __forceinline__ void ln_stats1(const f4 (&x)[NT][P], int p, float &rstd, float &shift) {
    constexpr float inv_c = 1.0f / (16 * NT);
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s += x[nt][p][r];
            ss = fmaf(x[nt][p][r], x[nt][p][r], ss);
        }
    quarter_allreduce2(s, ss);
    const float mean = s * inv_c;
    const float var = fmaf(ss, inv_c, -mean * mean);
    rstd = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
    shift = -mean * rstd;
}

// (x - mean) * rstd, split into the B fragments of the next Linear (affine part folded into its weights)
template <int P>
__device__ __forceinline__ void s1_ln_split(const f4 (&x)[2][P], HL (&b)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float rstd, shift;
        ln_stats1(x, p, rstd, shift);
        f4 y0, y1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y0[r] = fmaf(x[0][p][r], rstd, shift);
            y1[r] = fmaf(x[1][p][r], rstd, shift);
        }
        b[p] = split8(y0, y1);
    }
}

template <int P>
__device__ __forceinline__ void s1_split(const f4 (&x)[2][P], HL (&b)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) b[p] = split8(x[0][p], x[1][p]);
}

// acc[nt][p] (+)= W(row tile nt) . B[p]: one K-step, weight fragments from the LDS image (wl already + lane * 16)
template <int NT, int P>
__device__ __forceinline__ void s1_linear(f4 (&acc)[NT][P], const unsigned char *wl, int tile_stride, const HL (&b)[P]) {
    HL a[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        a[nt].hi = *reinterpret_cast<const h8 *>(wl + nt * tile_stride);
        a[nt].lo = *reinterpret_cast<const h8 *>(wl + nt * tile_stride + 1024);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[nt][p] = mfma16(a[nt].lo, b[p].hi, acc[nt][p]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[nt][p] = mfma16(a[nt].hi, b[p].lo, acc[nt][p]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[nt][p] = mfma16(a[nt].hi, b[p].hi, acc[nt][p]);
}

template <int NT, int P>
__device__ __forceinline__ void s1_bias(f4 (&t)[NT][P], const float *par, int q) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f4 b = *reinterpret_cast<const f4 *>(par + 16 * nt + 4 * q);
#pragma unroll
        for (int p = 0; p < P; ++p) t[nt][p] = b;
    }
}

// add the value of lane (l + n) mod 16 of the same 16-lane row (DPP row_ror): 4 steps = sum over the row in every lane
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, r);
}

template <int MODE>
__global__ __launch_bounds__(s1_waves<MODE>() * 64, 1) void stage1_kernel16(StageArgs A) {
    constexpr int C = kS1C, P = 4, NW = s1_waves<MODE>(), NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *par = reinterpret_cast<float *>(smem_raw + s1_weight_bytes<MODE>());
    const int lane = threadIdx.x & 63, q = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    // ---- stage the weights, parameters and the uint8 table once per workgroup ----
    {
        auto copy = [&](int dst, int src_floats, int bytes) {
            const char *s = reinterpret_cast<const char *>(blob + src_floats);
            for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16)
                *reinterpret_cast<uint4 *>(smem_raw + dst + i) = *reinterpret_cast<const uint4 *>(s + i);
        };
        copy(kS1Q1, S.q1_w + MODE * (2 * 512), 2 * 2048);      // rows MODE*C .. : tiles 2*MODE, 2*MODE+1 (512 floats each)
        copy(kS1D1, Br.d1_w, 4 * 2048);
        copy(kS1D2, Br.d2_w, 2 * 2048);
        if (MODE == 1) {
            copy(kS1Q2, S.q2_w, 4 * 2048);                       // tiles (nt, ks): nt*2 + ks
            copy(kS1R1, S.r1_w, 2 * 2048);
            copy(kS1R2, S.r2_w, 2 * 2048);
        }
        // token-mix matrix: output column li of token tile pt is token 4 li + pt (natural fragments: 16 nt + col)
        for (int i = threadIdx.x; i < 8 * 2 * 64; i += NTHR) {
            const int l = i & 63, part = (i >> 6) & 1, tile = i >> 7, pt = tile >> 1, ks = tile & 1;
            const int g = 4 * (l & 15) + pt;
            const int stile = (g >> 4) * 2 + ks, sl = (g & 15) + 16 * (l >> 4);
            const char *s = reinterpret_cast<const char *>(blob + Br.mix_w) + stile * 2048 + part * 1024 + sl * 16;
            *reinterpret_cast<uint4 *>(smem_raw + kS1Mix + tile * 2048 + part * 1024 + l * 16) =
                *reinterpret_cast<const uint4 *>(s);
        }
        // conv0 [32, 3] as A fragments with K padded to 32: input channel c in k-slot (q = 0, j = c)
        for (int i = threadIdx.x; i < 2 * 64; i += NTHR) {
            const int l = i & 63, nt = i >> 6;
            h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((l >> 4) == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float w = blob[S.conv0_w + (16 * nt + (l & 15)) * 3 + j];
                    hi[j] = (_Float16)w;
                    lo[j] = (_Float16)(w - (float)hi[j]);
                }
            }
            *reinterpret_cast<h8 *>(smem_raw + kS1Conv0 + nt * 2048 + l * 16) = hi;
            *reinterpret_cast<h8 *>(smem_raw + kS1Conv0 + nt * 2048 + 1024 + l * 16) = lo;
        }
        for (int i = threadIdx.x; i < kS1ParFloats; i += NTHR) {
            float v;
            if (i < kS1pQ1B) v = blob[S.conv0_b + i];
            else if (i < kS1pD1B) v = blob[S.q1_b + MODE * C + (i - kS1pQ1B)];
            else if (i < kS1pGlnG) v = blob[Br.d1_b + (i - kS1pD1B)];
            else if (i < kS1pGlnB) v = blob[Br.gln_g + (i - kS1pGlnG)];
            else if (i < kS1pMixB1) v = blob[Br.gln_b + (i - kS1pGlnB)];
            else if (i < kS1pD2B) v = blob[Br.mix_b + (i - kS1pMixB1)] + 1.0f;
            else if (i < kS1pQ2B) v = blob[Br.d2_b + (i - kS1pD2B)];
            else if (i < kS1pR1B) v = blob[S.q2_b + (i - kS1pQ2B)];
            else if (i < kS1pR2B) v = blob[S.r1_b + (i - kS1pR1B)];
            else if (i < kS1pLut) v = blob[S.r2_b + (i - kS1pR2B)];
            else v = blob[kLayout.u8_lut + (i - kS1pLut)];
            par[i] = v;
        }
        __syncthreads();                                         // the only barrier of the kernel
    }

    _Float16 *bT = reinterpret_cast<_Float16 *>(smem_raw + s1_weight_bytes<MODE>() + kS1ParFloats * 4 + wave * kS1BtBytes);
    const unsigned char *wl = smem_raw + lane * 16;              // weight fragments: + region + tile * 2048 (+ 1024: lo)

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    const int total = A.B * per_img;
    // XCD-aware persistent schedule (speed only): workgroups b and b + 8 share an XCD (L2); hand each XCD runs of
    // consecutive groups -- neighbours read the same lines of the strided grid gather -- one group per wave per round.
    const int nx = (gridDim.x >> 3) * NW;                        // waves per XCD (gridDim.x is a multiple of 8)
    const int wx = (blockIdx.x >> 3) * NW + wave, xcd = blockIdx.x & 7;

    for (int round = 0;; ++round) {
        const int item = (round * 8 + xcd) * nx + wx;            // wave-uniform
        if (item >= total) break;
        const int n = item / per_img, rem = item - n * per_img;
        const int gy = rem / fw, gx = rem - gy * fw;             // block (by, bx) or in-cell offset (iy, ix)
        const int ty = li >> 1, tx0 = 4 * (li & 1);
        long pix0;                                               // pixel of tile 0; tile p is pix0 + p * pstep
        int y, x0, pstep;
        if (MODE == 0) { y = ty * fh + gy; x0 = tx0 * fw + gx; pstep = fw; }
        else           { y = 8 * gy + ty;  x0 = 8 * gx + tx0;  pstep = 1; }
        pix0 = ((long)n * H + y) * W + x0;

        // ---- network input of the lane's four pixels as the B fragments of conv0 (k-slots: q = 0, j = 0..2) ----
        HL bx[P];
        {
            float in[P][3];
            if (q == 0) {
                if (A.u8_ch == 0) {
                    const long hw = (long)H * W;
                    const float *xp = A.X + (long)n * 3 * hw + (long)y * W + x0;
                    if (MODE == 1) {                             // four adjacent pixels: 16-byte aligned
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const f4 v = ldg4(xp + k * hw);
#pragma unroll
                            for (int p = 0; p < P; ++p) in[p][k] = v[p];
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < P; ++p)
#pragma unroll
                            for (int k = 0; k < 3; ++k) in[p][k] = xp[k * hw + p * pstep];
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const int yy = y - A.u8_top, xx = x0 + p * pstep - A.u8_left;
                        const bool ok = yy >= 0 && yy < A.u8_h && xx >= 0 && xx < A.u8_w;
                        const unsigned char *px8 =
                            A.X8 + (((long)n * A.u8_h + (ok ? yy : 0)) * A.u8_w + (ok ? xx : 0)) * A.u8_ch;
#pragma unroll
                        for (int k = 0; k < 3; ++k) in[p][k] = ok ? par[kS1pLut + px8[A.u8_ch == 3 ? k : 0]] : 0.0f;
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) in[p][0] = in[p][1] = in[p][2] = 0.0f;
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                h2 h01, l01, h2x, l2x;
                split_pair(in[p][0], in[p][1], h01, l01);
                split_pair(in[p][2], 0.0f, h2x, l2x);
                bx[p].hi = h8{h01[0], h01[1], h2x[0], 0, 0, 0, 0, 0};
                bx[p].lo = h8{l01[0], l01[1], l2x[0], 0, 0, 0, 0, 0};
            }
        }
        auto conv0 = [&](f4 (&x0v)[2][P]) {                      // x0 = relu(conv0(X)); bit-identical every time
            s1_bias(x0v, par + kS1pConv0B, q);
            s1_linear(x0v, wl + kS1Conv0, 2048, bx);
            relu(x0v);
        };

        HL b[P];                                                 // B fragments of the next Linear
        {
            f4 x0v[2][P];
            conv0(x0v);
            s1_ln_split(x0v, b);
        }
        f4 z[2][P];                                              // u (grid) / v (block): kept for the branch residual
        s1_bias(z, par + kS1pQ1B, q);
        s1_linear(z, wl + kS1Q1, 2048, b);
        gelu<false>(z);
        s1_ln_split(z, b);

        f4 ga[2][P];
        s1_bias(ga, par + kS1pD1B, q);
        s1_linear(ga, wl + kS1D1, 2048, b);
        gelu<false>(ga);
        {
            f4 gb[2][P];
            s1_bias(gb, par + kS1pD1B + C, q);
            s1_linear(gb, wl + kS1D1 + 2 * 2048, 2048, b);
            gelu<false>(gb);
            // gating LayerNorm (affine) -> transposed token tile bT[hi|lo][c][t], t = 4 li + p: 8-byte stores
            float rstd[P], shift[P];
#pragma unroll
            for (int p = 0; p < P; ++p) ln_stats1(gb, p, rstd[p], shift[p]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f4 gg = *reinterpret_cast<const f4 *>(par + kS1pGlnG + 16 * nt + 4 * q);
                const f4 bb = *reinterpret_cast<const f4 *>(par + kS1pGlnB + 16 * nt + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v[P];
#pragma unroll
                    for (int p = 0; p < P; ++p) v[p] = fmaf(fmaf(gb[nt][p][r], rstd[p], shift[p]), gg[r], bb[r]);
                    h2 h01, l01, h23, l23;
                    split_pair(v[0], v[1], h01, l01);
                    split_pair(v[2], v[3], h23, l23);
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    _Float16 *row = bT + (16 * nt + 4 * q + r) * kS1Pitch + 4 * li;
                    *reinterpret_cast<h4 *>(row) = h4{h01[0], h01[1], h23[0], h23[1]};
                    *reinterpret_cast<h4 *>(row + C * kS1Pitch) = h4{l01[0], l01[1], l23[0], l23[1]};
                }
            }
        }
        HL ub[(MODE == 1) ? P : 1];                              // block: u' rows of the lane's pixels (pre-split), needed
        if constexpr (MODE == 1) {                               // by RSHMAG.dense2 two Linears from here
#pragma unroll
            for (int p = 0; p < P; ++p) ub[p] = load_frag_px(A.U, pix0 + p * pstep, C, 0, q);
        }
        {
            // mix^T[c][g'] = sum_g bT[c][g] Wmix[g'][g] (+ bias[g'] + 1 as the accumulator's start value), then the gate
            HL a[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const _Float16 *row = bT + (16 * ct + li) * kS1Pitch + 32 * kk + 8 * q;
                    a[ct][kk].hi = *reinterpret_cast<const h8 *>(row);
                    a[ct][kk].lo = *reinterpret_cast<const h8 *>(row + C * kS1Pitch);
                }
            const f4 mb1 = *reinterpret_cast<const f4 *>(par + kS1pMixB1 + 4 * li);
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                HL w0, w1;
                w0.hi = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 2 + 0) * 2048);
                w0.lo = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 2 + 0) * 2048 + 1024);
                w1.hi = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 2 + 1) * 2048);
                w1.lo = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 2 + 1) * 2048 + 1024);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    f4 m = {mb1[pt], mb1[pt], mb1[pt], mb1[pt]};
                    m = mfma16x3(a[ct][0], w0, m);
                    m = mfma16x3(a[ct][1], w1, m);
                    ga[ct][pt] *= m;
                }
            }
        }
        s1_split(ga, b);
        f4 o[2][P];
        s1_bias(o, par + kS1pD2B, q);
        s1_linear(o, wl + kS1D2, 2048, b);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) store_frag_px(A.U, pix0 + p * pstep, C, 0, q, split8(o[0][p], o[1][p]));
        } else {
            s1_split(o, b);
            f4 x1[2][P];
            s1_bias(x1, par + kS1pQ2B, q);
            s1_linear(x1, wl + kS1Q2 + 2048, 2 * 2048, b);       // K-step 1 = v' half of cat[u', v']
            s1_linear(x1, wl + kS1Q2, 2 * 2048, ub);             // K-step 0 = u' half
            {
                f4 x0v[2][P];
                conv0(x0v);                                      // recomputed (24 MFMAs) instead of kept (32 registers)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        x1[nt][p] += x0v[nt][p];
                        *reinterpret_cast<f4 *>(A.R + (pix0 + p * pstep) * C + 16 * nt + 4 * q) = x1[nt][p] + x0v[nt][p];
                    }
            }
            s1_ln_split(x1, b);
            f4 m1[2][P];
            s1_bias(m1, par + kS1pR1B, q);
            s1_linear(m1, wl + kS1R1, 2048, b);
            lrelu(m1);
            s1_split(m1, b);
            f4 t[2][P];
            s1_bias(t, par + kS1pR2B, q);
            s1_linear(t, wl + kS1R2, 2048, b);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f4 s = t[nt][0];
                *reinterpret_cast<f4 *>(A.T + pix0 * C + 16 * nt + 4 * q) = t[nt][0];
#pragma unroll
                for (int p = 1; p < P; ++p) {
                    *reinterpret_cast<f4 *>(A.T + (pix0 + p * pstep) * C + 16 * nt + 4 * q) = t[nt][p];
                    s += t[nt][p];
                }
                // channel sums over the group's 64 pixels (fixed order): over the 16 lanes of the row, then lane li = 0 stores
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s[r]))));
                if (li == 0) *reinterpret_cast<f4 *>(A.partial + (long)item * C + 16 * nt + 4 * q) = s;
            }
        }
    }
}
