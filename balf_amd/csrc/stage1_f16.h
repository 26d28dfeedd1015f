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
//     no weight ever comes from L2 inside the loop; conv0 (3 -> 32) runs on the matrix pipe too, as eight exact-fp32
//     v_mfma_f32_16x16x4_f32 (K = 3 padded to 4: lane quarter q feeds input channel q, no operand split).
// Token t = 8 ty + tx of a group sits in MFMA column li of pixel tile p with t = 4 li + p: a lane's four tiles are
// four ADJACENT tokens, so the transposed tile is written with 8-byte LDS stores and (block branch) the NCHW input
// is read with one 16-byte load per colour plane.  The mixing matrix is re-ordered to that column order when it is
// staged.  Results are independent of the grid size and of which wave processes which group.
#pragma once

constexpr int kS1C = 32;
#ifndef BALF_S1_SPLIT_MIX
#define BALF_S1_SPLIT_MIX 2   // operand split form of the stage-1 kernels (split16.h)
#endif
// Transposed token tile of a wave (32 channels x 64 tokens, hi and lo planes of 4 KB): rows of 128 B whose eight 16-B
// chunks are XOR-swizzled by the row, so that the 8-byte writes (a lane's four adjacent tokens of one channel) and the
// 16-byte A-fragment reads (eight tokens of a row) both spread over the LDS banks without padding.
constexpr int kS1BtPlane = kS1C * 128;
constexpr int kS1BtBytes = 2 * kS1BtPlane;                     // 8192 B per wave
// byte offset of tokens 4 li .. 4 li + 3 of channel row c (writer) / of tokens 8 j .. 8 j + 7 of row c (reader)
__device__ __forceinline__ int s1_bt_wr(int c, int li) { return c * 128 + (((li >> 1) ^ (c & 7)) << 4) + (li & 1) * 8; }
__device__ __forceinline__ int s1_bt_rd(int c, int j) { return c * 128 + ((j ^ (c & 7)) << 4); }
// LDS image: the GELU chord table at offset 0 (its byte offsets come straight out of a bit mask, see gelu_lut1), then
// weight tiles (2 KiB each: [hi 64 x 16 B][lo 64 x 16 B]) ...
#ifndef BALF_GELU_LUT
#define BALF_GELU_LUT 1      // 1: GELU of the stage-1 kernels from the LDS chord table (3 vector instructions + 1 LDS read); 0: 2^P form (8)
#endif
constexpr int kS1LutBytes = BALF_GELU_LUT ? ((kGeluLutN + 1) * 8 + 15) / 16 * 16 : 0;
constexpr int kS1Conv0 = kS1LutBytes;                          // 2 row tiles (built in the kernel from the plain [32,3] matrix)
constexpr int kS1Q1 = kS1Conv0 + 2 * 2048;                     // 2 row tiles (this branch's half of RSHMAG.dense1)
constexpr int kS1D1 = kS1Q1 + 2 * 2048;                        // 4 row tiles
constexpr int kS1Mix = kS1D1 + 4 * 2048;                       // 4 token tiles x 2 K-steps, columns re-ordered
constexpr int kS1D2 = kS1Mix + 8 * 2048;                       // 2 row tiles
constexpr int kS1Q2 = kS1D2 + 2 * 2048;                        // block only: 2 row tiles x 2 K-steps
constexpr int kS1R1 = kS1Q2 + 4 * 2048;
constexpr int kS1R2 = kS1R1 + 2 * 2048;
template <int MODE> constexpr int s1_weight_bytes() { return MODE == 0 ? kS1Q2 : kS1R2 + 2 * 2048; }   // MODE 2 (tail): the block image (table region unused)
// ... then per-channel parameters (floats) ...
enum S1Par { kS1pConv0B = 0, kS1pQ1B = 32, kS1pD1B = 64, kS1pGlnG = 128, kS1pGlnB = 160, kS1pMixB1 = 192, kS1pD2B = 256,
             kS1pQ2B = 288, kS1pR1B = 320, kS1pR2B = 352, kS1pLut = 384, kS1ParFloats = 640 };
// ... then one transposed token tile per wave.
#ifndef BALF_S1_NW0
#define BALF_S1_NW0 8        // (12 = three waves per SIMD was best with the 2^P GELU; with the table reads in flight it spills at 168 registers)
#endif
#ifndef BALF_S1_NW1
#define BALF_S1_NW1 8
#endif
#ifndef BALF_S1_NW2
#define BALF_S1_NW2 12
#endif
template <int MODE> constexpr int s1_waves() { return MODE == 0 ? BALF_S1_NW0 : MODE == 1 ? BALF_S1_NW1 : BALF_S1_NW2; }   // 2 / 2 / 3 waves per SIMD
template <int MODE> constexpr int s1_lds_bytes() {
    return s1_weight_bytes<MODE>() + kS1ParFloats * 4 + (MODE == 2 ? 0 : s1_waves<MODE>() * kS1BtBytes);   // (no token mix in the tail)
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
    float s = x[0][p][0], ss = x[0][p][0] * x[0][p][0];      // (not 0 + x: hipcc keeps that add -- it turns -0 into +0)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = (nt == 0 ? 1 : 0); r < 4; ++r) {
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
        b[p] = split8<BALF_S1_SPLIT_MIX>(y0, y1);
    }
}

template <int P>
__device__ __forceinline__ void s1_split(const f4 (&x)[2][P], HL (&b)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) b[p] = split8<BALF_S1_SPLIT_MIX>(x[0][p], x[1][p]);
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
        for (int p = 0; p < P; ++p) if (!BALF_DROP_WLO) acc[nt][p] = mfma16(a[nt].lo, b[p].hi, acc[nt][p]);
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

// GELU from the chord table in LDS (layout.h: kGeluLutN intervals over [-L, L), weights.hip builds it):
//   y = clamp01(x / 2L + 1/2)            v_fma_f32 ... clamp  (saturates to the two asymptote entries outside [-L, L))
//   t = y * 8N + 1.5 * 2^23              the rounded 8 N y lands in the low mantissa bits
//   (a, b) = table[bits(t) & 0x7FF8]     byte offset of interval floor(round(8 N y) / 8): one ds_read_b64
//   gelu = a + b x
// 3 vector instructions (9.4 issue cycles at the measured class rates) + 1 LDS read against 8 (33 cycles) for the 2^P
// form; the LDS pipe of these kernels was idle three quarters of the time.  Chord error 7.6e-7 (the 2^P form: 6.4e-7).
__device__ __forceinline__ unsigned gelu_lut_off(float x, float magic) {
    static_assert((kGeluLutN + 1) * 8 <= 0x8000, "the byte offset mask below is 15 bits");
    // (x is an MFMA result: the instruction that reads it must be the compiler's -- hipcc pads the MFMA -> VALU read
    // hazard for its own instructions only, an inline-asm v_fma placed right behind the MFMA read stale registers;
    // fmed3(., 0, 1) folds into the fma's clamp modifier)
    const float y = __builtin_amdgcn_fmed3f(fmaf(x, 0.5f / kGeluLutL, 0.5f), 0.0f, 1.0f);
    const float t = fmaf(y, 8.0f * kGeluLutN, magic);
    return __builtin_bit_cast(unsigned, t) & 0x7FF8u;
}

// A tensor's 32 values per lane go through the table as a software pipeline in chunks of BALF_S1_GELU_CH: ordinary LDS
// loads from raw addresses (the dynamic LDS block of these kernels starts at LDS address 0 -- they have no static
// __shared__ --, so the masked bits ARE the address; hipcc otherwise spends a v_add_u32 per read on adding the LDS
// symbol's zero), the reads of chunk k + 1 issued before the fmas of chunk k, the source order pinned by scheduling fences
// that only matrix and scalar instructions may cross -- hipcc's own counted lgkmcnt waits then leave the younger reads in
// flight.  The final fma is asm with the result in x's own register: left to itself hipcc pairs two of them into a
// v_pk_fma_f32 behind three v_mov_b32, or (as v_fmac_f32) leaves each result in one half of the 64-bit pair the read
// returned, which fragments the register file into 60-80 spills.  (A first form issued eight reads and waited for them
// inside one asm statement: 5 % slower on both kernels -- the LDS round trip of every chunk was exposed.)
#ifndef BALF_S1_GELU_CH
#define BALF_S1_GELU_CH 8
#endif
template <int CH>
__device__ __forceinline__ void gelu_lut_pipe(f4 (&t)[2][4], float magic) {
    constexpr int NCH = 32 / CH;
    typedef const f2 __attribute__((address_space(3))) *lds_f2_ptr;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = t[i >> 4][(i >> 2) & 3][i & 3];
    auto fence = [] { __builtin_amdgcn_sched_barrier(0x8 | 0x4); };          // MFMA and SALU may cross, VALU and LDS not
    f2 ab[2][CH];
    auto issue = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) ab[buf][i] = *reinterpret_cast<lds_f2_ptr>(gelu_lut_off(v[c * CH + i], magic));
    };
    auto finish = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
            asm("v_fma_f32 %0, %1, %0, %2" : "+v"(v[c * CH + i]) : "v"(ab[buf][i][1]), "v"(ab[buf][i][0]));
    };
    issue(0, 0);
    fence();
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        issue(c, c & 1);
        fence();
        finish(c - 1, (c - 1) & 1);
        fence();
    }
    finish(NCH - 1, (NCH - 1) & 1);
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i >> 4][(i >> 2) & 3][i & 3] = v[i];
}

__device__ __forceinline__ void s1_gelu(f4 (&t)[2][4]) {
#if BALF_GELU_LUT
    if (BALF_ABLATE_GELU) return;
    float magic = 12582912.0f;                   // 1.5 * 2^23, kept in a vector register (the fma's other two operands use the constant bus)
    asm("" : "+v"(magic));
    gelu_lut_pipe<BALF_S1_GELU_CH>(t, magic);
#else
    gelu<false>(t);
#endif
}

// Vector-memory discipline of the float-input kernels (U8 = false).  vmcnt counts loads and stores together, in issue
// order, and the compiler drains it to zero at the loop's back edge as soon as it has a load of its own pending -- which
// makes every group wait for the previous group's 8-16 KiB of stores (measured: 40 % of the grid kernel's time).  So every
// LOAD of the loop is inline asm with a counted wait placed by hand (the compiler, seeing only stores, never waits):
//   top of group i:   wait for the input pixels of group i       (younger operations: the stores of group i-1 -> vmcnt(8))
//                     issue the u' rows of group i (block branch), then the input pixels of group i+1
//   before RSHMAG.dense2 (block): wait for the u' rows          (younger: the input load -> vmcnt(1))
// The loads of the last group's successor are issued anyway (clamped to a valid group) so that the counts are static.
// a wave-uniform pointer as a scalar-register pair (hipcc does 64-bit multiplies of uniform values on the vector unit
// and then hands the asm's "s" operand a VGPR pair)
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const T *>(((unsigned long long)hi << 32) | lo);
}

// The block kernel stores x1 and the channel sums of the RCAB's hidden layer only; the tail kernel (MODE 2) recomputes the
// RCAB branch from x1.  (The round-2 alternative that stored T and R for a pool kernel is gone.)
#ifndef BALF_S1_KEEP_X0
#define BALF_S1_KEEP_X0 1    // block kernel: keep x0 in registers (32; there is room since conv0 left the f16 path) instead of recomputing it: -5 %
#endif
#ifndef BALF_S1_STRICT
#define BALF_S1_STRICT 0     // 1: every hand-placed wait drains the queue (debugging aid)
#endif
#if BALF_S1_STRICT
#define BALF_S1_WAIT_IN "s_waitcnt vmcnt(0)"
#define BALF_S1_WAIT_U "s_waitcnt vmcnt(0)"
#else
#define BALF_S1_WAIT_IN "s_waitcnt vmcnt(8)"
#define BALF_S1_WAIT_U "s_waitcnt vmcnt(1)"
#endif
template <int MODE, bool U8>
__global__ __launch_bounds__(s1_waves<MODE>() * 64, 1) void stage1_kernel16(StageArgs A) {
    constexpr int C = kS1C, P = 4, NW = s1_waves<MODE>(), NTHR = NW * 64;
    constexpr int BM = MODE == 0 ? 0 : 1;                        // branch whose weights / token geometry this kernel uses
    constexpr bool TAIL = MODE == 2;                             // the stage's tail (see the loop body)
    constexpr int STAMP_KID = BM; (void)STAMP_KID;
    STAMP_DECL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *par = reinterpret_cast<float *>(smem_raw + s1_weight_bytes<MODE>());
    const int lane = threadIdx.x & 63, q = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[BM];

    // ---- stage the weights, parameters and the uint8 table once per workgroup ----
    {
        auto copy = [&](int dst, int src_floats, int bytes) {
            const char *s = reinterpret_cast<const char *>(blob + src_floats);
            for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16)
                *reinterpret_cast<uint4 *>(smem_raw + dst + i) = *reinterpret_cast<const uint4 *>(s + i);
        };
        if (BALF_GELU_LUT && !TAIL) copy(0, kLayout.gelu_lut, kS1LutBytes);
        copy(kS1Q1, S.q1_w + BM * (2 * 512), 2 * 2048);        // rows BM*C .. : tiles 2*BM, 2*BM+1 (512 floats each)
        copy(kS1D1, Br.d1_w, 4 * 2048);
        copy(kS1D2, Br.d2_w, 2 * 2048);
        if (BM == 1) {
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
        // conv0 [32, 3] as A fragments of v_mfma_f32_16x16x4_f32 (exact fp32, K = 3 padded to 4): lane (li, q) holds
        // W[16 nt + li][q] (q < 3)
        for (int i = threadIdx.x; i < 2 * 64; i += NTHR) {
            const int l = i & 63, nt = i >> 6, k = l >> 4;
            *reinterpret_cast<float *>(smem_raw + kS1Conv0 + nt * 256 + l * 4) =
                k < 3 ? blob[S.conv0_w + (16 * nt + (l & 15)) * 3 + k] : 0.0f;
        }
        for (int i = threadIdx.x; i < kS1ParFloats; i += NTHR) {
            float v;
            if (i < kS1pQ1B) v = blob[S.conv0_b + i];
            else if (i < kS1pD1B) v = blob[S.q1_b + BM * C + (i - kS1pQ1B)];
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

    unsigned char *bT = smem_raw + s1_weight_bytes<MODE>() + kS1ParFloats * 4 + wave * kS1BtBytes;
    const unsigned char *wl = smem_raw + lane * 16;              // weight fragments: + region + tile * 2048 (+ 1024: lo)

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    const int total = A.B * per_img;
    // XCD-aware persistent schedule (speed only): workgroups b and b + 8 share an XCD (L2); hand each XCD runs of
    // consecutive groups -- neighbours read the same lines of the strided grid gather -- one group per wave per round.
    const int nx = (gridDim.x >> 3) * NW;                        // waves per XCD (gridDim.x is a multiple of 8)
    const int wx = (blockIdx.x >> 3) * NW + wave, xcd = blockIdx.x & 7;

    // group index -> (image n, gy, gx) [block (by, bx) or in-cell offset (iy, ix)], advanced incrementally by the
    // schedule's stride (two integer divisions per group would cost ~80 scalar instructions each)
    struct Pos { int n, gy, gx; };
    struct Geo { int n, y, x0; long pix0; };
    const int ty = li >> 1, tx0 = 4 * (li & 1);
    const int pstep = (MODE == 0) ? fw : 1;
    auto decompose = [&](int i) {
        Pos c;
        c.n = i / per_img;
        const int rem = i - c.n * per_img;
        c.gy = rem / fw;
        c.gx = rem - c.gy * fw;
        // (integer division runs on the vector unit: bring the wave-uniform results back to scalar registers, the asm
        // loads below take their base addresses in SGPRs)
        c.n = __builtin_amdgcn_readfirstlane(c.n);
        c.gy = __builtin_amdgcn_readfirstlane(c.gy);
        c.gx = __builtin_amdgcn_readfirstlane(c.gx);
        return c;
    };
    auto geo = [&](const Pos &c) {
        Geo g;
        g.n = c.n;
        if (MODE == 0) { g.y = ty * fh + c.gy; g.x0 = tx0 * fw + c.gx; }
        else           { g.y = 8 * c.gy + ty;  g.x0 = 8 * c.gx + tx0; }
        g.pix0 = ((long)g.n * H + g.y) * W + g.x0;
        return g;
    };
    // raw network input of the lane's four pixels: lane quarter q holds input channel q (conv0's B operand: k = q; the
    // fourth quarter is the zero padding of K) -- float bits, or (U8) the uint8 value, 0x100 = outside the image
    const int cq = q < 3 ? q : 2;
    auto load_raw_u8 = [&](const Geo &g, unsigned (&raw)[P]) {
        if (q >= 3) return;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int yy = g.y - A.u8_top, xx = g.x0 + p * pstep - A.u8_left;
            const bool ok = yy >= 0 && yy < A.u8_h && xx >= 0 && xx < A.u8_w;
            const unsigned char *px8 = A.X8 + (((long)g.n * A.u8_h + (ok ? yy : 0)) * A.u8_w + (ok ? xx : 0)) * A.u8_ch;
            raw[p] = ok ? (unsigned)px8[A.u8_ch == 3 ? cq : 0] : 0x100u;
        }
    };
    const int hw = H * W;                                       // (32-bit: keeps the plane offsets on the scalar unit)
    // float input, asm loads (not counted by the compiler): the image's planes at a scalar base, the lane's plane and pixel
    // offset in a VGPR.  Lanes q = 3 are masked off and keep the zeros `raw` starts with.
    const unsigned qoff = (unsigned)(cq * hw) * 4u;
    // The prefetched pixels land in registers of their own (nraw / nrawv: zero, then the load -- masked-off lanes q = 3
    // keep the zero), and ONE statement waits for them and copies them into the registers the group computes from
    // (BALF_S1_TAKE_*: the copies sit behind the s_waitcnt inside the statement).  Round 2 loaded into the compute
    // variable's own register ("+v") and waited in a second statement whose operand was tied to the same variable: the
    // variable is live across the loop, so hipcc had to copy the load's destination into the loop-carried register IN
    // FRONT of the wait, a whole iteration after the load -- correct only as long as a load never takes longer than an
    // iteration (tools/vmcnt_audit.py; the check is now part of tests/test_build_invariants.py).
    auto issue_raw = [&](const Geo &g, unsigned (&nraw)[P], f4 &nrawv) {
        const unsigned voff = (unsigned)(g.y * W + g.x0) * 4u + qoff;
        const float *xb = uniform_ptr(A.X + (long)g.n * 3 * (long)hw);
#pragma unroll
        for (int p = 0; p < P; ++p) { nraw[p] = 0u; nrawv[p] = 0.0f; }
        if (q < 3) {
            if constexpr (BM == 1) {
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(nrawv) : "v"(voff), "s"(xb) : "memory");
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const unsigned vp = voff + (unsigned)(p * pstep) * 4u;
                    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(nraw[p]) : "v"(vp), "s"(xb) : "memory");
                }
            }
        }
    };
    // wait (WAIT = the s_waitcnt text), then raw <- nraw.  "+v" on raw: the copies write registers that hold the previous
    // group's pixels, live VALU-visible values -- never a register an MFMA in flight still reads (split16.h).
    auto take_raw = [&](unsigned (&raw)[P], f4 &rawv, const unsigned (&nraw)[P], const f4 &nrawv) {
        if constexpr (BM == 1)
            asm volatile(BALF_S1_WAIT_IN "\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "+v"(rawv[0]), "+v"(rawv[1]), "+v"(rawv[2]), "+v"(rawv[3])
                         : "v"(nrawv[0]), "v"(nrawv[1]), "v"(nrawv[2]), "v"(nrawv[3]) : "memory");
        else
            asm volatile(BALF_S1_WAIT_IN "\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3])
                         : "v"(nraw[0]), "v"(nraw[1]), "v"(nraw[2]), "v"(nraw[3]) : "memory");
    };
    const int stride = 8 * nx;
    const Pos step = decompose(stride);
    auto advance = [&](Pos c) {
        c.gx += step.gx;
        if (c.gx >= fw) { c.gx -= fw; ++c.gy; }
        c.gy += step.gy;
        if (c.gy >= fh) { c.gy -= fh; ++c.n; }
        c.n += step.n;
        c.n = __builtin_amdgcn_readfirstlane(c.n);
        c.gy = __builtin_amdgcn_readfirstlane(c.gy);
        c.gx = __builtin_amdgcn_readfirstlane(c.gx);
        return c;
    };
    int item = xcd * nx + wx;                                    // wave-uniform
    Pos nxt = decompose(item);
    unsigned raw[P] = {}, nraw[P] = {};
    f4 rawv = {}, nrawv = {};
    if (!U8 && !TAIL && item < total) {
        issue_raw(geo(nxt), nraw, nrawv);
        // the first group has no older stores in front of its pixels: drain here (the counted wait in the loop assumes them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float a0[2];                                                 // conv0's A fragments (loop-invariant)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) a0[nt] = *reinterpret_cast<const float *>(smem_raw + kS1Conv0 + nt * 256 + lane * 4);

    for (; item < total; item += stride) {
        const Geo g = geo(nxt);
        if (item + stride < total) nxt = advance(nxt);           // (the last group re-requests its own pixels)
        const long pix0 = g.pix0;
        if (U8) load_raw_u8(g, raw);
        // (tail) plain loads, no prefetch across groups: three waves per SIMD cover the latency, and with two stores per
        // group there is no store queue to count around.  x1 as the block kernel left it, this image's SE scale.
        f4 x1t[TAIL ? 2 : 1][TAIL ? P : 1], sct[2];
        if constexpr (TAIL) {
            if constexpr (!U8) {
                if (q < 3)
                    rawv = *reinterpret_cast<const f4 *>(A.X + ((long)g.n * 3 + cq) * (long)hw + (long)g.y * W + g.x0);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sct[nt] = *reinterpret_cast<const f4 *>(A.scale + (long)g.n * C + 16 * nt + 4 * q);
#pragma unroll
                for (int p = 0; p < P; ++p)
                    x1t[nt][p] = *reinterpret_cast<const f4 *>(A.R + (pix0 + p * pstep) * C + 16 * nt + 4 * q);
            }
        }
        STAMP(0);
        float bx[P];                                             // conv0's B fragments: the lane's channel of its four pixels
        {
            if constexpr (!U8 && !TAIL) {
                // the input pixels of this group have landed; the previous group's stores may still be in flight
                take_raw(raw, rawv, nraw, nrawv);
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (U8) bx[p] = (q < 3 && raw[p] < 256u) ? par[kS1pLut + (raw[p] & 255u)] : 0.0f;
                else if (BM == 1) bx[p] = rawv[p];
                else bx[p] = __builtin_bit_cast(float, raw[p]);
            }
        }
        HL ub[(MODE == 1) ? P : 1];                              // block: u' rows of the lane's pixels (pre-split in HBM)
        if constexpr (MODE == 1) {
            if constexpr (U8) {
#pragma unroll
                for (int p = 0; p < P; ++p) ub[p] = load_frag_px(A.U, pix0 + p * pstep, C, 0, q);
            } else {
                const unsigned uoff = (unsigned)(g.y * W + g.x0) * 128u + q * 16u;
                const char *ubase = uniform_ptr(reinterpret_cast<const char *>(A.U) + (long)g.n * (long)hw * 128);
#define BALF_S1_UB(PI)                                                                                              \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 offset:%c4\n\tglobal_load_dwordx4 %1, %2, %3 offset:%c5"        \
                 : "=&v"(ub[PI].hi), "=&v"(ub[PI].lo) : "v"(uoff), "s"(ubase), "i"(PI * 128), "i"(PI * 128 + 64) : "memory")
                BALF_S1_UB(0); BALF_S1_UB(1); BALF_S1_UB(2); BALF_S1_UB(3);
#undef BALF_S1_UB
            }
        }
        if constexpr (!U8 && !TAIL) issue_raw(geo(nxt), nraw, nrawv);   // next group's pixels (4 loads or 1, always)
        STAMP(1);   // input -> conv0 B fragments (waits for the prefetched pixels), next group's loads issued
        auto conv0 = [&](f4 (&x0v)[2][P]) {                      // x0 = relu(conv0(X)); bit-identical every time
            s1_bias(x0v, par + kS1pConv0B, q);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < P; ++p) x0v[nt][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nt], bx[p], x0v[nt][p], 0, 0, 0);
            relu(x0v);
        };

        HL b[P];                                                 // B fragments of the next Linear
        if constexpr (TAIL) {
            // ---- the stage's tail: x2 = maxpool2x2(x1 + x0 + s * conv2(lrelu(conv1(LN(x1))))) in fragment format ----
            // The block kernel stored x1 and the channel sums of the RCAB's hidden layer (conv2 is linear: the SE
            // kernel gets mean(t) from mean(hidden)); the branch itself is recomputed here from x1 -- the same
            // instructions on the same values as the block kernel's, so t is the t it would have stored -- instead of
            // a T and an R tensor travelling through HBM (4.3 GB per 8 images at 1088x1920 written and read again).
            s1_ln_split(x1t, b);
            f4 m1[2][P];
            s1_bias(m1, par + kS1pR1B, q);
            s1_linear(m1, wl + kS1R1, 2048, b);
            lrelu(m1);
            s1_split(m1, b);
            f4 t[2][P];
            s1_bias(t, par + kS1pR2B, q);
            s1_linear(t, wl + kS1R2, 2048, b);
            f4 x0v[2][P];
            conv0(x0v);
            // v = r + s t with r = x1 + x0 (the order of operations of pool_kernel16), max over the 2x2 window: the
            // lane's tokens 2 pp, 2 pp + 1 are horizontal neighbours, the rows 2 j, 2 j + 1 sit in lanes li, li ^ 2
            f4 mx[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v0 = fmaf(t[nt][2 * pp][r], sct[nt][r], x1t[nt][2 * pp][r] + x0v[nt][2 * pp][r]);
                        const float v1 = fmaf(t[nt][2 * pp + 1][r], sct[nt][r], x1t[nt][2 * pp + 1][r] + x0v[nt][2 * pp + 1][r]);
                        const float m = __builtin_fmaxf(v0, v1);
                        const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, m), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false));
                        mx[nt][pp][r] = __builtin_fmaxf(m, o);
                    }
            // both lanes of a row pair hold the same two pooled pixels: lane li stores pooled column 2 (li & 1) + ((li >> 1) & 1)
            const int sel = (li >> 1) & 1;
            const unsigned selm = 0u - (unsigned)sel;          // all ones in the lanes that take the second column (v_bfi, not v_cndmask)
            f4 o0, o1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o0[r] = lane_select(selm, mx[0][1][r], mx[0][0][r]);
                o1[r] = lane_select(selm, mx[1][1][r], mx[1][0][r]);
            }
            const long opix = ((long)g.n * (H / 2) + (g.y >> 1)) * (W / 2) + (g.x0 >> 1) + sel;
            store_frag_px(A.out, opix, C, 0, q, split8<BALF_S1_SPLIT_MIX>(o0, o1));
        } else {
        f4 x0k[(MODE == 1 && BALF_S1_KEEP_X0) ? 2 : 1][(MODE == 1 && BALF_S1_KEEP_X0) ? P : 1];   // (block) x0, kept for x1 = . + x0
        if constexpr (MODE == 1 && BALF_S1_KEEP_X0) {
            conv0(x0k);
            s1_ln_split(x0k, b);
        } else {
            f4 x0v[2][P];
            conv0(x0v);
            s1_ln_split(x0v, b);
        }
        STAMP(2);   // conv0 + relu + LN + split
        f4 z[2][P];                                              // u (grid) / v (block): kept for the branch residual
        s1_bias(z, par + kS1pQ1B, q);
        s1_linear(z, wl + kS1Q1, 2048, b);
        s1_gelu(z);
        STAMP(3);   // dense1 half + GELU
        s1_ln_split(z, b);
        STAMP(4);   // LN + split

        f4 ga[2][P];
        s1_bias(ga, par + kS1pD1B, q);
        s1_linear(ga, wl + kS1D1, 2048, b);
        s1_gelu(ga);
        STAMP(5);   // branch dense1 (a half) + GELU
        {
            f4 gb[2][P];
            s1_bias(gb, par + kS1pD1B + C, q);
            s1_linear(gb, wl + kS1D1 + 2 * 2048, 2048, b);
            s1_gelu(gb);
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
                    split_pair<BALF_S1_SPLIT_MIX>(v[0], v[1], h01, l01);
                    split_pair<BALF_S1_SPLIT_MIX>(v[2], v[3], h23, l23);
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    unsigned char *row = bT + s1_bt_wr(16 * nt + 4 * q + r, li);
                    *reinterpret_cast<h4 *>(row) = h4{h01[0], h01[1], h23[0], h23[1]};
                    *reinterpret_cast<h4 *>(row + kS1BtPlane) = h4{l01[0], l01[1], l23[0], l23[1]};
                }
            }
        }
        STAMP(6);   // branch dense1 (b half) + GELU + LN + transposed tile
        {
            // mix^T[c][g'] = sum_g bT[c][g] Wmix[g'][g] (+ bias[g'] + 1 as the accumulator's start value), then the gate
            HL a[2][2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const unsigned char *row = bT + s1_bt_rd(16 * ct + li, 4 * kk + q);
                    a[ct][kk].hi = *reinterpret_cast<const h8 *>(row);
                    a[ct][kk].lo = *reinterpret_cast<const h8 *>(row + kS1BtPlane);
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
        STAMP(7);   // token mix + gate
        s1_split(ga, b);
        f4 o[2][P];
        s1_bias(o, par + kS1pD2B, q);
        s1_linear(o, wl + kS1D2, 2048, b);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

        STAMP(8);   // dense2 + residual
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) store_frag_px(A.U, pix0 + p * pstep, C, 0, q, split8<BALF_S1_SPLIT_MIX>(o[0][p], o[1][p]));
            STAMP(9);   // u' store
        } else {
            s1_split(o, b);
            f4 x1[2][P];
            s1_bias(x1, par + kS1pQ2B, q);
            s1_linear(x1, wl + kS1Q2 + 2048, 2 * 2048, b);       // K-step 1 = v' half of cat[u', v']
            if constexpr (!U8)                                   // u' rows have landed (younger: next group's input load)
                asm volatile(BALF_S1_WAIT_U
                             : "+v"(ub[0].hi), "+v"(ub[0].lo), "+v"(ub[1].hi), "+v"(ub[1].lo), "+v"(ub[2].hi), "+v"(ub[2].lo),
                               "+v"(ub[3].hi), "+v"(ub[3].lo) :: "memory");
            s1_linear(x1, wl + kS1Q2, 2 * 2048, ub);             // K-step 0 = u' half
            STAMP(9);   // RSHMAG dense2 over cat[u', v'] (u' from HBM)
            {
#if BALF_S1_KEEP_X0
                f4 (&x0v)[2][P] = x0k;
#else
                f4 x0v[2][P];
                conv0(x0v);                                      // recomputed (8 MFMAs + relu) instead of kept (32 registers)
#endif
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        x1[nt][p] += x0v[nt][p];
                        // x1 itself: the tail kernel adds x0 and the scaled RCAB branch
                        *reinterpret_cast<f4 *>(A.R + (pix0 + p * pstep) * C + 16 * nt + 4 * q) = x1[nt][p];
                    }
            }
            STAMP(10);  // conv0 again, residuals, R store
            s1_ln_split(x1, b);
            f4 m1[2][P];
            s1_bias(m1, par + kS1pR1B, q);
            s1_linear(m1, wl + kS1R1, 2048, b);
            lrelu(m1);
            // conv2 is linear: its channel means follow from the means of its input (the hidden layer m1), and the tail
            // kernel recomputes the RCAB branch from x1 -- no T tensor through HBM
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f4 s = m1[nt][0];
#pragma unroll
                for (int p = 1; p < P; ++p) s += m1[nt][p];
                // channel sums over the group's 64 pixels (fixed order): over the 16 lanes of the row, then lane li = 0 stores
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s[r]))));
                if (li == 0) *reinterpret_cast<f4 *>(A.partial + (long)item * C + 16 * nt + 4 * q) = s;
            }
            STAMP(12);  // channel sums of the hidden layer
        }
        }   // !TAIL
    }
}
