// Stage 1 (C = 32, Cin = 3) of the split-f16 detector forward: persistent, barrier-free kernels in which ONE WAVE
// OWNS A WHOLE TOKEN GROUP, on v_mfma_f32_32x32x16_f16.  Included by detector_f16.hip inside balf::{anonymous}.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 32 (a third of the forward).
//
// Structure (unchanged since round 2):
//   * a wave's pixel tiles ARE the 64 tokens of one 8x8 block (block branch) or of the 64 grid cells at one in-cell
//     offset (grid branch): the token mix is wave-local (transpose through a wave-private LDS tile, no s_barrier in the
//     main loop);
//   * the B fragments of every Linear go from the accumulator registers straight into the next MFMA, no LDS round trip;
//   * all weights of the stage (36 KB grid / 52 KB block, split-f16 fragments) are staged ONCE per workgroup, the
//     workgroup is persistent (one per CU, waves loop over groups); conv0 (3 -> 32) runs on the matrix pipe as exact fp32
//     (v_mfma_f32_32x32x2_f32, K = 3 padded to 4).
// Round 3: 32 x 32 tiles instead of 16 x 16.  The SIMD hardly overlaps matrix and vector work, and the
// 32x32x16 instruction does the same MACs in fewer, longer issues (10 % faster alone, more beside vector work:
// tools/ubench/mfma_shapes.hip) -- and it puts a pixel's 32 channels into TWO lanes instead of four:
//   lane (n = lane & 31, h = lane >> 5); pixel tile p in {0, 1}; token t = 2 n + p (a lane's two tiles are horizontal
//   neighbours: ty = n >> 2, tx = 2 (n & 3) + p); accumulator register r of a tile holds channel 8 (r >> 2) + 4 h + (r & 3);
//   registers 8 s .. 8 s + 7 are the K-slots of K-step s (16 channels) of the next Linear (weights.hip: pack_frags32).
// A LayerNorm's statistics cross two lanes (two v_permlane32_swap per tile, two tiles) instead of four (three swaps per
// tile, four tiles), and are finalised twice per lane instead of four times.
#pragma once

constexpr int kS1C = 32;
#ifndef BALF_S1_SPLIT_MIX
#define BALF_S1_SPLIT_MIX 2   // operand split form of the stage-1 kernels (split16.h)
#endif
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16v mfma32(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Transposed token tile of a wave (32 channels x 64 tokens, hi and lo planes of 4 KB): rows of 128 B whose eight 16-B
// chunks are XOR-swizzled by the row, so that the 4-byte writes (a lane's two adjacent tokens of one channel) and the
// 16-byte A-fragment reads (eight tokens of a row, 32 rows at a time) both spread over the LDS banks without padding.
constexpr int kS1BtPlane = kS1C * 128;
constexpr int kS1BtBytes = 2 * kS1BtPlane;                     // 8192 B per wave
// byte offset of tokens 2 n, 2 n + 1 of channel row c (writer) / of tokens 8 j .. 8 j + 7 of row c (reader)
__device__ __forceinline__ int s1_bt_wr32(int c, int n) { return c * 128 + (((n >> 2) ^ (c & 7)) << 4) + (n & 3) * 4; }
__device__ __forceinline__ int s1_bt_rd(int c, int j) { return c * 128 + ((j ^ (c & 7)) << 4); }
// the 16x16 writer of the channel-split kernels (stage_cs_f16.h): tokens 4 li .. 4 li + 3 of row c
__device__ __forceinline__ int s1_bt_wr(int c, int li) { return c * 128 + (((li >> 1) ^ (c & 7)) << 4) + (li & 1) * 8; }
// LDS image: the GELU chord table at offset 0 (its byte offsets come straight out of a bit mask, see gelu_lut_off), then
// weight tiles (2 KiB each: [hi 64 x 16 B][lo 64 x 16 B], one per (32 output rows, 16 inputs)) ...
#ifndef BALF_GELU_LUT
#define BALF_GELU_LUT 1      // 1: GELU of the stage-1 kernels from the LDS chord table (3 vector instructions + 1 LDS read); 0: 2^P form (8)
#endif
constexpr int kS1LutBytes = BALF_GELU_LUT ? ((kGeluLutN + 1) * 8 + 15) / 16 * 16 : 0;
constexpr int kS1Conv0 = kS1LutBytes;                          // 64 lanes x 2 floats (built in the kernel from the plain [32,3] matrix)
constexpr int kS1Q1 = kS1Conv0 + 512;                          // 2 K-steps (this branch's half of RSHMAG.dense1)
constexpr int kS1D1 = kS1Q1 + 2 * 2048;                        // 2 row tiles (a half, b half) x 2 K-steps
constexpr int kS1Mix = kS1D1 + 4 * 2048;                       // 2 token tiles x 4 K-steps, rows in the lanes' token order
constexpr int kS1D2 = kS1Mix + 8 * 2048;                       // 2 K-steps
constexpr int kS1Q2 = kS1D2 + 2 * 2048;                        // block only: 4 K-steps (0, 1: u'; 2, 3: v')
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

// eight accumulator registers (the K-slots of one K-step) -> one B fragment
__device__ __forceinline__ HL s1_split8(const f16v &t, int s) {
    HL o;
    h2 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        split_pair<BALF_S1_SPLIT_MIX>(t[8 * s + 2 * i], t[8 * s + 2 * i + 1], h, l);
        o.hi[2 * i] = h[0]; o.hi[2 * i + 1] = h[1]; o.lo[2 * i] = l[0]; o.lo[2 * i + 1] = l[1];
    }
    return o;
}

// sum over the two lanes l, l ^ 32 of TWO values at once (pure VALU row swaps; the ds_bpermute form of __shfl_xor goes
// through the LDS crossbar and its latency)
__device__ __forceinline__ void half_allreduce2(float &s, float &ss) {
    // (the builtin pads the VALU-write -> permlane-read hazard itself; it mis-folds a swap whose two operands are the
    // same SSA value -- both results come out as one register -- hence the opaque copy)
    auto swap32 = [](unsigned a, unsigned b) { return __builtin_amdgcn_permlane32_swap(a, b, false, false); };
    auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
    auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
    const auto r0 = swap32(u(s), u(ss));                  // [s.lo ss.lo], [s.hi ss.hi]
    const float c = f(r0[0]) + f(r0[1]);                  // [S SS]
    unsigned c1 = u(c);
    asm("" : "+v"(c1));
    const auto r1 = swap32(u(c), c1);                     // [S S], [SS SS]
    s = f(r1[0]);
    ss = f(r1[1]);
}

// LayerNorm statistics of one pixel tile in one pass (sum and sum of squares; var = E[x^2] - mean^2: the inputs here are
// O(1) activations with |mean| of the order of the deviation, so the cancellation costs ~1e-6 relative on the variance)
__device__ __forceinline__ void ln_stats32(const f16v &x, float &rstd, float &shift) {
    constexpr float inv_c = 1.0f / 32;
    float s = x[0], ss = x[0] * x[0];                     // (not 0 + x: hipcc keeps that add -- it turns -0 into +0)
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        s += x[r];
        ss = fmaf(x[r], x[r], ss);
    }
    half_allreduce2(s, ss);
    const float mean = s * inv_c;
    const float var = fmaf(ss, inv_c, -mean * mean);
    rstd = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
    shift = -mean * rstd;
}

// (x - mean) * rstd, split into the B fragments of the next Linear (affine part folded into its weights): b[p][K-step]
__device__ __forceinline__ void s1_ln_split(const f16v (&x)[2], HL (&b)[2][2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float rstd, shift;
        ln_stats32(x[p], rstd, shift);
        f16v y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = fmaf(x[p][r], rstd, shift);
        b[p][0] = s1_split8(y, 0);
        b[p][1] = s1_split8(y, 1);
    }
}

__device__ __forceinline__ void s1_split(const f16v (&x)[2], HL (&b)[2][2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        b[p][0] = s1_split8(x[p], 0);
        b[p][1] = s1_split8(x[p], 1);
    }
}

// acc[p] (+)= W(32 output rows) . B[p]: KS K-steps of 16, weight fragments from the LDS image (wl already + lane * 16),
// K-step s at wl + s * 2048
template <int KS>
__device__ __forceinline__ void s1_linear(f16v (&acc)[2], const unsigned char *wl, const HL (&b)[2][KS]) {
    HL a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        a[s].hi = *reinterpret_cast<const h8 *>(wl + s * 2048);
        a[s].lo = *reinterpret_cast<const h8 *>(wl + s * 2048 + 1024);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) if (!BALF_DROP_WLO) acc[p] = mfma32(a[s].lo, b[p][s].hi, acc[p]);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[p] = mfma32(a[s].hi, b[p][s].lo, acc[p]);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc[p] = mfma32(a[s].hi, b[p][s].hi, acc[p]);
}

// accumulator start value: the bias of the lane's 16 channels (8 g + 4 h + r), the same for both tiles
__device__ __forceinline__ void s1_bias(f16v (&t)[2], const float *par, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4 b = *reinterpret_cast<const f4 *>(par + 8 * g + 4 * h);
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[0][4 * g + r] = b[r]; t[1][4 * g + r] = b[r]; }
    }
}

template <int N>
__device__ __forceinline__ void relu32(f16v (&t)[N]) {
#pragma unroll
    for (int p = 0; p < N; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[p][r] = max0(t[p][r]);
}

template <int N>
__device__ __forceinline__ void lrelu32(f16v (&t)[N]) {
#pragma unroll
    for (int p = 0; p < N; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = t[p][r];
            const float m = 0.2f * v;                     // (see lrelu in det_common.h: the result goes into v's own register)
            asm("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(m));
            t[p][r] = v;
        }
}

// GELU from the chord table in LDS (layout.h: kGeluLutN intervals over [-L, L), weights.hip builds it):
//   y = clamp01(x / 2L + 1/2)            v_fma_f32 ... clamp  (saturates to the two asymptote entries outside [-L, L))
//   t = y * 8N + 1.5 * 2^23              the rounded 8 N y lands in the low mantissa bits
//   (a, b) = table[bits(t) & 0x7FF8]     byte offset of interval floor(round(8 N y) / 8): one ds_read_b64
//   gelu = a + b x
// 3 vector instructions (9.4 issue cycles at the measured class rates) + 1 LDS read against 8 (33 cycles) for the 2^P
// form; the LDS pipe of these kernels was idle three quarters of the time.  Chord error 7.6e-7 (the 2^P form: 6.4e-7).
template <int LUTN = kGeluLutN>
__device__ __forceinline__ unsigned gelu_lut_off(float x, float magic) {
    static_assert((LUTN + 1) * 8 <= 0x8000, "the byte offset mask below is 15 bits");
    // (x is an MFMA result: the instruction that reads it must be the compiler's -- hipcc pads the MFMA -> VALU read
    // hazard for its own instructions only, an inline-asm v_fma placed right behind the MFMA read stale registers;
    // fmed3(., 0, 1) folds into the fma's clamp modifier)
    const float y = __builtin_amdgcn_fmed3f(fmaf(x, 0.5f / kGeluLutL, 0.5f), 0.0f, 1.0f);
    const float t = fmaf(y, 8.0f * LUTN, magic);
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
template <int CH, int LUTN = kGeluLutN>
__device__ __forceinline__ void gelu_lut_pipe(f16v (&t)[2], float magic) {
    constexpr int NCH = 32 / CH;
    typedef const f2 __attribute__((address_space(3))) *lds_f2_ptr;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = t[i >> 4][i & 15];
    auto fence = [] { __builtin_amdgcn_sched_barrier(0x8 | 0x4); };          // MFMA and SALU may cross, VALU and LDS not
    f2 ab[2][CH];
    auto issue = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < CH; ++i) ab[buf][i] = *reinterpret_cast<lds_f2_ptr>(gelu_lut_off<LUTN>(v[c * CH + i], magic));
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
    for (int i = 0; i < 32; ++i) t[i >> 4][i & 15] = v[i];
}

__device__ __forceinline__ void s1_gelu(f16v (&t)[2]) {
    if (BALF_ABLATE_GELU) return;
#if BALF_GELU_LUT
    float magic = 12582912.0f;                   // 1.5 * 2^23, kept in a vector register (the fma's other two operands use the constant bus)
    asm("" : "+v"(magic));
    gelu_lut_pipe<BALF_S1_GELU_CH>(t, magic);
#else
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[p][r] = gelu1<false>(t[p][r]);
#endif
}

// Vector-memory discipline of the float-input kernels (U8 = false).  vmcnt counts loads and stores together, in issue
// order, and the compiler drains it to zero at the loop's back edge as soon as it has a load of its own pending -- which
// makes every group wait for the previous group's 8-16 KiB of stores (measured: 40 % of the grid kernel's time).  So every
// LOAD of the loop is inline asm with a counted wait placed by hand (the compiler, seeing only stores, never waits):
//   top of group i:   wait for the input pixels of group i       (younger operations: the stores of group i-1 -> vmcnt(8))
//                     issue the u' rows of group i (block branch), then the input pixels of group i+1
//   before RSHMAG.dense2 (block): wait for the u' rows          (younger: the two input loads -> vmcnt(2))
// The loads of the last group's successor are issued anyway (clamped to a valid group) so that the counts are static.
// tools/vmcnt_audit.py checks all of this in the built code (tests/test_build_invariants.py).
// a wave-uniform pointer as a scalar-register pair (hipcc does 64-bit multiplies of uniform values on the vector unit
// and then hands the asm's "s" operand a VGPR pair)
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const T *>(((unsigned long long)hi << 32) | lo);
}

// pre-split activation row in HBM, 32x32 fragment format: per pixel and K-step of 16 channels 64 B = [hi: lane half 0,
// lane half 1 x 8 halves][lo: ...]; lane half h's eight halves are channels 16 s + 4 h + (0..3) and 16 s + 8 + 4 h + (0..3)
__device__ __forceinline__ HL load_frag32(const float *base, long pix, int C, int s, int h) {
    const char *p = reinterpret_cast<const char *>(base) + pix * (long)C * 4 + s * 64 + h * 16;
    HL o;
    o.hi = *reinterpret_cast<const h8 *>(p);
    o.lo = *reinterpret_cast<const h8 *>(p + 32);
    return o;
}
__device__ __forceinline__ void store_frag32(float *base, long pix, int C, int s, int h, const HL &v) {
    char *p = reinterpret_cast<char *>(base) + pix * (long)C * 4 + s * 64 + h * 16;
    *reinterpret_cast<h8 *>(p) = v.hi;
    *reinterpret_cast<h8 *>(p + 32) = v.lo;
}

// BALF_ABLATE_UWINDOW (diag.h): pixel index of u' wrapped into a window of that many MiB (wrong results, timing only)
__device__ __forceinline__ long uwin_pix(long pix, int C) {
#if BALF_ABLATE_UWINDOW
    return pix & ((((long)BALF_ABLATE_UWINDOW << 20) / (C * 4)) - 1);
#else
    return pix;
#endif
}

// The block kernel stores x1 and the channel sums of the RCAB's hidden layer only; the tail kernel (MODE 2) recomputes the
// RCAB branch from x1.
#ifndef BALF_S1_KEEP_X0
#define BALF_S1_KEEP_X0 1    // block kernel: keep x0 in registers (32) instead of recomputing it: -5 %
#endif
#if BALF_S1_STRICT
#define BALF_S1_WAIT_IN "s_waitcnt vmcnt(0)"
#define BALF_S1_WAIT_U "s_waitcnt vmcnt(0)"
#else
#define BALF_S1_WAIT_IN "s_waitcnt vmcnt(8)"
#define BALF_S1_WAIT_U "s_waitcnt vmcnt(2)"
#endif
// NEXT32: the stage's output (tail kernel) is written in the 32x32 fragment format (the consumer, stage 2, runs 32x32
// kernels: layout.h kFmt32[1]) or in the 16x16 one (per pixel and 32 channels 128 B = [hi: q0..q3 x 16 B][lo: ...])
template <int MODE, bool U8>
__global__ __launch_bounds__(s1_waves<MODE>() * 64, 1) void stage1_kernel16(StageArgs A) {
    constexpr int C = kS1C, P = 2, NW = s1_waves<MODE>(), NTHR = NW * 64;
    constexpr int BM = MODE == 0 ? 0 : 1;                        // branch whose weights / token geometry this kernel uses
    constexpr bool TAIL = MODE == 2;                             // the stage's tail (see the loop body)
    constexpr bool NEXT32 = kFmt32[1];
    constexpr int STAMP_KID = BM; (void)STAMP_KID;
    STAMP_DECL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *par = reinterpret_cast<float *>(smem_raw + s1_weight_bytes<MODE>());
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
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
        copy(kS1Q1, S.q1_w + BM * (2 * 512), 2 * 2048);        // output rows BM*C ..: row tile BM, K-steps 0, 1 (512 floats each)
        copy(kS1D1, Br.d1_w, 4 * 2048);                          // (row tile, K-step): 2 R + s
        copy(kS1D2, Br.d2_w, 2 * 2048);
        if (!TAIL) copy(kS1Mix, Br.mix_w, 8 * 2048);            // (token tile p', K-step): 4 p' + s, rows already in lane order
        if (BM == 1) {
            copy(kS1Q2, S.q2_w, 4 * 2048);
            copy(kS1R1, S.r1_w, 2 * 2048);
            copy(kS1R2, S.r2_w, 2 * 2048);
        }
        // conv0 [32, 3] as A operands of two v_mfma_f32_32x32x2_f32 (exact fp32, K = 3 padded to 4): lane (m, h) holds
        // W[m][h] for the first and W[m][2] (h = 0) / 0 (h = 1) for the second
        for (int i = threadIdx.x; i < 64; i += NTHR) {
            const int m = i & 31, hh = i >> 5;
            *reinterpret_cast<float2 *>(smem_raw + kS1Conv0 + i * 8) =
                make_float2(blob[S.conv0_w + m * 3 + hh], hh == 0 ? blob[S.conv0_w + m * 3 + 2] : 0.0f);
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
    const int ty_ = n >> 2, tx0_ = 2 * (n & 3);
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
        int ty = ty_, tx0 = tx0_;
        if constexpr (MODE == 2) {
            // (tail, three waves per SIMD = 168 registers: two instructions per group instead of two registers held across
            // the loop -- hipcc otherwise spills one of them and reloads it from scratch in every iteration)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            ty = (ln & 31) >> 2;
            tx0 = 2 * (ln & 3);
        }
        if (MODE == 0) { g.y = ty * fh + c.gy; g.x0 = tx0 * fw + c.gx; }
        else           { g.y = 8 * c.gy + ty;  g.x0 = 8 * c.gx + tx0; }
        g.pix0 = ((long)g.n * H + g.y) * W + g.x0;
        return g;
    };
    // raw network input of the lane's two pixels, conv0's B operands: raw[p] = input channel h (first MFMA: k = h),
    // raw[2 + p] = input channel 2 (second MFMA: k = 0; its k = 1 slot meets a zero weight, any finite value will do) --
    // float bits, or (U8) the uint8 value, 0x100 = outside the image
    auto load_raw_u8 = [&](const Geo &g, unsigned (&raw)[4]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int yy = g.y - A.u8_top, xx = g.x0 + p * pstep - A.u8_left;
            const bool ok = yy >= 0 && yy < A.u8_h && xx >= 0 && xx < A.u8_w;
            const unsigned char *px8 = A.X8 + (((long)g.n * A.u8_h + (ok ? yy : 0)) * A.u8_w + (ok ? xx : 0)) * A.u8_ch;
            raw[p] = ok ? (unsigned)px8[A.u8_ch == 3 ? h : 0] : 0x100u;
            raw[2 + p] = ok ? (unsigned)px8[A.u8_ch == 3 ? 2 : 0] : 0x100u;
        }
    };
    const int hw = H * W;                                       // (32-bit: keeps the plane offsets on the scalar unit)
    // float input, asm loads (not counted by the compiler): the image's planes at a scalar base, the lane's plane and pixel
    // offset in a VGPR.  The prefetched pixels land in registers of their own (nraw), and ONE statement waits for them and
    // copies them into the registers the group computes from (take_raw: the copies sit behind the s_waitcnt inside the
    // statement).  Round 2 loaded into the compute variable's own register ("+v") and waited in a second statement whose
    // operand was tied to the same variable: the variable is live across the loop, so hipcc had to copy the load's
    // destination into the loop-carried register IN FRONT of the wait, a whole iteration after the load -- correct only as
    // long as a load never takes longer than an iteration (tools/vmcnt_audit.py).
    const unsigned offA = (unsigned)(h * hw) * 4u, offB = (unsigned)(2 * hw) * 4u;
    auto issue_raw = [&](const Geo &g, unsigned (&nraw)[4]) {
        const unsigned vo = (unsigned)(g.y * W + g.x0) * 4u;
        const unsigned voA = vo + offA, voB = vo + offB;
        const float *xb = uniform_ptr(A.X + (long)g.n * 3 * (long)hw);
        if constexpr (BM == 1) {                                 // the two pixels are adjacent (x0 even): 8-byte loads
            unsigned long long a, b;
            asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %4\n\tglobal_load_dwordx2 %1, %3, %4"
                         : "=&v"(a), "=&v"(b) : "v"(voA), "v"(voB), "s"(xb) : "memory");
            nraw[0] = (unsigned)a; nraw[1] = (unsigned)(a >> 32); nraw[2] = (unsigned)b; nraw[3] = (unsigned)(b >> 32);
        } else {
            const unsigned st = (unsigned)pstep * 4u;
            const unsigned voA1 = voA + st, voB1 = voB + st;
            asm volatile("s_nop 4\n\tglobal_load_dword %0, %4, %8\n\tglobal_load_dword %1, %5, %8\n\t"
                         "global_load_dword %2, %6, %8\n\tglobal_load_dword %3, %7, %8"
                         : "=&v"(nraw[0]), "=&v"(nraw[1]), "=&v"(nraw[2]), "=&v"(nraw[3])
                         : "v"(voA), "v"(voA1), "v"(voB), "v"(voB1), "s"(xb) : "memory");
        }
    };
    // wait, then raw <- nraw.  "+v" on raw: the copies write registers that hold the previous group's pixels, live
    // VALU-visible values -- never a register an MFMA in flight still reads (split16.h).
    auto take_raw = [&](unsigned (&raw)[4], const unsigned (&nraw)[4]) {
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
    unsigned raw[4] = {}, nraw[4] = {};
    if (!U8 && !TAIL && item < total) {
        issue_raw(geo(nxt), nraw);
        // the first group has no older stores in front of its pixels: drain here (the counted wait in the loop assumes them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float2 a0 = *reinterpret_cast<const float2 *>(smem_raw + kS1Conv0 + lane * 8);   // conv0's A operands (loop-invariant)

    unsigned long long sat = 0;                                  // (tail) lanes that saw a stage output at or beyond the largest f16
    for (; item < total; item += stride) {
        const Geo g = geo(nxt);
        if (item + stride < total) nxt = advance(nxt);           // (the last group re-requests its own pixels)
        const long pix0 = g.pix0;
        if (U8 && !TAIL) load_raw_u8(g, raw);                  // (tail: right before its conv0, see below)
        // (tail) plain loads, no prefetch across groups: three waves per SIMD cover the latency, and with few stores per
        // group there is no store queue to count around.  x1 as the block kernel left it, this image's SE scale.
        f16v x1t[TAIL ? 2 : 1];
        f4 sct[4];
        if constexpr (TAIL) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const f4 v = *reinterpret_cast<const f4 *>(A.R + (long)item * (64 * C) + ((p * 4 + gq) * 64 + lane) * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) x1t[p][4 * gq + r] = v[r];
                }
            }
        }
        STAMP(0);
        float bx[4];                                             // conv0's B operands
        auto make_bx = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (U8) bx[i] = raw[i] < 256u ? par[kS1pLut + (raw[i] & 255u)] : 0.0f;
                else bx[i] = __builtin_bit_cast(float, raw[i]);
            }
        };
        if constexpr (!TAIL) {
            if constexpr (!U8) {
                // the input pixels of this group have landed; the previous group's stores may still be in flight
                take_raw(raw, nraw);
            }
            make_bx();
        }
        HL ub[(MODE == 1) ? P : 1][2];                           // block: u' rows of the lane's pixels (pre-split in HBM)
        if constexpr (MODE == 1) {
            if constexpr (U8) {
#pragma unroll
                for (int p = 0; p < P; ++p) { ub[p][0] = load_frag32(A.U, pix0 + p * pstep, C, 0, h); ub[p][1] = load_frag32(A.U, pix0 + p * pstep, C, 1, h); }
            } else {
#if BALF_ABLATE_UWINDOW
                const unsigned uoff = (unsigned)uwin_pix((long)g.n * hw + g.y * W + g.x0, 32) * 128u + h * 16u;
                const char *ubase = uniform_ptr(reinterpret_cast<const char *>(A.U));
#else
                const unsigned uoff = (unsigned)(g.y * W + g.x0) * 128u + h * 16u;
                const char *ubase = uniform_ptr(reinterpret_cast<const char *>(A.U) + (long)g.n * (long)hw * 128);
#endif
#define BALF_S1_UB(PI, SI)                                                                                          \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 offset:%c4\n\tglobal_load_dwordx4 %1, %2, %3 offset:%c5"        \
                 : "=&v"(ub[PI][SI].hi), "=&v"(ub[PI][SI].lo) : "v"(uoff), "s"(ubase), "i"(PI * 128 + SI * 64), "i"(PI * 128 + SI * 64 + 32) : "memory")
                BALF_S1_UB(0, 0); BALF_S1_UB(0, 1); BALF_S1_UB(1, 0); BALF_S1_UB(1, 1);
#undef BALF_S1_UB
            }
        }
        if constexpr (!U8 && !TAIL) issue_raw(geo(nxt), nraw);   // next group's pixels (2 or 4 loads, always)
        STAMP(1);   // input -> conv0 B operands (waits for the prefetched pixels), next group's loads issued
        auto conv0 = [&](f16v (&x0v)[2]) {                       // x0 = relu(conv0(X)); bit-identical every time
            s1_bias(x0v, par + kS1pConv0B, h);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x0v[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bx[p], x0v[p], 0, 0, 0);
                x0v[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bx[2 + p], x0v[p], 0, 0, 0);
            }
            relu32(x0v);
        };

        HL b[P][2];                                              // B fragments of the next Linear: [tile][K-step]
        if constexpr (TAIL) {
            // ---- the stage's tail: x2 = maxpool2x2(x1 + x0 + s * conv2(lrelu(conv1(LN(x1))))) in fragment format ----
            // The block kernel stored x1 and the channel sums of the RCAB's hidden layer (conv2 is linear: the SE
            // kernel gets mean(t) from mean(hidden)); the branch itself is recomputed here from x1 -- the same
            // instructions on the same values as the block kernel's, so t is the t it would have stored -- instead of
            // a T and an R tensor travelling through HBM (4.3 GB per 8 images at 1088x1920 written and read again).
            s1_ln_split(x1t, b);
            f16v m1[2];
            s1_bias(m1, par + kS1pR1B, h);
            s1_linear<2>(m1, wl + kS1R1, b);
            lrelu32(m1);
            s1_split(m1, b);
            f16v t[2];
            s1_bias(t, par + kS1pR2B, h);
            s1_linear<2>(t, wl + kS1R2, b);
            // the raw pixels likewise (plain loads; the tail has no prefetch across groups)
            if constexpr (U8) load_raw_u8(g, raw);
            else {
                const float *xp = A.X + (long)g.n * 3 * (long)hw + (long)g.y * W + g.x0;
                const float2 va = *reinterpret_cast<const float2 *>(xp + (long)h * hw), vb = *reinterpret_cast<const float2 *>(xp + 2L * hw);
                raw[0] = __builtin_bit_cast(unsigned, va.x); raw[1] = __builtin_bit_cast(unsigned, va.y);
                raw[2] = __builtin_bit_cast(unsigned, vb.x); raw[3] = __builtin_bit_cast(unsigned, vb.y);
            }
            make_bx();
            f16v x0v[2];
            conv0(x0v);
            // r = x1 + x0 first (x1's 32 registers die here), then the image's SE scale -- requested here, not at the top: at
            // three waves per SIMD the kernel otherwise spills; an L2 hit that the other waves cover
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) x0v[p][r] = x1t[p][r] + x0v[p][r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) sct[gq] = *reinterpret_cast<const f4 *>(A.scale + (long)g.n * C + 8 * gq + 4 * h);
            // v = r + s t with r = x1 + x0, max over the 2x2 window: the lane's two tiles are horizontal neighbours, the
            // rows 2 j, 2 j + 1 sit in lanes n, n + 4
            f16v mx;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = sct[r >> 2][r & 3];
                const float v0 = fmaf(t[0][r], sc, x0v[0][r]);
                const float v1 = fmaf(t[1][r], sc, x0v[1][r]);
                const float m = __builtin_fmaxf(v0, v1);
                // the vertical partner's value in EVERY lane: lane n + 4's by row_shl:4, then -- into lane banks 1 and 3 only
                // (bank_mask 0xA) -- lane n - 4's by row_shr:4.  (A first form copied the finished maximum from lane n to
                // lane n + 4 with update_dpp(old = x, src = x): hipcc treated the sixteen results as one value and deleted
                // fifteen of the sixteen maxima -- caught only because the wide-range test does not fall back to fp32.)
                const int mi = __builtin_bit_cast(int, m);
                const int up = __builtin_amdgcn_update_dpp(0, mi, 0x104 /* row_shl:4 */, 0xF, 0xF, false);
                const int pv = __builtin_amdgcn_update_dpp(up, mi, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
                mx[r] = __builtin_fmaxf(m, __builtin_bit_cast(float, pv));
            }
            // the stage's output, about to be split: status block.  The verdict lives in SCALAR registers (a lane mask per
            // comparison, OR-ed): a running maximum in a vector register is one more than this kernel has at three waves per SIMD
#pragma unroll
            for (int r = 0; r < 16; ++r) sat |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(mx[r]) >= kF16Max);
            // lanes n and n + 4 (n & 4 == 0) both hold pooled pixel (ty / 2, n & 3) of the 4 x 4 output block, so the FOUR lanes
            // of a pooled pixel -- two lane halves x two partners -- store its 128-byte row as two instructions of 64
            // contiguous bytes each (hi, lo), like the four lane quarters of the 16x16 layout did.  (Stored by the lanes n
            // alone -- four 16-byte pieces each, 32 contiguous bytes per pixel and instruction -- the stage-2 grid kernel that
            // reads this tensor next ran 5 % slower.)
            {
                const long opix = ((long)g.n * (H / 2) + (g.y >> 1)) * (W / 2) + (g.x0 >> 1);
                const HL s0 = s1_split8(mx, 0), s1 = s1_split8(mx, 1);
                const int part = (n >> 2) & 1;                   // 0: the lane n of the pair, 1: its partner n + 4
                const unsigned pm = 0u - (unsigned)part;
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                auto pick = [&](const h8 &a, const h8 &b) {     // partner ? b : a, one v_bfi per register
                    const u4 ua = __builtin_bit_cast(u4, a), ub = __builtin_bit_cast(u4, b);
                    u4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (ub[i] & pm) | (ua[i] & ~pm);
                    return __builtin_bit_cast(h8, o);
                };
                if constexpr (NEXT32) {
                    // 32x32 fragment format: K-step s at + 64 s: [hi: h0, h1][lo: h0, h1]; lane n stores K-step 0, its partner K-step 1
                    const HL v{pick(s0.hi, s1.hi), pick(s0.lo, s1.lo)};
                    store_frag32(A.out, opix, C, part, h, v);
                } else {
                    // 16x16 fragment format: lane quarter q's 16 bytes are channels 4 q + (0..3) and 16 + 4 q + (0..3); this
                    // lane holds channels 8 g + 4 h + (0..3): quarter h <- (g = 0, g = 2), quarter 2 + h <- (g = 1, g = 3);
                    // lane n stores quarter h, its partner quarter 2 + h
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                    auto lo4 = [](const h8 &v) { return h4{v[0], v[1], v[2], v[3]}; };
                    auto hi4 = [](const h8 &v) { return h4{v[4], v[5], v[6], v[7]}; };
                    auto cat = [](const h4 &a, const h4 &c) { return h8{a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]}; };
                    char *op = reinterpret_cast<char *>(A.out) + opix * (long)C * 4 + (h + 2 * part) * 16;
                    *reinterpret_cast<h8 *>(op) = pick(cat(lo4(s0.hi), lo4(s1.hi)), cat(hi4(s0.hi), hi4(s1.hi)));
                    *reinterpret_cast<h8 *>(op + 64) = pick(cat(lo4(s0.lo), lo4(s1.lo)), cat(hi4(s0.lo), hi4(s1.lo)));
                }
            }
        } else {
        f16v x0k[(MODE == 1 && BALF_S1_KEEP_X0) ? 2 : 1];       // (block) x0, kept for x1 = . + x0
        if constexpr (MODE == 1 && BALF_S1_KEEP_X0) {
            conv0(x0k);
            s1_ln_split(x0k, b);
        } else {
            f16v x0v[2];
            conv0(x0v);
            s1_ln_split(x0v, b);
        }
        STAMP(2);   // conv0 + relu + LN + split
        f16v z[2];                                               // u (grid) / v (block): kept for the branch residual
        s1_bias(z, par + kS1pQ1B, h);
        s1_linear<2>(z, wl + kS1Q1, b);
        s1_gelu(z);
        STAMP(3);   // dense1 half + GELU
        s1_ln_split(z, b);
        STAMP(4);   // LN + split

        f16v ga[2];
        s1_bias(ga, par + kS1pD1B, h);
        s1_linear<2>(ga, wl + kS1D1, b);
        s1_gelu(ga);
        STAMP(5);   // branch dense1 (a half) + GELU
        {
            f16v gb[2];
            s1_bias(gb, par + kS1pD1B + C, h);
            s1_linear<2>(gb, wl + kS1D1 + 2 * 2048, b);
            s1_gelu(gb);
            // gating LayerNorm (affine) -> transposed token tile bT[hi|lo][c][t], t = 2 n + p: 4-byte stores
            float rstd[P], shift[P];
#pragma unroll
            for (int p = 0; p < P; ++p) ln_stats32(gb[p], rstd[p], shift[p]);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f4 gg = *reinterpret_cast<const f4 *>(par + kS1pGlnG + 8 * gq + 4 * h);
                const f4 bb = *reinterpret_cast<const f4 *>(par + kS1pGlnB + 8 * gq + 4 * h);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = fmaf(fmaf(gb[0][4 * gq + r], rstd[0], shift[0]), gg[r], bb[r]);
                    const float v1 = fmaf(fmaf(gb[1][4 * gq + r], rstd[1], shift[1]), gg[r], bb[r]);
                    h2 hh, ll;
                    split_pair<BALF_S1_SPLIT_MIX>(v0, v1, hh, ll);
                    unsigned char *row = bT + s1_bt_wr32(8 * gq + 4 * h + r, n);
                    *reinterpret_cast<h2 *>(row) = hh;
                    *reinterpret_cast<h2 *>(row + kS1BtPlane) = ll;
                }
            }
        }
        STAMP(6);   // branch dense1 (b half) + GELU + LN + transposed tile
        {
            // mix^T[c][t'] = sum_t bT[c][t] Wmix[t'][t] (+ bias[t'] + 1 as the accumulator's start value), then the gate
            HL a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned char *row = bT + s1_bt_rd(n, 2 * s + h);
                a[s].hi = *reinterpret_cast<const h8 *>(row);
                a[s].lo = *reinterpret_cast<const h8 *>(row + kS1BtPlane);
            }
            const float2 mb1 = *reinterpret_cast<const float2 *>(par + kS1pMixB1 + 2 * n);
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const float mb = pt ? mb1.y : mb1.x;
                f16v m;
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = mb;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    HL w;
                    w.hi = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 4 + s) * 2048);
                    w.lo = *reinterpret_cast<const h8 *>(wl + kS1Mix + (pt * 4 + s) * 2048 + 1024);
                    m = mfma32(a[s].lo, w.hi, m);
                    m = mfma32(a[s].hi, w.lo, m);
                    m = mfma32(a[s].hi, w.hi, m);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ga[pt][r] *= m[r];
            }
        }
        STAMP(7);   // token mix + gate
        s1_split(ga, b);
        f16v o[2];
        s1_bias(o, par + kS1pD2B, h);
        s1_linear<2>(o, wl + kS1D2, b);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[p][r] += z[p][r];

        STAMP(8);   // dense2 + residual
        if constexpr (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                store_frag32(A.U, uwin_pix(pix0 + p * pstep, C), C, 0, h, s1_split8(o[p], 0));
                store_frag32(A.U, uwin_pix(pix0 + p * pstep, C), C, 1, h, s1_split8(o[p], 1));
            }
            STAMP(9);   // u' store
        } else {
            s1_split(o, b);
            f16v x1[2];
            s1_bias(x1, par + kS1pQ2B, h);
            s1_linear<2>(x1, wl + kS1Q2 + 2 * 2048, b);          // K-steps 2, 3 = v' half of cat[u', v']
            if constexpr (!U8)                                   // u' rows have landed (younger: next group's two input loads)
                asm volatile(BALF_S1_WAIT_U
                             : "+v"(ub[0][0].hi), "+v"(ub[0][0].lo), "+v"(ub[0][1].hi), "+v"(ub[0][1].lo), "+v"(ub[1][0].hi),
                               "+v"(ub[1][0].lo), "+v"(ub[1][1].hi), "+v"(ub[1][1].lo) :: "memory");
            s1_linear<2>(x1, wl + kS1Q2, ub);                    // K-steps 0, 1 = u' half
            STAMP(9);   // RSHMAG dense2 over cat[u', v'] (u' from HBM)
            {
#if BALF_S1_KEEP_X0
                f16v (&x0v)[2] = x0k;
#else
                f16v x0v[2];
                conv0(x0v);                                      // recomputed instead of kept (32 registers)
#endif
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x1[p][r] += x0v[p][r];
                    // x1 itself: the tail kernel adds x0 and the scaled RCAB branch.  Only the tail kernel reads it, with the
                    // same group and lane geometry, so it is stored in REGISTER ORDER -- per group 8 KB = [tile][register
                    // quad][lane] x 16 B: every store (and the tail's loads) moves 1 KiB of contiguous memory.  (In pixel-major
                    // order a 32x32 lane pair covers 32 B of a pixel's row per instruction, half of what the four lanes of
                    // the 16x16 layout did: twice the cache-line touches; measured on the tail kernel: 0.59 -> 0.71 ms.)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<f4 *>(A.R + (long)item * (64 * C) + ((p * 4 + gq) * 64 + lane) * 4) =
                            f4{x1[p][4 * gq], x1[p][4 * gq + 1], x1[p][4 * gq + 2], x1[p][4 * gq + 3]};
                }
            }
            STAMP(10);  // residuals, R store
            s1_ln_split(x1, b);
            f16v m1[2];
            s1_bias(m1, par + kS1pR1B, h);
            s1_linear<2>(m1, wl + kS1R1, b);
            lrelu32(m1);
            // conv2 is linear: its channel means follow from the means of its input (the hidden layer m1), and the tail
            // kernel recomputes the RCAB branch from x1 -- no T tensor through HBM.
            // Channel sums over the group's 64 pixels (fixed order): the two tiles, then the two 16-lane rows of a lane half
            // as a reduce-scatter (one row swap serves registers r and r + 8: rows 0 / 2 end up with register r of lane half
            // 0 / 1, rows 1 / 3 with register r + 8), then the 16 lanes of the row; lane 0 of each row stores 8 channels.
            {
                float cs[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float sa = m1[0][r] + m1[1][r], sb = m1[0][r + 8] + m1[1][r + 8];
                    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sa), __builtin_bit_cast(unsigned, sb), false, false);
                    // (hipcc folded sw[1] into sw[0] here -- the sum came out as 2 * sw[0], seen in the assembly and as SE
                    // scales that were off by a few per cent; the opaque pass-through keeps the two results apart)
                    unsigned w0 = sw[0], w1 = sw[1];
                    asm("" : "+v"(w0), "+v"(w1));
                    float s = __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
                    cs[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s))));
                }
                if ((lane & 15) == 0) {
                    // row 0: (h 0, regs 0-7): channels 0-3, 8-11; row 1: (h 0, regs 8-15): 16-19, 24-27; row 2: (h 1, regs 0-7):
                    // 4-7, 12-15; row 3: (h 1, regs 8-15): 20-23, 28-31
                    const int row = lane >> 4, c0 = 16 * (row & 1) + 4 * (row >> 1);
                    float *pp = A.partial + (long)item * C + c0;
                    *reinterpret_cast<f4 *>(pp) = f4{cs[0], cs[1], cs[2], cs[3]};
                    *reinterpret_cast<f4 *>(pp + 8) = f4{cs[4], cs[5], cs[6], cs[7]};
                }
            }
            STAMP(12);  // channel sums of the hidden layer
        }
        }   // !TAIL
    }
    if (TAIL && sat != 0) status_raise(A.status, 1 /* BALF_STATUS_RANGE */);
}
