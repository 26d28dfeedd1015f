// Stage 2 (C = 64, Cin = 32) of the split-f16 detector forward: persistent kernels with ALL weights of the branch resident
// in LDS (the block kernel streams one Linear), in which a PAIR OF WAVES owns a token group and every wave owns 32 of its
// 64 tokens with all 64 channels ("token split"), on v_mfma_f32_32x32x16_f16.  Included by detector_f16.hip inside
// balf::{anonymous}, after stage1_f16.h whose helpers it shares.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 64.
//
// Why (round 4).  Rounds 2-3 ran this stage on the channel-split kernels (stage_cs_f16.h): two waves per token group, each
// owning 32 CHANNELS of all 64 tokens.  That form pays 77 vector instructions per value where stage 1 pays 53: every Linear's
// input is published through LDS as shared B fragments (split, store, barrier, re-read), LayerNorm statistics cross the waves
// through an LDS table behind a barrier (7 / 13 barriers per group), the weights stream from L2 for every group, a workgroup
// lives for one group (table copy, input latency exposed), and the 16x16 tiles spread a pixel's channels over four lanes.
// With the TOKENS split instead, a wave holds complete pixels: lane (n = lane & 31, h = lane >> 5) carries token
// t = 32 w + n of the group (w = the wave's half; ty = t >> 3, tx = t & 7) and channel 32 rt + 8 (r >> 2) + 4 h + (r & 3) in
// register r of accumulator tile rt -- the stage-1 layout with "two row tiles of one pixel tile" in the place of "one row tile
// of two pixel tiles".  So, as in stage 1:
//   * every Linear's B operand comes straight out of the accumulator registers (split in place), LayerNorm statistics cross
//     ONE lane pair, nothing is published and there is NO s_barrier in the main loop;
//   * the weights are staged once per workgroup (88 KB grid / 104 KB block / 40 KB tail as split-f16 fragments) and the
//     workgroup is persistent: one per CU, every wave pair walks its own list of groups;
//   * GELU comes from the uniform chord table at LDS address 0 (3 vector instructions + 1 LDS read).
// The one thing a wave cannot do alone is the 64x64 token mix (a sum over ALL tokens of a channel).  The pair exchanges the
// gating LayerNorm's output through a transposed token tile in LDS, one 32-channel row tile at a time (8 KB per pair: four
// pairs + weights + table do not fit with 16 KB each): both waves write their 32 tokens of the 32 channels, each reads all 64
// tokens back as MFMA A fragments and mixes for its own 32 output tokens.  Synchronisation is between the TWO waves only,
// through monotonic counters in LDS (s_barrier would stop all eight waves of the workgroup): a wave posts "written" /
// "read" after its LDS instructions (which execute in issue order) and polls its partner's counter.
//
// LDS budget (160 KB): grid 24.0 (table, 3072 intervals) + 88 + 2.8 + 32 = 146.9 KB; block 16.0 (2048 intervals) + 104 + 2.8
// + 32 = 154.9 KB with RSHMAG.dense2 (32 KB) streamed from L2 into registers per group (asm loads, counted waits).
#pragma once

constexpr int kS2C = 64, kS2Cin = 32;                             // a Linear with C inputs is 4 K-steps of 16 channels, conv0 2

#ifndef BALF_S2_NW0
#define BALF_S2_NW0 8
#endif
#ifndef BALF_S2_NW1
#define BALF_S2_NW1 8
#endif
#ifndef BALF_S2_NW2
#define BALF_S2_NW2 8
#endif
template <int MODE> constexpr int s2_waves() { return MODE == 0 ? BALF_S2_NW0 : MODE == 1 ? BALF_S2_NW1 : BALF_S2_NW2; }
template <int MODE> constexpr int s2_lut_n() { return MODE == 0 ? kGeluLutN : MODE == 1 ? kGeluLut2N : 0; }
template <int MODE> constexpr int s2_lut_bytes() { return s2_lut_n<MODE>() ? ((s2_lut_n<MODE>() + 1) * 8 + 15) / 16 * 16 : 0; }

// per-channel parameters in LDS (floats)
enum S2Par { kS2pConv0B = 0, kS2pQ1B = 64, kS2pD1B = 128, kS2pGlnG = 256, kS2pGlnB = 320, kS2pMixB1 = 384, kS2pD2B = 448,
             kS2pQ2B = 512, kS2pR1B = 576, kS2pR2B = 640, kS2ParFloats = 704 };

// LDS image: [chord table][weight tiles of 2 KiB: (32 output rows, 16 inputs) as [hi 64 x 16 B][lo 64 x 16 B]][parameters]
// [pair counters][one transposed token tile per wave pair]
template <int MODE> struct S2Map {
    static constexpr int conv0 = s2_lut_bytes<MODE>();           // 2 row tiles x 2 K-steps
    static constexpr int q1 = conv0 + 4 * 2048;                  // this branch's half of RSHMAG.dense1: 2 x 4   | tail: RCAB.conv1
    static constexpr int d1 = q1 + 8 * 2048;                     // branch dense1: 4 row tiles (a a b b) x 4      | tail: RCAB.conv2
    static constexpr int mix = d1 + 16 * 2048;                   // token mix: output-token tile (= wave half) x 4 K-steps
    static constexpr int d2 = mix + 8 * 2048;
    static constexpr int r1 = d2 + 8 * 2048;                     // block only: RCAB.conv1
    static constexpr int wend = MODE == 0 ? r1 : MODE == 1 ? r1 + 8 * 2048 : d1 + 8 * 2048;
    static constexpr int par = wend;
    static constexpr int flags = par + kS2ParFloats * 4;         // per wave: (written, read) pass counters
    static constexpr int tiles = flags + 64;
    static constexpr int total = tiles + (MODE == 2 ? 0 : (s2_waves<MODE>() / 2) * kS1BtBytes);
};
template <int MODE> constexpr int s2_lds_bytes() { return S2Map<MODE>::total; }

// accumulator start value: the bias of the lane's 32 channels (32 rt + 8 g + 4 h + r)
__device__ __forceinline__ void s2_bias(f16v (&t)[2], const float *par, int h) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4 b = *reinterpret_cast<const f4 *>(par + 32 * rt + 8 * g + 4 * h);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[rt][4 * g + r] = b[r];
        }
}

// LayerNorm statistics of the lane's pixel over its 64 channels (32 registers here, 32 in the partner lane l ^ 32)
__device__ __forceinline__ void s2_ln_stats(const f16v (&x)[2], float &rstd, float &shift) {
    constexpr float inv_c = 1.0f / kS2C;
    float s = x[0][0], ss = x[0][0] * x[0][0];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = (rt == 0 ? 1 : 0); r < 16; ++r) {
            s += x[rt][r];
            ss = fmaf(x[rt][r], x[rt][r], ss);
        }
    half_allreduce2(s, ss);
    const float mean = s * inv_c;
    const float var = fmaf(ss, inv_c, -mean * mean);
    rstd = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
    shift = -mean * rstd;
}

// the B fragments of a Linear with 64 inputs: K-step s = registers 8 (s & 1) .. + 7 of tile s >> 1
__device__ __forceinline__ void s2_split(const f16v (&x)[2], HL (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = s1_split8(x[s >> 1], s & 1);
}
__device__ __forceinline__ void s2_ln_split(const f16v (&x)[2], HL (&b)[4]) {
    float rstd, shift;
    s2_ln_stats(x, rstd, shift);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f16v y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = fmaf(x[rt][r], rstd, shift);
        b[2 * rt] = s1_split8(y, 0);
        b[2 * rt + 1] = s1_split8(y, 1);
    }
}

// acc[rt] += W(row tile rt) . B over KS K-steps; weight fragments from the LDS image: tile (rt, s) at wl + (rt * KS + s) * 2048
// (wl already + lane * 16)
template <int KS>
__device__ __forceinline__ void s2_linear_rt(f16v &acc, const unsigned char *wl, const HL (&b)[KS]) {
    HL a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        a[s].hi = *reinterpret_cast<const h8 *>(wl + s * 2048);
        a[s].lo = *reinterpret_cast<const h8 *>(wl + s * 2048 + 1024);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) if (!BALF_DROP_WLO) acc = mfma32(a[s].lo, b[s].hi, acc);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma32(a[s].hi, b[s].lo, acc);
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma32(a[s].hi, b[s].hi, acc);
}
template <int KS>
__device__ __forceinline__ void s2_linear(f16v (&acc)[2], const unsigned char *wl, const HL (&b)[KS]) {
    s2_linear_rt<KS>(acc[0], wl, b);
    s2_linear_rt<KS>(acc[1], wl + KS * 2048, b);
}
// the same with the weight fragments already in registers (the streamed Linear of the block kernel)
__device__ __forceinline__ void s2_linear_regs(f16v &acc, const HL (&a)[4], const HL (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) if (!BALF_DROP_WLO) acc = mfma32(a[s].lo, b[s].hi, acc);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(a[s].hi, b[s].lo, acc);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(a[s].hi, b[s].hi, acc);
}

#ifndef BALF_S2_GELU_CH
#define BALF_S2_GELU_CH 8    // table reads in flight per chunk of the software pipeline (stage1_f16.h: gelu_lut_pipe)
#endif
template <int MODE>
__device__ __forceinline__ void s2_gelu(f16v (&t)[2]) {
    if (BALF_ABLATE_GELU) return;
    float magic = 12582912.0f;                   // 1.5 * 2^23 (see s1_gelu)
    asm("" : "+v"(magic));
    gelu_lut_pipe<BALF_S2_GELU_CH, s2_lut_n<MODE>()>(t, magic);
}

// ---- pair synchronisation through LDS counters (raw LDS addresses: these kernels have no static LDS) ----
__device__ __forceinline__ void s2_post(unsigned addr, unsigned value) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(value) : "memory");
}
// INVARIANT the poll relies on: both waves of a pair run the SAME sequence of posts and polls -- `item`, `stride`, `pass` and the
// trip count of the group loop are pair-uniform (they depend on blockIdx and wave >> 1 only), and no wave leaves the loop or
// skips a pass on its own.  A future per-wave early-out would park its partner here for ever: keep every exit pair-uniform.
// A wave that finds the counter not there yet sleeps 64 cycles before it looks again, so that its spin does not take issue
// slots from the partner (which may sit on the same SIMD) it is waiting for.
__device__ __forceinline__ void s2_poll(unsigned addr, unsigned target) {
    unsigned v;
    for (;;) {
        // (not a vector-ALU instruction: the LDS return writes v long after any MFMA in flight has read its operands)
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        v = __builtin_amdgcn_readfirstlane(v);
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);          // (A/B on one box: 4.55-4.57 ms per 32 images with it, 4.53-4.56 without)
    }
}

#if BALF_S1_STRICT
#define BALF_S2_WAIT(n) "s_waitcnt vmcnt(0)"
#else
#define BALF_S2_WAIT(n) "s_waitcnt vmcnt(" #n ")"
#endif

// Vector-memory discipline, as in stage 1: vmcnt counts loads and stores together in issue order
// and hipcc drains it at the loop's back edge as soon as a load of its own is pending, so every LOAD of the loop is inline asm
// with a hand-counted wait, the compiler sees only stores and never waits (tools/vmcnt_audit.py checks the built code).
//   grid:  top of group i: wait for the input fragments of group i (younger: the 8 u' stores of group i-1 -> vmcnt(8));
//          after conv0 has consumed them, request group i+1's into the same registers.
//   block: the input fragments of group i+1 are requested late in group i and are covered by that group's last wait;
//          RSHMAG.dense2's weights stream in four chunks of 8 loads through two register sets, the u' rows (8 loads) are
//          requested in front of the first; see the waits in the loop body, each annotated with what is younger.
//   tail:  x1 (8 loads) and the input fragments (4) of group i+1 are requested once group i has consumed its own (after
//          r = x1 + x0); top of a group: vmcnt(2) (younger: the previous group's two stores).  The compiler's own loads of
//          the squeeze-excite scale follow the prefetch in the queue; its counted waits for them cover the prefetch too.
template <int MODE>
__global__ __launch_bounds__(s2_waves<MODE>() * 64, 1) void stage2_kernel16(StageArgs A) {
    constexpr int C = kS2C, NW = s2_waves<MODE>(), NTHR = NW * 64, NP = NW / 2;
    constexpr int BM = MODE == 0 ? 0 : 1;                        // branch whose weights / token geometry this kernel uses
    constexpr bool TAIL = MODE == 2;
    static_assert(kFmt32[1] && !kFmt32[2], "stage 2 reads and writes 32x32 fragments; its output feeds the 16x16 kernels of stage 3");
    static_assert(NW % 2 == 0, "waves come in pairs");
    using M = S2Map<MODE>;
    constexpr int STAMP_KID = 2 + BM; (void)STAMP_KID;           // (diagnostic build only: tools/stamps_s2.py)
    STAMP_DECL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *par = reinterpret_cast<float *>(smem_raw + M::par);
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, w = wave & 1;                    // w: which 32 tokens of the group this wave owns
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[BM];

    // ---- stage the weights, parameters and the chord table once per workgroup ----
    {
        auto copy = [&](int dst, int src_floats, int bytes) {
            const char *s = reinterpret_cast<const char *>(blob + src_floats);
            for (int i = threadIdx.x * 16; i < bytes; i += NTHR * 16)
                *reinterpret_cast<uint4 *>(smem_raw + dst + i) = *reinterpret_cast<const uint4 *>(s + i);
        };
        if (MODE == 0) copy(0, kLayout.gelu_lut, s2_lut_bytes<0>());
        if (MODE == 1) copy(0, kLayout.gelu_lut2, s2_lut_bytes<1>());
        copy(M::conv0, S.conv0_w, 4 * 2048);
        if (TAIL) {
            copy(M::q1, S.r1_w, 8 * 2048);
            copy(M::d1, S.r2_w, 8 * 2048);
        } else {
            copy(M::q1, S.q1_w + BM * (8 * 512), 8 * 2048);      // output rows BM * 64 ..: row tiles 2 BM, 2 BM + 1 (512 floats per tile)
            copy(M::d1, Br.d1_w, 16 * 2048);
            copy(M::mix, Br.mix_w, 8 * 2048);
            copy(M::d2, Br.d2_w, 8 * 2048);
            if (MODE == 1) copy(M::r1, S.r1_w, 8 * 2048);
        }
        for (int i = threadIdx.x; i < kS2ParFloats; i += NTHR) {
            float v;
            if (i < kS2pQ1B) v = blob[S.conv0_b + i];
            else if (i < kS2pD1B) v = blob[S.q1_b + BM * C + (i - kS2pQ1B)];
            else if (i < kS2pGlnG) v = blob[Br.d1_b + (i - kS2pD1B)];
            else if (i < kS2pGlnB) v = blob[Br.gln_g + (i - kS2pGlnG)];
            else if (i < kS2pMixB1) v = blob[Br.gln_b + (i - kS2pGlnB)];
            else if (i < kS2pD2B) v = blob[Br.mix_b + (i - kS2pMixB1)] + 1.0f;
            else if (i < kS2pQ2B) v = blob[Br.d2_b + (i - kS2pD2B)];
            else if (i < kS2pR1B) v = blob[S.q2_b + (i - kS2pQ2B)];
            else if (i < kS2pR2B) v = blob[S.r1_b + (i - kS2pR1B)];
            else v = blob[S.r2_b + (i - kS2pR2B)];
            par[i] = v;
        }
        if (threadIdx.x < 16) *reinterpret_cast<unsigned *>(smem_raw + M::flags + threadIdx.x * 4) = 0u;
        __syncthreads();                                         // the only barrier of the kernel
    }
    const unsigned char *wl = smem_raw + lane * 16;              // weight fragments: + region + tile * 2048 (+ 1024: lo)

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    const int total = A.B * per_img;
    // XCD-aware persistent schedule, as in stage 1, in units of wave PAIRS: workgroups b and b + 8 share an XCD (L2)
    const int npx = (gridDim.x >> 3) * NP;                       // pairs per XCD
    const int px = (blockIdx.x >> 3) * NP + pair, xcd = blockIdx.x & 7;

    struct Pos { int n, gy, gx; };
    struct Geo { int n, y, x; };
    auto decompose = [&](int i) {
        Pos c;
        c.n = i / per_img;
        const int rem = i - c.n * per_img;
        c.gy = rem / fw;
        c.gx = rem - c.gy * fw;
        c.n = __builtin_amdgcn_readfirstlane(c.n);
        c.gy = __builtin_amdgcn_readfirstlane(c.gy);
        c.gx = __builtin_amdgcn_readfirstlane(c.gx);
        return c;
    };
    const int ty_ = 4 * w + (n >> 3), tx_ = n & 7;               // the lane's token t = 32 w + n = 8 ty + tx
    auto geo = [&](const Pos &c) {
        Geo g;
        g.n = c.n;
        if (MODE == 0) { g.y = ty_ * fh + c.gy; g.x = tx_ * fw + c.gx; }
        else           { g.y = 8 * c.gy + ty_;  g.x = 8 * c.gx + tx_; }
        return g;
    };
    const int stride = 8 * npx;
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
    const int hw = H * W;
    // the stage input's fragments of the lane's pixel (32x32 format, 128 B per pixel): K-step s at + 64 s: [hi: h0 h1][lo: h0 h1].
    // asm loads into the loop-carried registers nx ("+v" everywhere: one register set, never copied -- a copy in front of the
    // wait would read data that has not arrived)
    auto issue_in = [&](const Geo &g, HL (&nx)[2]) {
        const unsigned vo = (unsigned)(g.y * W + g.x) * 128u + (unsigned)h * 16u;
        const char *xb = uniform_ptr(reinterpret_cast<const char *>(A.X) + (long)g.n * (long)hw * 128);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:32\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:64\n\tglobal_load_dwordx4 %3, %4, %5 offset:96"
                     : "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi), "+v"(nx[1].lo) : "v"(vo), "s"(xb) : "memory");
    };

    int item = xcd * npx + px;                                   // wave-uniform; the same for both waves of a pair
    Pos nxt = decompose(item);
    HL nx[2] = {};
    // (tail) x1 of the next half group as the block kernel left it -- register order: per wave half 8 KB = [tile][register quad]
    // [lane] x 16 B --, prefetched like the input fragments
    f4 nx1[TAIL ? 2 : 1][TAIL ? 4 : 1] = {};
    auto issue_x1 = [&](int it) {
        if constexpr (TAIL) {
            const char *rp = uniform_ptr(reinterpret_cast<const char *>(A.R) + ((long)it * 2 + w) * (32 * C * 4));
            const unsigned lo16 = (unsigned)lane * 16u;
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %9\n\tglobal_load_dwordx4 %1, %8, %9 offset:1024\n\t"
                         "global_load_dwordx4 %2, %8, %9 offset:2048\n\tglobal_load_dwordx4 %3, %8, %9 offset:3072\n\t"
                         "global_load_dwordx4 %4, %8, %10\n\tglobal_load_dwordx4 %5, %8, %10 offset:1024\n\t"
                         "global_load_dwordx4 %6, %8, %10 offset:2048\n\tglobal_load_dwordx4 %7, %8, %10 offset:3072"
                         : "+v"(nx1[0][0]), "+v"(nx1[0][1]), "+v"(nx1[0][2]), "+v"(nx1[0][3]), "+v"(nx1[1][0]), "+v"(nx1[1][1]),
                           "+v"(nx1[1][2]), "+v"(nx1[1][3])
                         : "v"(lo16), "s"(rp), "s"(rp + 4096) : "memory");
        }
    };
    if (item < total) {
        issue_in(geo(nxt), nx);
        issue_x1(item);
        if constexpr (TAIL)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx1[0][0]), "+v"(nx1[0][1]), "+v"(nx1[0][2]), "+v"(nx1[0][3]), "+v"(nx1[1][0]),
                         "+v"(nx1[1][1]), "+v"(nx1[1][2]), "+v"(nx1[1][3])::"memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi), "+v"(nx[1].lo)::"memory");
    }
    unsigned char *bT = smem_raw + M::tiles + (TAIL ? 0 : pair * kS1BtBytes);
    const unsigned my_flags = (unsigned)(M::flags + wave * 8), peer_flags = (unsigned)(M::flags + (wave ^ 1) * 8);
    unsigned pass = 0;                                           // token-tile passes completed by this pair (wave-uniform)

    float rng = 0.0f;                                            // (tail) max |v| over the stage's output (status block)
    for (; item < total; item += stride) {
        const Geo g = geo(nxt);
        const bool more = item + stride < total;
        if (more) nxt = advance(nxt);                            // (the last group re-requests its own pixels)
        const long pix = ((long)g.n * H + g.y) * W + g.x;

        if constexpr (TAIL) {
            // ---- the stage's tail: x_next = maxpool2x2(x1 + x0 + s * conv2(lrelu(conv1(LN(x1))))) in 16x16 fragment format ----
            // x1 and the stage input's fragments were requested a group ago (younger: the previous group's two stores)
            asm volatile(BALF_S2_WAIT(2) : "+v"(nx1[0][0]), "+v"(nx1[0][1]), "+v"(nx1[0][2]), "+v"(nx1[0][3]), "+v"(nx1[1][0]),
                         "+v"(nx1[1][1]), "+v"(nx1[1][2]), "+v"(nx1[1][3]), "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi),
                         "+v"(nx[1].lo)::"memory");
            // x0 = relu(conv0(X)) first, then LN(x1), then r = x1 + x0 IN x0's registers: from there on x1 and the input
            // fragments are dead and the next group's can be requested into the same registers (no copies), with conv1, conv2
            // and the pooling in front of them to cover the round trip
            f16v x0[2];
            s2_bias(x0, par + kS2pConv0B, h);
            s2_linear<2>(x0, wl + M::conv0, nx);
            relu32(x0);
            HL b[4];
            {
                float s = nx1[0][0][0], ss = nx1[0][0][0] * nx1[0][0][0];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int r = (rt == 0 && gq == 0 ? 1 : 0); r < 4; ++r) { s += nx1[rt][gq][r]; ss = fmaf(nx1[rt][gq][r], nx1[rt][gq][r], ss); }
                half_allreduce2(s, ss);
                const float mean = s * (1.0f / C);
                const float var = fmaf(ss, 1.0f / C, -mean * mean);
                const float rstd = __builtin_amdgcn_rsqf(max0(var) + kLnEps), shift = -mean * rstd;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                // K-step ks = register quads 2 (ks & 1), + 1 of tile ks >> 1
                    f4 y0, y1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        y0[r] = fmaf(nx1[ks >> 1][2 * (ks & 1)][r], rstd, shift);
                        y1[r] = fmaf(nx1[ks >> 1][2 * (ks & 1) + 1][r], rstd, shift);
                    }
                    b[ks] = split8<BALF_S1_SPLIT_MIX>(y0, y1);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x0[rt][4 * gq + r] += nx1[rt][gq][r];           // r = x1 + x0
            __builtin_amdgcn_sched_barrier(0);                   // everything that reads nx1 / nx has been issued
            issue_in(geo(nxt), nx);
            issue_x1(more ? item + stride : item);
            f16v m1[2];
            s2_bias(m1, par + kS2pR1B, h);
            s2_linear<4>(m1, wl + M::q1, b);
            lrelu32(m1);
            s2_split(m1, b);
            f16v t[2];
            s2_bias(t, par + kS2pR2B, h);
            s2_linear<4>(t, wl + M::d1, b);
            // v = r + s t; max over the 2x2 window: tx partner = lane ^ 1, ty partner = lane ^ 8 (same 16-lane row)
            f16v mx[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f4 sc = *reinterpret_cast<const f4 *>(A.scale + (long)g.n * C + 32 * rt + 8 * gq + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * gq + r;
                        const float v = fmaf(t[rt][i], sc[r], x0[rt][i]);
                        const int vi = __builtin_bit_cast(int, v);
                        const float o1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false));
                        const float m = __builtin_fmaxf(v, o1);
                        const float o2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
                        mx[rt][i] = __builtin_fmaxf(m, o2);
                    }
                }
            // The four lanes of a 2x2 window (x the two lane halves) hold the same pooled pixel: lane (a = tx & 1, b = ty & 1, h)
            // stores K-step a (channels 32 a ..), lane quarter q = h + 2 b of the 16x16 format: channels 4 q + (0..3) and
            // 16 + 4 q + (0..3) of the K-step = this lane's registers 4 b + (0..3) and 8 + 4 b + (0..3) of tile a.
            {
                const unsigned am = 0u - (unsigned)(n & 1), bm = 0u - (unsigned)((n >> 3) & 1);
                float sel[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r0 = (k < 4) ? k : 8 + (k - 4);
                    const float c0 = lane_select(bm, mx[0][r0 + 4], mx[0][r0]);
                    const float c1 = lane_select(bm, mx[1][r0 + 4], mx[1][r0]);
                    sel[k] = lane_select(am, c1, c0);
                }
                HL o;
                h2 hh, ll;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rng = range_max(rng, sel[2 * i], sel[2 * i + 1]);       // the stage's output, about to be split (status block)
                    split_pair<BALF_S1_SPLIT_MIX>(sel[2 * i], sel[2 * i + 1], hh, ll);
                    o.hi[2 * i] = hh[0]; o.hi[2 * i + 1] = hh[1]; o.lo[2 * i] = ll[0]; o.lo[2 * i + 1] = ll[1];
                }
                const long opix = ((long)g.n * (H / 2) + (g.y >> 1)) * (W / 2) + (g.x >> 1);
                store_frag_px(A.out, opix, C, n & 1, h + 2 * ((n >> 3) & 1), o);
            }
        } else {
        STAMP(0);
        // ---- x0 = relu(conv0(X)) from the prefetched fragments; then the next group's are requested ----
        if constexpr (MODE == 0)
            asm volatile(BALF_S2_WAIT(8) : "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi), "+v"(nx[1].lo)::"memory");
        f16v x0[2];
        s2_bias(x0, par + kS2pConv0B, h);
        s2_linear<2>(x0, wl + M::conv0, nx);
        relu32(x0);
        if constexpr (MODE == 0) {
            __builtin_amdgcn_sched_barrier(0);                   // conv0's MFMAs have been issued: nx may be overwritten
            issue_in(geo(nxt), nx);
        }
        HL b[4];
        s2_ln_split(x0, b);
        STAMP(1);   // wait for the input, conv0, relu, LN + split
        f16v z[2];                                               // u (grid) / v (block): kept for the branch residual
        s2_bias(z, par + kS2pQ1B, h);
        s2_linear<4>(z, wl + M::q1, b);
        s2_gelu<MODE>(z);
        STAMP(2);   // dense1 half + GELU
        s2_ln_split(z, b);
        STAMP(3);   // LN + split
        f16v ga[2];
        s2_bias(ga, par + kS2pD1B, h);
        s2_linear<4>(ga, wl + M::d1, b);
        s2_gelu<MODE>(ga);
        STAMP(4);   // branch dense1 (a half) + GELU
        {
            f16v gb[2];
            s2_bias(gb, par + kS2pD1B + C, h);
            s2_linear<4>(gb, wl + M::d1 + 8 * 2048, b);
            s2_gelu<MODE>(gb);
            float rstd, shift;
            s2_ln_stats(gb, rstd, shift);                        // gating LayerNorm (affine) over the pixel's 64 channels
            STAMP(5);   // branch dense1 (b half) + GELU + LN statistics
            // write offsets of the lane's token (column t = 32 w + n) in channel rows 4 h + r (+ 8 g: an immediate)
            int wo[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wo[r] = (4 * h + r) * 128 + (((4 * w + (n >> 3)) ^ (4 * h + r)) << 4) + (n & 7) * 2;
            const float mb1 = par[kS2pMixB1 + 32 * w + n];       // mix bias + 1 of the lane's output token
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                // one 32-channel row tile per pass through the pair's token tile
                s2_poll(peer_flags + 4, pass + rt);             // the partner has read the previous pass
                if (rt == 0) STAMP(6); else STAMP(10);           // waiting for the partner's reads
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f4 gg = *reinterpret_cast<const f4 *>(par + kS2pGlnG + 32 * rt + 8 * gq + 4 * h);
                    const f4 bb = *reinterpret_cast<const f4 *>(par + kS2pGlnB + 32 * rt + 8 * gq + 4 * h);
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const float v0 = fmaf(fmaf(gb[rt][4 * gq + r], rstd, shift), gg[r], bb[r]);
                        const float v1 = fmaf(fmaf(gb[rt][4 * gq + r + 1], rstd, shift), gg[r + 1], bb[r + 1]);
                        h2 hh, ll;
                        split_pair<BALF_S1_SPLIT_MIX>(v0, v1, hh, ll);
                        unsigned char *p0 = bT + wo[r] + gq * 1024, *p1 = bT + wo[r + 1] + gq * 1024;
                        *reinterpret_cast<_Float16 *>(p0) = hh[0];
                        *reinterpret_cast<_Float16 *>(p1) = hh[1];
                        *reinterpret_cast<_Float16 *>(p0 + kS1BtPlane) = ll[0];
                        *reinterpret_cast<_Float16 *>(p1 + kS1BtPlane) = ll[1];
                    }
                }
                s2_post(my_flags, pass + rt + 1);                // written (LDS instructions of a wave execute in order)
                if (rt == 0) STAMP(7); else STAMP(11);           // gating LN + split + tile writes
                s2_poll(peer_flags, pass + rt + 1);              // ... and the partner's half is there too
                if (rt == 0) STAMP(8); else STAMP(12);           // waiting for the partner's writes
                HL a[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const unsigned char *row = bT + s1_bt_rd(n, 2 * s + h);
                    a[s].hi = *reinterpret_cast<const h8 *>(row);
                    a[s].lo = *reinterpret_cast<const h8 *>(row + kS1BtPlane);
                }
                s2_post(my_flags + 4, pass + rt + 1);            // read
                // mix^T[c][t'] = sum_t tile[c][t] Wmix[t'][t] (+ bias[t'] + 1 as the start value), then the gate
                f16v m;
#pragma unroll
                for (int r = 0; r < 16; ++r) m[r] = mb1;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    HL wv;
                    wv.hi = *reinterpret_cast<const h8 *>(wl + M::mix + (w * 4 + s) * 2048);
                    wv.lo = *reinterpret_cast<const h8 *>(wl + M::mix + (w * 4 + s) * 2048 + 1024);
                    m = mfma32(a[s].lo, wv.hi, m);
                    m = mfma32(a[s].hi, wv.lo, m);
                    m = mfma32(a[s].hi, wv.hi, m);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ga[rt][r] *= m[r];
                if (rt == 0) STAMP(9); else STAMP(13);           // tile reads, mix, gate
            }
            pass += 2;
        }
        f16v o[2];
        if constexpr (MODE == 0) {
            s2_split(ga, b);
            s2_bias(o, par + kS2pD2B, h);
            s2_linear<4>(o, wl + M::d2, b);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[rt][r] += z[rt][r];
            STAMP(14);  // split + dense2 + residual
#pragma unroll
            for (int s = 0; s < 4; ++s) store_frag32(A.U, uwin_pix(pix, C), C, s, h, s1_split8(o[s >> 1], s & 1));
            STAMP(15);  // split + u' store
        } else {
            // ---- block branch: RSHMAG.dense2 over cat[u', v'] with its weights streamed from L2 ----
            // chunk c = (row tile c & 1, K-steps 4 (1 - (c >> 1)) ..+3): c0, c1 = the v' half (K-steps 4-7), c2, c3 = the u' half
            HL wq[2][4];
            const char *qb = reinterpret_cast<const char *>(blob + S.q2_w);
            const unsigned lo16 = (unsigned)lane * 16u;
#define BALF_S2_QLOAD(SET, CH)                                                                                               \
    do {                                                                                                                     \
        const char *qp = uniform_ptr(qb + ((CH & 1) * 8 + ((CH >> 1) ? 0 : 4)) * 2048);                                      \
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %9\n\tglobal_load_dwordx4 %1, %8, %9 offset:1024\n\t"           \
                     "global_load_dwordx4 %2, %8, %9 offset:2048\n\tglobal_load_dwordx4 %3, %8, %9 offset:3072\n\t"          \
                     "global_load_dwordx4 %4, %8, %10\n\tglobal_load_dwordx4 %5, %8, %10 offset:1024\n\t"                    \
                     "global_load_dwordx4 %6, %8, %10 offset:2048\n\tglobal_load_dwordx4 %7, %8, %10 offset:3072"            \
                     : "=&v"(wq[SET][0].hi), "=&v"(wq[SET][0].lo), "=&v"(wq[SET][1].hi), "=&v"(wq[SET][1].lo),               \
                       "=&v"(wq[SET][2].hi), "=&v"(wq[SET][2].lo), "=&v"(wq[SET][3].hi), "=&v"(wq[SET][3].lo)                \
                     : "v"(lo16), "s"(qp), "s"(qp + 4096) : "memory");                                                        \
    } while (0)
#define BALF_S2_QWAIT(SET, N, ...)                                                                                           \
    asm volatile(BALF_S2_WAIT(N) : "+v"(wq[SET][0].hi), "+v"(wq[SET][0].lo), "+v"(wq[SET][1].hi), "+v"(wq[SET][1].lo),       \
                 "+v"(wq[SET][2].hi), "+v"(wq[SET][2].lo), "+v"(wq[SET][3].hi), "+v"(wq[SET][3].lo) __VA_ARGS__::"memory")
            // The u' rows of the lane's pixel (written by the grid kernel just before: L2 / Infinity Cache) are requested BEFORE
            // dense2, whose split + 24 MFMAs cover most of their round trip; the weight chunks and the next group's input
            // fragments behind it (with z, the gated branch, dense2's accumulators and weight fragments live, 32 more
            // registers there spill).  The phase stamps showed 2.5 k of a group's 49 k cycles waiting for u' when it was
            // requested after dense2.
            HL ub[4];
            {
#if BALF_ABLATE_UWINDOW
                const unsigned uo = (unsigned)uwin_pix((long)g.n * hw + g.y * W + g.x, 64) * 256u + (unsigned)h * 16u;
                const char *ubase = uniform_ptr(reinterpret_cast<const char *>(A.U));
#else
                const unsigned uo = (unsigned)(g.y * W + g.x) * 256u + (unsigned)h * 16u;
                const char *ubase = uniform_ptr(reinterpret_cast<const char *>(A.U) + (long)g.n * (long)hw * 256);
#endif
                asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %8, %9\n\tglobal_load_dwordx4 %1, %8, %9 offset:32\n\t"
                             "global_load_dwordx4 %2, %8, %9 offset:64\n\tglobal_load_dwordx4 %3, %8, %9 offset:96\n\t"
                             "global_load_dwordx4 %4, %8, %9 offset:128\n\tglobal_load_dwordx4 %5, %8, %9 offset:160\n\t"
                             "global_load_dwordx4 %6, %8, %9 offset:192\n\tglobal_load_dwordx4 %7, %8, %9 offset:224"
                             : "=&v"(ub[0].hi), "=&v"(ub[0].lo), "=&v"(ub[1].hi), "=&v"(ub[1].lo), "=&v"(ub[2].hi), "=&v"(ub[2].lo),
                               "=&v"(ub[3].hi), "=&v"(ub[3].lo)
                             : "v"(uo), "s"(ubase) : "memory");
            }
            s2_split(ga, b);
            s2_bias(o, par + kS2pD2B, h);
            s2_linear<4>(o, wl + M::d2, b);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[rt][r] += z[rt][r];
            STAMP(14);  // requests, split + dense2 + residual
            f16v x1[2];
            if constexpr (BALF_ABLATE_QSTREAM) {                 // (timing experiment: what the streaming itself costs)
                issue_in(geo(nxt), nx);
                s2_split(o, b);
                s2_bias(x1, par + kS2pQ2B, h);
                s2_linear_rt<4>(x1[0], wl + M::d1, b);
                s2_linear_rt<4>(x1[1], wl + M::d1 + 4 * 2048, b);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ub[0].hi), "+v"(ub[0].lo), "+v"(ub[1].hi), "+v"(ub[1].lo), "+v"(ub[2].hi), "+v"(ub[2].lo),
                             "+v"(ub[3].hi), "+v"(ub[3].lo), "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi), "+v"(nx[1].lo)::"memory");
                s2_linear_rt<4>(x1[0], wl + M::d1 + 8 * 2048, ub);
                s2_linear_rt<4>(x1[1], wl + M::d1 + 12 * 2048, ub);
            } else {
            BALF_S2_QLOAD(0, 0);
            BALF_S2_QLOAD(1, 1);
            issue_in(geo(nxt), nx);
            s2_split(o, b);                                      // v'
            s2_bias(x1, par + kS2pQ2B, h);
            STAMP(15);  // requests, split of v'
            BALF_S2_QWAIT(0, 12);                                // c0 has landed (and u', older); younger: c1 8 + next input 4
            STAMP(16);  // waiting for the first weight chunk
            s2_linear_regs(x1[0], wq[0], b);
            __builtin_amdgcn_sched_barrier(0);                   // c0's MFMAs are issued: its registers take c2
            BALF_S2_QLOAD(0, 2);
            BALF_S2_QWAIT(1, 12);                                // c1; younger: next input 4 + c2 8
            s2_linear_regs(x1[1], wq[1], b);
            __builtin_amdgcn_sched_barrier(0);
            BALF_S2_QLOAD(1, 3);
            // c2 (u' and the next input's fragments, older, have landed before it); younger: c3 8
            STAMP(17);  // v' half of RSHMAG.dense2
            BALF_S2_QWAIT(0, 8, , "+v"(ub[0].hi), "+v"(ub[0].lo), "+v"(ub[1].hi), "+v"(ub[1].lo), "+v"(ub[2].hi), "+v"(ub[2].lo),
                          "+v"(ub[3].hi), "+v"(ub[3].lo), "+v"(nx[0].hi), "+v"(nx[0].lo), "+v"(nx[1].hi), "+v"(nx[1].lo));
            STAMP(18);  // waiting for u' and the third chunk
            s2_linear_regs(x1[0], wq[0], ub);
            BALF_S2_QWAIT(1, 0);                                 // c3
            s2_linear_regs(x1[1], wq[1], ub);
            }
            STAMP(19);  // u' half
#undef BALF_S2_QLOAD
#undef BALF_S2_QWAIT
            // x1 = . + x0, stored in REGISTER ORDER for the tail kernel (same group and lane geometry): per wave half 8 KB =
            // [tile][register quad][lane] x 16 B, every store moves 1 KiB of contiguous memory
            float *rp = A.R + ((long)item * 2 + w) * (32 * C);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x1[rt][r] += x0[rt][r];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<f4 *>(rp + ((rt * 4 + gq) * 64 + lane) * 4) =
                        f4{x1[rt][4 * gq], x1[rt][4 * gq + 1], x1[rt][4 * gq + 2], x1[rt][4 * gq + 3]};
            }
            STAMP(20);  // residual + x1 store
            s2_ln_split(x1, b);
            f16v m1[2];
            s2_bias(m1, par + kS2pR1B, h);
            s2_linear<4>(m1, wl + M::r1, b);
            lrelu32(m1);
            STAMP(21);  // LN + split + conv1 + lrelu
            // conv2 is linear: its channel means follow from those of its input (SE kernel); the tail kernel recomputes the
            // branch from x1.  Channel sums over the wave's 32 pixels (fixed order), one partial row per wave half: the two
            // 16-lane rows of a lane half as a reduce-scatter (one row swap serves registers r and r + 8), then the row.
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float cs[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    // (plain float temporaries: __builtin_bit_cast applied to a vector ELEMENT, bit_cast(unsigned, m1[rt][r]), reads
                    // the vector's first element whatever r is -- clang takes the object representation at the vector's address.
                    // Found by tests/experiments/s2_debug.py as SE scales off by 0.4: every channel sum was that of register 0.)
                    const float ea = m1[rt][r], eb = m1[rt][r + 8];
                    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                    unsigned w0 = sw[0], w1 = sw[1];             // (opaque: hipcc once folded sw[1] into sw[0] here, see stage1_f16.h)
                    asm("" : "+v"(w0), "+v"(w1));
                    const float s = __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
                    cs[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s))));
                }
                if ((lane & 15) == 0) {
                    // row 0: (h 0, regs 0-7): channels 0-3, 8-11; row 1: (h 0, regs 8-15): 16-19, 24-27; row 2: (h 1, regs 0-7):
                    // 4-7, 12-15; row 3: (h 1, regs 8-15): 20-23, 28-31 (+ 32 rt)
                    const int row = lane >> 4, c0 = 32 * rt + 16 * (row & 1) + 4 * (row >> 1);
                    float *pp = A.partial + ((long)item * 2 + w) * C + c0;
                    *reinterpret_cast<f4 *>(pp) = f4{cs[0], cs[1], cs[2], cs[3]};
                    *reinterpret_cast<f4 *>(pp + 8) = f4{cs[4], cs[5], cs[6], cs[7]};
                }
            }
            STAMP(22);  // channel sums
        }
        }   // !TAIL
    }
    if (TAIL && rng >= kF16Max) status_raise(A.status, 1 /* BALF_STATUS_RANGE */);
}
