// Channel-split stage kernels of the split-f16 path: the grid branch, block branch and (C <= 128) tail kernels of every
// stage that is not served by a persistent wave-owns-tokens kernel (stage1_f16.h) -- which stages those are is decided in
// detector_f16.hip (run_stage16).  Included by detector_f16.hip inside balf::{anonymous}.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 64 / 128 / 256.
//
// One workgroup = one token group (64 tokens); it splits every Linear by OUTPUT CHANNELS: wave w (of C/32) owns channels
// 32w .. 32w+31 of all 64 tokens -- a 2 x 4 register tile (two 16-channel row tiles x four 16-pixel tiles) on
// v_mfma_f32_16x16x32_f16.  Consequences:
//   * a weight fragment is needed by exactly ONE wave: it comes straight from L2 into double-buffered registers (two
//     K-steps ahead of the MFMAs), there is no LDS copy of the weights and no barrier per weight tile;
//   * the 64x64 token mix of a channel is wave-local (transposed tile in wave-private LDS, as in stage 1);
//   * what the waves exchange is the ACTIVATION: a Linear's input is the B operand of all waves, so every wave writes its
//     32 channels as one K-step of split-f16 fragments into a shared buffer (64 KB at C = 256) and a barrier publishes
//     it; LayerNorm needs the statistics of all channels of a pixel, exchanged through a small LDS table.  7 barriers per
//     token group in the grid branch, 13 in the block branch.
// (History: rounds 1-2 shared every weight tile through an LDS ring with one barrier per ring unit, 86 / 157 per token
// group at C = 256; those kernels were removed in round 3.)
// Token t = 8 ty + tx of the group sits in MFMA column li of pixel tile p with t = 4 li + p.
#pragma once

template <int C> constexpr int cs_waves() { return C / 32; }
template <int C> constexpr int cs_bx_bytes() { return (C / 32) * 4 * 2048; }          // [K-step][tile][hi|lo][64 x 16 B]
template <int C> constexpr int cs_bt_bytes() {                                          // token tiles / u' staging
    constexpr int bt = cs_waves<C>() * kS1BtBytes, bu = cs_bx_bytes<C>();
    return bt > bu ? bt : bu;
}
template <int C> constexpr bool cs_mix_in_lds() { return C >= 256; }
// Fused tail (as in stages 1-2): at C = 128 the block kernel stores x1 and the channel sums of the RCAB's hidden layer, and a tail
// kernel (stage3_tail_f16.h; until round 4 MODE 2 of the kernel below) recomputes x0 and the RCAB branch from x1, scales, pools and
// writes the next stage's input.  Not at C = 256: there the consumer is the head kernel.
template <int C> constexpr bool cs_fused() { return C <= 128; }
template <int C> constexpr int cs_lds_bytes() {
    return cs_bx_bytes<C>() + cs_bt_bytes<C>() + (cs_mix_in_lds<C>() ? 8 * 2048 : 0) + cs_waves<C>() * 4 * 16 * 8;
}
#ifndef BALF_CS_GELU_CHUNK
#define BALF_CS_GELU_CHUNK 8 // table reads in flight per wave (3 registers per value)
#endif
#ifndef BALF_CS_GELU_LUT
#define BALF_CS_GELU_LUT 1   // bit m set: GELU of the MODE-m kernels from the log-spaced chord table (layout.h: kGeluLogM) at LDS offset 0; else 2^P form
#endif
// Measured per 8 x 1088x1920, same box: grid kernels C = 64 1.13 -> 1.05 ms, C = 256 0.56 -> 0.54, C = 128 unchanged; the block
// kernels (11-13 barriers per group: every exposed LDS round trip of a table chunk is on the group's critical path) lose
// at C = 64 (1.37 -> 1.45) and gain nothing above, so they keep the 2^P form.
template <int MODE> constexpr bool cs_gelu_lut() { return MODE < 2 && ((BALF_CS_GELU_LUT >> MODE) & 1) != 0; }
template <int MODE> constexpr int cs_lut_bytes() { return cs_gelu_lut<MODE>() ? (kGeluLogEntries * 8 + 15) / 16 * 16 : 0; }

// GELU from the log-spaced table: gelu(x) = x / 2 + E(|x|), E from the chord of the interval that the float
// s = clamp01((|x| + 1) / 8) names with its low two exponent bits and top eight mantissa bits:
//   s          v_fma_f32 |x|, 1/8, 1/8 clamp     (the compiler's instruction: x is an MFMA result, see stage1_f16.h)
//   off        (bits(s) >> 12) & 0x1FF8          byte offset of the (A, B) pair; s = 1 (|x| >= 7) names the asymptote entry
//   gelu       x / 2 + fma(B, |x|, A)
// 6 vector instructions (20 issue cycles at the measured class rates) + 1 LDS read against 8 (33) for the 2^P form.
template <int N>
__device__ __forceinline__ void gelu_log_n(float (&x)[N]) {
    static_assert(N == 4 || N == 8, "");
    static_assert(kGeluLogM == 256, "the bit field below is 2 exponent + 8 mantissa bits");
    unsigned o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float s = __builtin_amdgcn_fmed3f(fmaf(__builtin_fabsf(x[i]), 0.125f, 0.125f), 0.0f, 1.0f);
        o[i] = (__builtin_bit_cast(unsigned, s) >> 12) & 0x1FF8u;
    }
    f2 ab[N];
    // (the dynamic LDS block starts at LDS address 0 and the table is its first region: the offset IS the address)
    if constexpr (N == 8)
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                     "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(ab[0]), "=&v"(ab[1]), "=&v"(ab[2]), "=&v"(ab[3]), "=&v"(ab[4]), "=&v"(ab[5]), "=&v"(ab[6]), "=&v"(ab[7])
                     : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]));
    else
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(ab[0]), "=&v"(ab[1]), "=&v"(ab[2]), "=&v"(ab[3])
                     : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // x / 2 by the compiler, E = B |x| + A as asm IN x's OWN REGISTER, the sum by the compiler.  No inline-asm write
        // into a fresh register: such a write can land on the SrcC of an MFMA of the next Linear that hipcc hoisted in
        // front of it (split16.h) -- hipcc pads that hazard for its own instructions only.  (Both fmas left to hipcc: it
        // pairs them into v_pk_fma_f32 behind 100 extra v_mov_b32 per wave.)
        const float half = 0.5f * x[i];
        asm("v_fma_f32 %0, %1, |%0|, %2" : "+v"(x[i]) : "v"(ab[i][1]), "v"(ab[i][0]));
        x[i] += half;
    }
}

template <bool LUT, int CHUNK>
__device__ __forceinline__ void cs_gelu(f4 (&t)[2][4]) {
    if constexpr (!LUT) {
        gelu<false>(t);
        return;
    }
    if (BALF_ABLATE_GELU) return;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int p = 0; p < 4; p += CHUNK / 4) {
            float v[CHUNK];
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) v[i] = t[nt][p + (i >> 2)][i & 3];
            gelu_log_n(v);
#pragma unroll
            for (int i = 0; i < CHUNK; ++i) t[nt][p + (i >> 2)][i & 3] = v[i];
        }
}

#ifndef BALF_CS_SCHED
#define BALF_CS_SCHED 1      // 1: a scheduling fence only behind the weight requests (measured best; 0: none, 2: also behind the MFMAs)
#endif
// Weight fragments of one K-step pair of this wave's two row tiles (32 registers) and a Linear's bias slice.
struct CsW {
    HL a[2][2];       // [K-step of the pair][row tile]
};

// Weight tile (t, ks) of this wave at byte wbase + (t * kstot + ks) * 2048 (+ 1024: lo) of the blob, straight from L2.
// Buffer loads: the blob's resource descriptor and every tile offset are wave-uniform (scalar registers), the lane's
// 16 bytes are one shared 32-bit vector offset -- no vector instructions for addresses (global_load with 64-bit lane
// pointers cost ~100 of them per token group).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct CsBlob {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned loff;                                        // lane * 16
    __device__ __forceinline__ f4 vec(unsigned off) const {
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, loff, off, 0));
    }
    __device__ __forceinline__ h8 frag(unsigned off) const {
#if BALF_ABLATE_WSTREAM
        off &= 1023u;                                         // timing experiment: every weight tile is the same L1-resident KiB (wrong results)
#endif
        return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, loff, off, 0));
    }
};
template <int NSTEP = 2>
__device__ __forceinline__ void cs_wload(CsW &w, const CsBlob &bl, unsigned wbase, int kstot, int ks) {
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned o = wbase + (unsigned)((t * kstot + ks + s) * 2048);
            w.a[s][t].hi = bl.frag(o);
            w.a[s][t].lo = bl.frag(o + 1024);
        }
}

// acc[t][p] += W(row tiles t = 0, 1 of this wave) . B over KSN K-steps.  `first` holds the fragments of K-steps 0, 1,
// requested by the caller BEFORE the epilogue / barriers in front of this Linear (an L2 round trip at the head of every
// Linear is otherwise exposed: the workgroup's waves move in step, nothing else covers it); inside the loop the next
// pair is requested while the current one is consumed.  bsrc(ks) -> the shared B fragments of K-step ks, [p][hi|lo][lane].
template <int KSN, typename BS>
__device__ __forceinline__ void cs_linear(f4 (&acc)[2][4], const CsW &first, const CsBlob &bl, unsigned wbase, int kstot, BS bsrc, int lane) {
    static_assert(KSN == 1 || KSN % 2 == 0, "K-steps come in pairs (or a single one: conv0 of stage 2)");
    auto compute = [&](const CsW &w, int ks) {
#pragma unroll
        for (int s = 0; s < (KSN == 1 ? 1 : 2); ++s) {
            const h8 *bk = bsrc(ks + s);
            HL b[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                b[p].hi = bk[(p * 2 + 0) * 64 + lane];
                b[p].lo = bk[(p * 2 + 1) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) if (!BALF_DROP_WLO) acc[t][p] = mfma16(w.a[s][t].lo, b[p].hi, acc[t][p]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[t][p] = mfma16(w.a[s][t].hi, b[p].lo, acc[t][p]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[t][p] = mfma16(w.a[s][t].hi, b[p].hi, acc[t][p]);
        }
    };
    if constexpr (KSN <= 2) {
        compute(first, 0);
    } else {
        static_assert(KSN % 4 == 0, "K-step pairs come in pairs");
        CsW a0 = first, a1;
#pragma unroll
        for (int ks = 0; ks < KSN; ks += 4) {
            cs_wload(a1, bl, wbase, kstot, ks + 2);
            if (BALF_CS_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);     // the requests go out before the MFMAs they hide behind
            compute(a0, ks);
            if (BALF_CS_SCHED >= 2) __builtin_amdgcn_sched_barrier(0);
            if (ks + 4 < KSN) cs_wload(a0, bl, wbase, kstot, ks + 4);
            if (BALF_CS_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
            compute(a1, ks + 2);
            if (BALF_CS_SCHED >= 2) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifndef BALF_CS_G
#define BALF_CS_G 1          // token groups per workgroup at C <= 128 (2: the paired waves of the two groups request the same weight lines in step)
#endif
template <int C> constexpr int cs_groups() { return C <= 128 ? BALF_CS_G : 1; }
template <int C, int CIN, int MODE>
__global__ __launch_bounds__(cs_waves<C>() * cs_groups<C>() * 64, 2) void stage_cs_kernel16(StageArgs A) {
    static_assert(MODE == 0 || MODE == 1, "grid branch / block branch");
    constexpr int NW = cs_waves<C>(), KS = C / 32, KI = CIN / 32, NT = C / 16, P = 4, G = cs_groups<C>();
    constexpr int STAMP_KID = (C == 64 ? 1 : C == 128 ? 2 : 3) * 2 + (MODE & 1); (void)STAMP_KID;   // (diagnostic build only)
    STAMPV_DECL;
    STAMPV(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    // (G = 2) two token groups per workgroup, each with its own LDS image and its own waves; they share the barriers only
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = (G == 1) ? 0 : wave_all / NW;
    unsigned char *smem_raw = smem_all + cs_lut_bytes<MODE>() + grp * cs_lds_bytes<C>();
    h8 *bx = reinterpret_cast<h8 *>(smem_raw);                                            // shared B fragments
    unsigned char *btr = smem_raw + cs_bx_bytes<C>();                                      // token tiles, later u'
    h8 *bu = reinterpret_cast<h8 *>(btr);
    const unsigned char *wmix_l = btr + cs_bt_bytes<C>();                                  // (C = 256) re-ordered Wmix
    float *stats = reinterpret_cast<float *>(smem_raw + cs_bx_bytes<C>() + cs_bt_bytes<C>() +   // [NW][4][16] (sum, sumsq)
                                             (cs_mix_in_lds<C>() ? 8 * 2048 : 0));
    const int lane = threadIdx.x & 63, q = lane >> 4, li = lane & 15;
    const int wave = wave_all - grp * NW;
    unsigned char *bT = btr + wave * kS1BtBytes;
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE == 0 ? 0 : 1];
    constexpr int kGeluChunk = BALF_CS_GELU_CHUNK; (void)kGeluChunk;
    const char *bb = reinterpret_cast<const char *>(blob);
    const CsBlob bl{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(blob), 0, 0x7fffffff, 0x27000), (unsigned)lane * 16u};
    const CsBlob bq{bl.rsrc, (unsigned)q * 16u};                                           // per-channel vectors: channels 4 q .. 4 q + 3
    const int c0 = 32 * wave;                                                              // this wave's first channel

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    // XCD-aware group order (speed only), as in the ring kernels
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xl = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int item = ((xl < xr ? xl * (xq + 1) : xr * (xq + 1) + (xl - xr) * xq) + xj) * G + grp;
    const int n = item / per_img, rem = item - n * per_img;
    const int gy = rem / fw, gx = rem - gy * fw;
    const int ty = li >> 1, tx0 = 4 * (li & 1);
    int y, xl0;
    if (MODE == 0) { y = ty * fh + gy; xl0 = tx0 * fw + gx; }
    else           { y = 8 * gy + ty;  xl0 = 8 * gx + tx0; }
    const int pstep = (MODE == 0) ? fw : 1;
    const long pix0 = ((long)n * H + y) * W + xl0;

    auto barrier = [&]() { lds_barrier(); };                 // lgkmcnt(0) + s_barrier: never drains loads or stores
    // ---- stage input -> shared B fragments (KI x 4 fragments, two per wave); (C = 256) the re-ordered mixing matrix ----
    {
        static_assert(KI * 4 == 2 * NW, "two input fragments per wave");           // holds for C = 64, 128, 256
        HL xin[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int fi = 2 * wave + f, kk = fi >> 2, p = fi & 3;
            xin[f] = load_frag_px(A.X, pix0 + p * pstep, CIN, kk, q);
        }
        STAMPV(17);  // (diagnostic) kernel entry -> geometry and the input loads issued
        if constexpr (cs_gelu_lut<MODE>() && !BALF_ABLATE_LUTCOPY) {     // GELU chord table -> LDS offset 0 (published by the barrier below)
            // (the phase stamps show ~6 k cycles here at C = 64: that is the strided input gather's HBM latency, which these
            // loads queue behind -- requesting all of a thread's chunks before the first store changes nothing)
            for (int i = threadIdx.x; i < cs_lut_bytes<MODE>() / 16; i += NW * G * 64)
                *reinterpret_cast<uint4 *>(smem_all + i * 16) = *reinterpret_cast<const uint4 *>(bb + (size_t)kLayout.gelu_log * 4 + i * 16);
        }
        if constexpr (cs_mix_in_lds<C>()) {
            static_assert(!cs_mix_in_lds<C>() || G == 1, "the LDS copy of the mixing matrix is per workgroup");
            for (int i = threadIdx.x; i < 8 * 2 * 64; i += NW * 64) {
                const int l = i & 63, part = (i >> 6) & 1, tile = i >> 7, pt = tile >> 1, ks = tile & 1;
                const int g = 4 * (l & 15) + pt;
                const int stile = (g >> 4) * 2 + ks, sl = (g & 15) + 16 * (l >> 4);
                *reinterpret_cast<uint4 *>(const_cast<unsigned char *>(wmix_l) + tile * 2048 + part * 1024 + l * 16) =
                    *reinterpret_cast<const uint4 *>(bb + (size_t)Br.mix_w * 4 + stile * 2048 + part * 1024 + sl * 16);
            }
        }
        STAMPV(18);  // (diagnostic) GELU table / mixing matrix copied
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int fi = 2 * wave + f;
            bx[(fi * 2 + 0) * 64 + lane] = xin[f].hi;
            bx[(fi * 2 + 1) * 64 + lane] = xin[f].lo;
        }
    }
    STAMPV(19);  // (diagnostic) input fragments arrived and written to LDS
    barrier();
    STAMPV(1);   // input staged

    struct Bias { f4 b[2]; };
    auto bias_load = [&](int off_floats) {                    // this wave's 32 channels of a bias vector (requested early)
        Bias r;
        r.b[0] = bq.vec((unsigned)(off_floats + c0) * 4u);
        r.b[1] = bq.vec((unsigned)(off_floats + c0 + 16) * 4u);
        return r;
    };
    auto bias_fill = [&](f4 (&t)[2][4], const Bias &b) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) t[nt][p] = b.b[nt];
    };
    auto wptr = [&](int w_off_floats, int row_tile0, int kstot, int ks0) {      // first weight tile of this wave
        return (unsigned)w_off_floats * 4u + (unsigned)(((row_tile0 + 2 * wave) * kstot + ks0) * 2048);
    };
    auto from_bx = [&](int ks) { return bx + ks * (4 * 2 * 64); };
    // per-pixel LayerNorm statistics over ALL channels: this wave's partial sums through LDS (one barrier)
    auto ln_stats_all = [&](const f4 (&x)[2][4], float (&rstd)[P], float (&shift)[P]) {
        // this wave's partial (sum, sumsq) per pixel: 8 values per lane, reduced over the four lane quarters as a
        // reduce-scatter (6 row swaps + 6 adds instead of an all-reduce per pixel tile) -- quarter 0 ends up with
        // (S_0, S_2), quarter 1 with (SS_0, SS_2), quarter 2 with (S_1, S_3), quarter 3 with (SS_1, SS_3) of its
        // pixel column, and every lane writes its pair: stats[wave][quarter][li]
        {
            auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
            auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
            float c[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float s = x[0][p][0], ss = x[0][p][0] * x[0][p][0];     // (not 0 + x: hipcc keeps that add)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = (nt == 0 ? 1 : 0); r < 4; ++r) { s += x[nt][p][r]; ss = fmaf(x[nt][p][r], x[nt][p][r], ss); }
                const auto r0 = __builtin_amdgcn_permlane16_swap(u(s), u(ss), false, false);   // rows [s0 ss0 s2 ss2], [s1 ss1 s3 ss3]
                c[p] = f(r0[0]) + f(r0[1]);                                                    // [S01 SS01 S23 SS23]
            }
            const auto r01 = __builtin_amdgcn_permlane32_swap(u(c[0]), u(c[1]), false, false);  // [c0.lo32 c1.lo32], [c0.hi32 c1.hi32]
            const auto r23 = __builtin_amdgcn_permlane32_swap(u(c[2]), u(c[3]), false, false);
            *reinterpret_cast<float2 *>(stats + (wave * 64 + lane) * 2) =
                make_float2(f(r01[0]) + f(r01[1]), f(r23[0]) + f(r23[1]));
        }
        barrier();
        float S[P] = {0.0f, 0.0f, 0.0f, 0.0f}, SS[P] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float2 a = *reinterpret_cast<const float2 *>(stats + (w * 64 + 0 * 16 + li) * 2);
            const float2 b = *reinterpret_cast<const float2 *>(stats + (w * 64 + 1 * 16 + li) * 2);
            const float2 cc = *reinterpret_cast<const float2 *>(stats + (w * 64 + 2 * 16 + li) * 2);
            const float2 d = *reinterpret_cast<const float2 *>(stats + (w * 64 + 3 * 16 + li) * 2);
            S[0] += a.x; S[2] += a.y; SS[0] += b.x; SS[2] += b.y;
            S[1] += cc.x; S[3] += cc.y; SS[1] += d.x; SS[3] += d.y;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float mean = S[p] * (1.0f / C);
            const float var = fmaf(SS[p], 1.0f / C, -mean * mean);
            rstd[p] = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
            shift[p] = -mean * rstd[p];
        }
    };
    // publish this wave's 32 channels as K-step `wave` of the shared B fragments (callers put the barriers)
    auto publish = [&](h8 *dst, const f4 (&t)[2][4]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const HL v = split8(t[0][p], t[1][p]);
            dst[((wave * 4 + p) * 2 + 0) * 64 + lane] = v.hi;
            dst[((wave * 4 + p) * 2 + 1) * 64 + lane] = v.lo;
        }
    };
    auto ln_publish = [&](const f4 (&x)[2][4]) {               // (x - mean) * rstd -> shared B (affine folded into the weights)
        float rstd[P], shift[P];
        ln_stats_all(x, rstd, shift);                          // its barrier also says: everyone is done reading bx
        f4 yv[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[nt][p][r] = fmaf(x[nt][p][r], rstd[p], shift[p]);
        publish(bx, yv);
        barrier();
    };

    // Every Linear's first weight fragments and bias are requested before the epilogue / barriers in front of it.
    // ---- x0 = relu(conv0(X)) ----
    const unsigned w_c0 = wptr(S.conv0_w, 0, KI, 0), w_q1 = wptr(S.q1_w, (MODE == 0 ? 0 : 1) * NT, KS, 0);
    const unsigned w_d1a = wptr(Br.d1_w, 0, KS, 0), w_d1b = wptr(Br.d1_w, NT, KS, 0), w_d2 = wptr(Br.d2_w, 0, KS, 0);
    f4 x0[2][4];
    {
        CsW w;
        cs_wload<(KI == 1 ? 1 : 2)>(w, bl, w_c0, KI, 0);
        bias_fill(x0, bias_load(S.conv0_b));
        cs_linear<KI>(x0, w, bl, w_c0, KI, from_bx, lane);
    }
    STAMPV(2);   // conv0
    CsW wn;                                                    // the NEXT Linear's first fragments
    cs_wload(wn, bl, w_q1, KS, 0);
    Bias bn = bias_load(S.q1_b + MODE * C);
    relu(x0);
    ln_publish(x0);
    STAMPV(3);   // relu, LN exchange, publish
    // ---- z = GELU(dense1 half) ----
    f4 z[2][4];
    bias_fill(z, bn);
    cs_linear<KS>(z, wn, bl, w_q1, KS, from_bx, lane);
    STAMPV(4);   // dense1 half
    cs_wload(wn, bl, w_d1a, KS, 0);
    bn = bias_load(Br.d1_b);
    cs_gelu<cs_gelu_lut<MODE>(), kGeluChunk>(z);
    ln_publish(z);
    STAMPV(5);   // GELU, LN exchange, publish
    // ---- branch dense1: a half, b half (same B operand) ----
    f4 ga[2][4];
    bias_fill(ga, bn);
    cs_linear<KS>(ga, wn, bl, w_d1a, KS, from_bx, lane);
    STAMPV(6);   // branch dense1, a half
    cs_wload(wn, bl, w_d1b, KS, 0);
    bn = bias_load(Br.d1_b + C);
    cs_gelu<cs_gelu_lut<MODE>(), kGeluChunk>(ga);
    {
        f4 gb[2][4];
        bias_fill(gb, bn);
        cs_linear<KS>(gb, wn, bl, w_d1b, KS, from_bx, lane);
        STAMPV(7);   // GELU(a), b half
        cs_wload(wn, bl, w_d2, KS, 0);                             // dense2's first fragments travel through the token mix
        bn = bias_load(Br.d2_b);
        cs_gelu<cs_gelu_lut<MODE>(), kGeluChunk>(gb);
        float rstd[P], shift[P];
        ln_stats_all(gb, rstd, shift);                         // gating LayerNorm (affine) over all C channels
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f4 gg = bq.vec((unsigned)(Br.gln_g + c0 + 16 * nt) * 4u), be = bq.vec((unsigned)(Br.gln_b + c0 + 16 * nt) * 4u);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[P];
#pragma unroll
                for (int p = 0; p < P; ++p) v[p] = fmaf(fmaf(gb[nt][p][r], rstd[p], shift[p]), gg[r], be[r]);
                h2 h01, l01, h23, l23;
                split_pair(v[0], v[1], h01, l01);
                split_pair(v[2], v[3], h23, l23);
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                unsigned char *row = bT + s1_bt_wr(16 * nt + 4 * q + r, li);
                *reinterpret_cast<h4 *>(row) = h4{h01[0], h01[1], h23[0], h23[1]};
                *reinterpret_cast<h4 *>(row + kS1BtPlane) = h4{l01[0], l01[1], l23[0], l23[1]};
            }
        }
    }
    STAMPV(8);   // GELU(b), gating LN exchange, token tile
    {   // token mix of this wave's 32 channels (wave-local) and the gate
        HL a[2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char *row = bT + s1_bt_rd(16 * ct + li, 4 * kk + q);
                a[ct][kk].hi = *reinterpret_cast<const h8 *>(row);
                a[ct][kk].lo = *reinterpret_cast<const h8 *>(row + kS1BtPlane);
            }
        const f4 mbv = CsBlob{bl.rsrc, (unsigned)li * 16u}.vec((unsigned)Br.mix_b * 4u);
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            HL w0, w1;
            if constexpr (cs_mix_in_lds<C>()) {
                const unsigned char *wl = wmix_l + lane * 16;
                w0.hi = *reinterpret_cast<const h8 *>(wl + (pt * 2 + 0) * 2048);
                w0.lo = *reinterpret_cast<const h8 *>(wl + (pt * 2 + 0) * 2048 + 1024);
                w1.hi = *reinterpret_cast<const h8 *>(wl + (pt * 2 + 1) * 2048);
                w1.lo = *reinterpret_cast<const h8 *>(wl + (pt * 2 + 1) * 2048 + 1024);
            } else {                                            // natural fragments in the blob: column li of tile pt = token 4 li + pt
                const int g = 4 * li + pt;
                const char *wg = bb + (size_t)Br.mix_w * 4 + ((g & 15) + 16 * q) * 16;
                w0.hi = *reinterpret_cast<const h8 *>(wg + ((g >> 4) * 2 + 0) * 2048);
                w0.lo = *reinterpret_cast<const h8 *>(wg + ((g >> 4) * 2 + 0) * 2048 + 1024);
                w1.hi = *reinterpret_cast<const h8 *>(wg + ((g >> 4) * 2 + 1) * 2048);
                w1.lo = *reinterpret_cast<const h8 *>(wg + ((g >> 4) * 2 + 1) * 2048 + 1024);
            }
            const float mb1 = mbv[pt] + 1.0f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f4 m = {mb1, mb1, mb1, mb1};
                m = mfma16x3(a[ct][0], w0, m);
                m = mfma16x3(a[ct][1], w1, m);
                ga[ct][pt] *= m;
            }
        }
    }
    STAMPV(9);   // token mix + gate
    publish(bx, ga);                                           // every wave is past the gating-LN barrier: bx is free
    barrier();
    // ---- branch dense2 + residual ----
    f4 o[2][4];
    bias_fill(o, bn);
    STAMPV(10);  // publish + barrier
    cs_linear<KS>(o, wn, bl, w_d2, KS, from_bx, lane);
    STAMPV(11);  // dense2
    if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            f4 o0 = o[0][p] + z[0][p], o1 = o[1][p] + z[1][p];
            store_frag_px(A.U, pix0 + p * pstep, C, wave, q, split8(o0, o1));
        }
        STAMPV(12);  // residual + u' store
        STAMPV_FLUSH();
    } else {
        const unsigned w_q2 = wptr(S.q2_w, 0, 2 * KS, 0), w_r1 = wptr(S.r1_w, 0, KS, 0), w_r2 = wptr(S.r2_w, 0, KS, 0);
        // this wave's four u' fragments (of KS x 4): written by the grid kernel just before, served from L2 / the
        // Infinity Cache; requested here (not before dense2: 32 more live registers there mean spills)
        static_assert(KS * 4 == 4 * NW, "four u' fragments per wave");
        HL ub[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fi = 4 * wave + f, kk = fi >> 2, p = fi & 3;
            ub[f] = load_frag_px(A.U, pix0 + p * pstep, C, kk, q);
        }
        cs_wload(wn, bl, w_q2, 2 * KS, 0);
        bn = bias_load(S.q2_b);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];
        barrier();                                             // everyone is done with dense2's B operand (and with the token tiles)
        publish(bx, o);                                        // v'
#pragma unroll
        for (int f = 0; f < 4; ++f) {                          // u' -> the token-tile region, as shared B fragments
            const int fi = 4 * wave + f;
            bu[(fi * 2 + 0) * 64 + lane] = ub[f].hi;
            bu[(fi * 2 + 1) * 64 + lane] = ub[f].lo;
        }
        barrier();
        // RSHMAG.dense2 over cat[u', v']: K-steps 0 .. KS-1 from the u' fragments, KS .. 2KS-1 from v'
        f4 x1[2][4];
        bias_fill(x1, bn);
        STAMPV(12);  // u' loads, residual, barrier, publish v' + u', barrier
        cs_linear<2 * KS>(x1, wn, bl, w_q2, 2 * KS,
                          [&](int ks) { return ks < KS ? bu + ks * (4 * 2 * 64) : bx + (ks - KS) * (4 * 2 * 64); }, lane);
        cs_wload(wn, bl, w_r1, KS, 0);
        bn = bias_load(S.r1_b);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x1[nt][p] += x0[nt][p];
                if constexpr (cs_fused<C>())                   // x1 itself: the tail kernel adds x0 and the scaled RCAB branch
                    *reinterpret_cast<f4 *>(A.R + (pix0 + p * pstep) * C + c0 + 16 * nt + 4 * q) = x1[nt][p];
                else
                    *reinterpret_cast<f4 *>(A.R + (pix0 + p * pstep) * C + c0 + 16 * nt + 4 * q) = x1[nt][p] + x0[nt][p];
            }
        STAMPV(13);  // RSHMAG dense2, residual, x1 store
        ln_publish(x1);
        STAMPV(14);  // LN exchange, publish
        f4 m1[2][4];
        bias_fill(m1, bn);
        cs_linear<KS>(m1, wn, bl, w_r1, KS, from_bx, lane);
        STAMPV(15);  // conv1
        if constexpr (!cs_fused<C>()) {
            cs_wload(wn, bl, w_r2, KS, 0);
            bn = bias_load(S.r2_b);
        }
        lrelu(m1);
        f4 t[2][4];
        if constexpr (cs_fused<C>()) {
            // conv2 is linear: its channel means follow from those of its input (SE kernel), the tail kernel recomputes t
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int p = 0; p < P; ++p) t[nt][p] = m1[nt][p];
        } else {
            barrier();                                         // everyone is done with conv1's B operand
            publish(bx, m1);
            barrier();
            bias_fill(t, bn);
            cs_linear<KS>(t, wn, bl, w_r2, KS, from_bx, lane);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f4 s = t[nt][0];
            if constexpr (!cs_fused<C>()) *reinterpret_cast<f4 *>(A.T + pix0 * C + c0 + 16 * nt + 4 * q) = t[nt][0];
#pragma unroll
            for (int p = 1; p < P; ++p) {
                if constexpr (!cs_fused<C>()) *reinterpret_cast<f4 *>(A.T + (pix0 + p * pstep) * C + c0 + 16 * nt + 4 * q) = t[nt][p];
                s += t[nt][p];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s[r]))));
            if (li == 0) *reinterpret_cast<f4 *>(A.partial + (long)item * C + c0 + 16 * nt + 4 * q) = s;
        }
        STAMPV(16);  // lrelu, (stage 4: exchange, conv2, T store), channel sums
        STAMPV_FLUSH();
    }
}
