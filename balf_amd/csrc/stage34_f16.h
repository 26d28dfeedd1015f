// Stages 3 and 4 (C = 128 / 256) of the split-f16 detector forward: WAVE-TEAM kernels on v_mfma_f32_32x32x16_f16.
// Included by detector_f16.hip inside balf::{anonymous}, after stage1_f16.h / stage_cs_f16.h whose helpers it shares.
//
// Reference: Down.forward / ResidualSplitHeadMultiAxisGmlpLayer / {Grid,Block}GmlpLayer / RCAB of
// /root/reference/balf/model/mlp_ma_decoder.py:25-149,173-244 at C = 128 / 256.
//
// Why (round 5).  Rounds 2-4 ran these stages on the channel-split kernels of stage_cs_f16.h: C / 32 waves per token group,
// each owning 32 channels of all 64 tokens as 2 x 4 tiles of the 16x16x32 MFMA.  Their counters (profiles/r4_pmc.json):
// matrix pipe busy 0.40-0.50, vector ALU 0.33-0.46, the SUM 0.83-0.88 -- the two pipes of a SIMD hardly overlapped, because
// (a) the 16x16x32 instruction holds the vector issue port for half of its 16 cycles (the 32x32x16 one for a quarter of its
// 32), (b) at C = 256 all eight waves of the CU belonged to ONE workgroup and sat in the same phase between the same
// barriers, (c) every Linear's B operand was read from LDS by C / 32 waves.  Here:
//   * a wave owns 64 CHANNELS of all 64 tokens: 2 row tiles x 2 pixel tiles of the 32x32x16 MFMA (64 accumulator registers
//     per tensor); lane (n = lane & 31, h = lane >> 5), pixel tile p in {0, 1}, token t = 2 n + p (stage 1's token order:
//     ty = n >> 2, tx = 2 (n & 3) + p), register r of row tile rt = channel 64 w + 32 rt + 8 (r >> 2) + 4 h + (r & 3);
//   * a workgroup ("team") is C / 64 waves -- TWO at C = 128, FOUR at C = 256 -- and owns one token group; its LDS image is
//     the GELU table (6 KB) + ONE activation buffer of 64 tokens x C channels of split-f16 B fragments (32 / 64 KB) + a
//     LayerNorm statistics table: 39 / 72 KB, so FOUR / TWO independent teams share a CU and run out of phase: one team's
//     matrix phase beside another's GELU / LayerNorm / split phase on the same SIMD;
//   * the wave's transposed token tile (token mix) ALIASES the wave's own four K-steps of the activation buffer (both are
//     16 KB: 64 channels x 64 tokens x (hi, lo)), which is what makes the image small enough;
//   * two tensors that would not fit the 256 registers of a wave at two waves per SIMD are parked in global memory in
//     register order and come back as accumulator start values: z + dense2.bias (the branch residual) and, in the block
//     kernel, x0 (the stage residual);
//   * weights still stream from L2 straight into registers (a weight fragment is needed by exactly one wave of a team),
//     three K-steps of 16 in flight; GELU from the log-spaced chord table in both kernels.
// HBM formats: the stage input X and the stage's outputs (x1 / t / r as NHWC fp32, partial channel sums) are those of the
// channel-split kernels -- the tail kernel of stage 3 and the head kernel are unchanged --; u' (grid -> block kernel) is in the
// 32x32 fragment format of stage1_f16.h (store_frag32).
#pragma once

// The register tile of a wave: RT row tiles (32 channels each) x P pixel tiles (32 tokens each), RT * P = 4.
//   RT = 2, P = 2: 64 channels of ONE token group (the first form, above);
//   RT = 1, P = 4: 32 channels of TWO token groups (pixel tiles 2 g, 2 g + 1 = group g of the workgroup's pair): a weight
//                  fragment feeds twelve MFMAs instead of six -- half the bytes from L2 per token and, with the same 48
//                  registers of fragments in flight, six K-steps (2 300 cycles) of prefetch instead of three.
#ifndef BALF_TM_RT
#define BALF_TM_RT 1
#endif
constexpr int kTmRT = BALF_TM_RT, kTmP = 4 / kTmRT, kTmG = kTmP / 2;
static_assert(kTmRT == 1 || kTmRT == 2, "");
template <int C> constexpr int tm_waves() { return C / (32 * kTmRT); }
template <int C> constexpr int tm_bx_bytes() { return C * 128 * kTmP; }                   // [K-step of 16][p][hi|lo][64 x 16 B]
constexpr int kTmLutBytes = (kGeluLogEntries * 8 + 15) / 16 * 16;
template <int C> constexpr int tm_stats_bytes() { return tm_waves<C>() * kTmP * 32 * 8; } // [wave][p][n] (sum, sum of squares)
template <int C> constexpr int tm_lds_bytes() { return kTmLutBytes + tm_bx_bytes<C>() + tm_stats_bytes<C>(); }
constexpr int kTmPlane = kTmRT * 32 * 128;     // one plane (hi or lo) of a token tile: 32 RT channel rows x 64 tokens x 2 B
constexpr int kTmStashFloats = 64 * 64;        // one parked tensor of a wave: 64 registers x 64 lanes
template <int C> constexpr int tm_wgs_per_cu() { return (160 * 1024) / tm_lds_bytes<C>() < 8 / tm_waves<C>() ? (160 * 1024) / tm_lds_bytes<C>() : 8 / tm_waves<C>(); }
static_assert(tm_wgs_per_cu<128>() * tm_waves<128>() == 8 && tm_wgs_per_cu<256>() * tm_waves<256>() == 8, "eight waves per CU");

#ifndef BALF_TM_PRIO
#define BALF_TM_PRIO 0
#endif
#ifndef BALF_TM_SPLIT_MIX
#define BALF_TM_SPLIT_MIX 2                   // operand split form (split16.h)
#endif

// eight accumulator registers (the K-slots of one K-step of 16 channels) -> one B fragment
__device__ __forceinline__ HL tm_split8(const f16v &t, int s) {
    HL o;
    h2 hh, ll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        split_pair<BALF_TM_SPLIT_MIX>(t[8 * s + 2 * i], t[8 * s + 2 * i + 1], hh, ll);
        o.hi[2 * i] = hh[0]; o.hi[2 * i + 1] = hh[1]; o.lo[2 * i] = ll[0]; o.lo[2 * i + 1] = ll[1];
    }
    return o;
}

__device__ __forceinline__ void tm_gelu(f16v (&t)[kTmRT][kTmP]) {
    if (BALF_ABLATE_GELU) return;
#pragma unroll
    for (int rt = 0; rt < kTmRT; ++rt)
#pragma unroll
        for (int p = 0; p < kTmP; ++p)
#pragma unroll
            for (int c = 0; c < 16; c += 8) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = t[rt][p][c + i];
                gelu_log_n(v);
#pragma unroll
                for (int i = 0; i < 8; ++i) t[rt][p][c + i] = v[i];
            }
}

// Activations pass through once: loads and stores with the non-temporal hint, so that they do not push the weights -- which
// every CU of the XCD streams again for every token group -- out of the XCD's 4 MB L2.
#ifndef BALF_TM_NT
#define BALF_TM_NT 0      // measured: 8-15 % SLOWER with the hint (s4 block 3.66 -> 4.21 ms per 32 images)
#endif
template <typename T>
__device__ __forceinline__ void tm_store_nt(T *p, const T &v) {
    if (BALF_TM_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <typename T>
__device__ __forceinline__ T tm_load_nt(const T *p) {
    if (BALF_TM_NT) return __builtin_nontemporal_load(p);
    return *p;
}

// weight fragments of ONE K-step of this wave's row tiles (8 registers per row tile)
struct TmW {
    HL a[kTmRT];
};
// tile (rt, ks) at byte wbase + (rt * kstot + ks) * 2048 (+ 1024: lo) of the blob
__device__ __forceinline__ void tm_wload(TmW &w, const CsBlob &bl, unsigned wbase, int kstot, int ks) {
#pragma unroll
    for (int rt = 0; rt < kTmRT; ++rt) {
        const unsigned o = wbase + (unsigned)((rt * kstot + ks) * 2048);
        w.a[rt].hi = bl.frag(o);
        w.a[rt].lo = bl.frag(o + 1024);
    }
}
#ifndef BALF_TM_RING
#define BALF_TM_RING (kTmRT == 1 ? 4 : 3)
#endif
constexpr int kTmRing = BALF_TM_RING;         // K-steps of weight fragments in flight (8 RT registers each)
struct TmPre {                                 // the first kTmRing - 1 K-steps of a Linear, requested by its caller
    TmW w[kTmRing - 1];
};
template <int KSN>
__device__ __forceinline__ void tm_preload(TmPre &pre, const CsBlob &bl, unsigned wbase, int kstot) {
#pragma unroll
    for (int i = 0; i < kTmRing - 1; ++i)
        if (i < KSN) tm_wload(pre.w[i], bl, wbase, kstot, i);
}

// acc[rt][p] += W(row tiles of this wave) . B over KSN K-steps of 16.  `pre`: the fragments of the first K-steps, requested by
// the caller BEFORE the epilogue / barriers in front of this Linear; inside, K-step ks + kTmRing - 1's fragments are requested
// and K-step ks + 1's B fragments read from LDS before the twelve MFMAs of K-step ks.
// bsrc(ks, p, hl) -> the lane's 16 bytes of the team's B fragment (K-step ks, pixel tile p, hi | lo).
template <int KSN, typename BS>
__device__ __forceinline__ void tm_linear(f16v (&acc)[kTmRT][kTmP], const TmPre &pre, const CsBlob &bl, unsigned wbase,
                                          int kstot, BS bsrc, int lane) {
    TmW w[kTmRing];
#pragma unroll
    for (int i = 0; i < kTmRing - 1; ++i) w[i] = pre.w[i];
    // B fragments: ONE set of registers (8 per pixel tile); a pixel tile's three products per row tile run back to back (a
    // dependent chain on one accumulator issues at the full rate) and its fragments of the NEXT K-step are requested right
    // behind them, nine MFMAs (~290 cycles) before their first use
    HL b[kTmP];
    auto bread = [&](HL &d, int ks, int p) {
        d.hi = bsrc(ks, p, 0);
        d.lo = bsrc(ks, p, 1);
    };
#pragma unroll
    for (int p = 0; p < kTmP; ++p) bread(b[p], 0, p);
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
        if (ks + kTmRing - 1 < KSN) tm_wload(w[(ks + kTmRing - 1) % kTmRing], bl, wbase, kstot, ks + kTmRing - 1);
        __builtin_amdgcn_sched_barrier(0);                   // the requests go out before the MFMAs they hide behind
        const TmW &a = w[ks % kTmRing];
#pragma unroll
        for (int p = 0; p < kTmP; ++p) {
#pragma unroll
            for (int rt = 0; rt < kTmRT; ++rt) {
                if (!BALF_DROP_WLO) acc[rt][p] = mfma32(a.a[rt].lo, b[p].hi, acc[rt][p]);
                acc[rt][p] = mfma32(a.a[rt].hi, b[p].lo, acc[rt][p]);
                acc[rt][p] = mfma32(a.a[rt].hi, b[p].hi, acc[rt][p]);
            }
            if (ks + 1 < KSN) bread(b[p], ks + 1, p);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int C, int CIN, int MODE>
__global__ __launch_bounds__(tm_waves<C>() * 64, 2) void stage_tm_kernel16(StageArgs A) {
    static_assert(MODE == 0 || MODE == 1, "grid branch / block branch");
    constexpr int RT = kTmRT, P = kTmP, G = kTmG;
    constexpr int NW = tm_waves<C>(), NKS = C / 16, KI = CIN / 16, NRT = C / 32, WKS = 2 * RT;   // WKS: K-steps a wave's channels make
    static_assert(KI * P == 4 * NW, "four input fragments per wave");                 // holds for C = 128, 256
    constexpr bool FUSED = C <= 128;           // a tail kernel follows (stage 3): x1 and the hidden layer's channel sums leave
    typedef f16v Tile[RT][P];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h8 *bx = reinterpret_cast<h8 *>(smem + kTmLutBytes);                               // the team's B fragments
    // A DS instruction's immediate offset is 16 bits.  At C = 256 with two groups the fragments span 128 KB: left alone hipcc
    // forms the addresses beyond 64 KB in vector registers, keeps them across the Linears and spills ~100 of them.  So K-steps
    // whose fragments lie beyond 64 KB go through a second, opaque base pointer.
    // (a raw LDS address: these kernels have no static __shared__, the dynamic block starts at LDS address 0)
    typedef const h8 __attribute__((address_space(3))) *lds_h8;
    unsigned bx_hi = (unsigned)kTmLutBytes + 65536u + (unsigned)lane * 16u;
    asm("" : "+v"(bx_hi));
    // B fragment (K-step ks, pixel tile p, hi | lo) of the lane
    auto bfrag = [&](int ks, int p, int hl) -> h8 {
        const int off = ((ks * P + p) * 2 + hl) * 1024;
        if (off < 65536) return bx[off / 16 + lane];
        return *reinterpret_cast<lds_h8>(bx_hi + (unsigned)(off - 65536));
    };
    float2 *stats = reinterpret_cast<float2 *>(smem + kTmLutBytes + tm_bx_bytes<C>());
    // this wave's own K-steps WKS w .. of bx (16 KB) double as its transposed token tiles: per group a hi and a lo plane
    unsigned char *tile = smem + kTmLutBytes + wave * (WKS * P * 2048);
    static_assert(WKS * P * 2048 == G * 2 * kTmPlane, "a wave's K-steps and its token tiles are the same 16 KB");
    h8 *bxw = reinterpret_cast<h8 *>(tile);                                            // the same region as B fragments: [2 rt + s][p][hi|lo][lane]
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];
    const char *bb = reinterpret_cast<const char *>(blob);
    const CsBlob bl{__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(blob), 0, 0x7fffffff, 0x27000), (unsigned)lane * 16u};
    const CsBlob bh{bl.rsrc, (unsigned)h * 16u};                                       // per-channel vectors: channels 4 h .. 4 h + 3 (+ 8 g)
    const int c0 = 32 * RT * wave;                                                     // this wave's first channel

    const int H = A.H, W = A.W, fh = H / 8, fw = W / 8;
    const int per_img = fh * fw;
    const int total = A.B * per_img;
    // XCD-aware order of the workgroups' items (speed only): workgroups b and b + 8 share an XCD
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xl = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int witem = (xl < xr ? xl * (xq + 1) : xr * (xq + 1) + (xl - xr) * xq) + xj;
    const int ty = n >> 2, tx0 = 2 * (n & 3);
    const int pstep = (MODE == 0) ? fw : 1;
    int item[G];                               // the workgroup's token groups (an odd total: the last workgroup does its group twice)
    long pixg[G];                              // the lane's two pixels of group g: pixg[g], pixg[g] + pstep
#pragma unroll
    for (int g = 0; g < G; ++g) {
        item[g] = G * witem + g < total ? G * witem + g : total - 1;
        const int img = item[g] / per_img, rem = item[g] - img * per_img;
        const int gy = rem / fw, gx = rem - gy * fw;
        int y, x0p;
        if (MODE == 0) { y = ty * fh + gy; x0p = tx0 * fw + gx; }
        else           { y = 8 * gy + ty;  x0p = 8 * gx + tx0; }
        pixg[g] = ((long)img * H + y) * W + x0p;
    }
    auto pix = [&](int p) { return pixg[p >> 1] + (p & 1) * pstep; };

    auto barrier = [&]() { lds_barrier(); };                 // lgkmcnt(0) + s_barrier: never drains loads or stores
#if BALF_TM_PRIO
    // Two waves that run the SAME program on one SIMD fall into lockstep: while both want the matrix pipe each gets half of it,
    // so they leave their Linear together and meet again at the next one -- matrix phase beside matrix phase, vector phase
    // beside vector phase.  A static priority for the wave in the odd hardware slot of its SIMD breaks the tie once: it takes
    // the pipe, the other falls a phase behind, and from then on one wave's Linear runs beside the other's epilogue.
    {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1u) __builtin_amdgcn_s_setprio(BALF_TM_PRIO);
    }
#endif
    // ---- stage input (16x16 fragment format in HBM) -> the team's B fragments; GELU table -> LDS offset 0 ----
    {
        HL xin[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * wave + i, ks = f / P, p = f % P;
            const char *px = reinterpret_cast<const char *>(A.X) + pix(p) * (long)CIN * 4 + (ks >> 1) * 128 + (2 * (ks & 1) + h) * 16;
            xin[i].hi = tm_load_nt(reinterpret_cast<const h8 *>(px));
            xin[i].lo = tm_load_nt(reinterpret_cast<const h8 *>(px + 64));
        }
        if (!BALF_ABLATE_LUTCOPY)
            for (int i = threadIdx.x; i < kTmLutBytes / 16; i += NW * 64)
                *reinterpret_cast<uint4 *>(smem + i * 16) = *reinterpret_cast<const uint4 *>(bb + (size_t)kLayout.gelu_log * 4 + i * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * wave + i;                      // = ks * P + p (the stage input is at most C / 2 channels: lower 64 KB)
            bx[(f * 2 + 0) * 64 + lane] = xin[i].hi;
            bx[(f * 2 + 1) * 64 + lane] = xin[i].lo;
        }
    }
    barrier();

    struct Bias { f4 b[RT][4]; };
    auto bias_load = [&](int off_floats) {                    // this wave's channels of a bias vector: 16 RT per lane half
        Bias r;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) r.b[rt][g] = bh.vec((unsigned)(off_floats + c0 + 32 * rt + 8 * g) * 4u);
        return r;
    };
    auto bias_fill = [&](Tile &t, const Bias &b) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[rt][p][4 * g + r] = b.b[rt][g][r];
    };
    auto bias_add = [&](Tile &t, const Bias &b) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[rt][p][4 * g + r] += b.b[rt][g][r];
    };
    auto wptr = [&](int w_off_floats, int row_tile0, int kstot, int ks0) {      // first weight tile of this wave
        return (unsigned)w_off_floats * 4u + (unsigned)(((row_tile0 + RT * wave) * kstot + ks0) * 2048);
    };
    auto from_bx = bfrag;
    // parked tensors: register order, 1 KiB per store instruction; every lane reads back what it wrote itself
    float *stash = A.scratch + ((long)blockIdx.x * NW + wave) * (2 * kTmStashFloats);
    auto stash_store = [&](int which, const Tile &t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    tm_store_nt(reinterpret_cast<f4 *>(stash + which * kTmStashFloats + (((rt * P + p) * 4 + g) * 64 + lane) * 4),
                                f4{t[rt][p][4 * g], t[rt][p][4 * g + 1], t[rt][p][4 * g + 2], t[rt][p][4 * g + 3]});
    };
    auto stash_load = [&](int which, Tile &t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f4 v = tm_load_nt(reinterpret_cast<const f4 *>(stash + which * kTmStashFloats + (((rt * P + p) * 4 + g) * 64 + lane) * 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[rt][p][4 * g + r] = v[r];
                }
    };
    // per-token LayerNorm statistics over ALL channels: this wave's partial sums through LDS (one barrier)
    auto ln_stats_all = [&](const Tile &x, float (&rstd)[P], float (&shift)[P]) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float s = x[0][p][0], ss = x[0][p][0] * x[0][p][0];        // (not 0 + x: hipcc keeps that add)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = (rt == 0 ? 1 : 0); r < 16; ++r) { s += x[rt][p][r]; ss = fmaf(x[rt][p][r], x[rt][p][r], ss); }
            half_allreduce2(s, ss);                                     // both lane halves hold the wave's channels now
            stats[(wave * P + p) * 32 + n] = make_float2(s, ss);        // (the two halves store the same pair)
        }
        barrier();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float Sx = 0.0f, SS = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {                              // fixed order: the same bits in every wave
                const float2 v = stats[(w * P + p) * 32 + n];
                Sx += v.x;
                SS += v.y;
            }
            const float mean = Sx * (1.0f / C);
            const float var = fmaf(SS, 1.0f / C, -mean * mean);
            rstd[p] = __builtin_amdgcn_rsqf(max0(var) + kLnEps);
            shift[p] = -mean * rstd[p];
        }
    };
    // publish this wave's channels as K-steps WKS w .. of the team's B fragments (callers put the barriers)
    auto publish = [&](const Tile &t) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const HL v = tm_split8(t[rt][p], s);
                    h8 *dst = bxw + ((2 * rt + s) * P + p) * 2 * 64;
                    dst[lane] = v.hi;
                    dst[64 + lane] = v.lo;
                }
    };
    auto normalize = [&](Tile &y, const Tile &x, const float (&rstd)[P], const float (&shift)[P]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[rt][p][r] = fmaf(x[rt][p][r], rstd[p], shift[p]);
    };
    auto ln_publish = [&](const Tile &x) {                     // (x - mean) * rstd -> B fragments (affine folded into the weights)
        float rstd[P], shift[P];
        ln_stats_all(x, rstd, shift);                          // its barrier also says: everyone is done reading bx
        Tile yv;
        normalize(yv, x, rstd, shift);
        publish(yv);
        barrier();
    };
    // channel sums over each group's 64 tokens of this wave's channels (fixed order) -> one row of `partial` per group
    auto chan_sums = [&](const Tile &t) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float cs[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    // (plain float temporaries: a bit_cast applied to a vector ELEMENT reads the vector's first element, stage2_f16.h)
                    const float ea = t[rt][2 * g][r] + t[rt][2 * g + 1][r], eb = t[rt][2 * g][r + 8] + t[rt][2 * g + 1][r + 8];
                    const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
                    unsigned w0 = sw[0], w1 = sw[1];             // (opaque: hipcc once folded sw[1] into sw[0] here, see stage1_f16.h)
                    asm("" : "+v"(w0), "+v"(w1));
                    const float s = __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
                    cs[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s))));
                }
                if ((lane & 15) == 0) {
                    // row 0: (h 0, regs 0-7): channels 0-3, 8-11; row 1: (h 0, regs 8-15): 16-19, 24-27; row 2: (h 1, regs 0-7):
                    // 4-7, 12-15; row 3: (h 1, regs 8-15): 20-23, 28-31 (+ 32 rt + c0)
                    const int row = lane >> 4, cb = c0 + 32 * rt + 16 * (row & 1) + 4 * (row >> 1);
                    float *pp = A.partial + (long)item[g] * C + cb;
                    *reinterpret_cast<f4 *>(pp) = f4{cs[0], cs[1], cs[2], cs[3]};
                    *reinterpret_cast<f4 *>(pp + 8) = f4{cs[4], cs[5], cs[6], cs[7]};
                }
            }
    };
    auto nhwc_store = [&](float *base, const Tile &t) {        // fp32 NHWC: 16 bytes per (pixel, row tile, g, lane half)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    tm_store_nt(reinterpret_cast<f4 *>(base + pix(p) * C + c0 + 32 * rt + 8 * g + 4 * h),
                                f4{t[rt][p][4 * g], t[rt][p][4 * g + 1], t[rt][p][4 * g + 2], t[rt][p][4 * g + 3]});
    };

    // Every Linear's first K-steps of weight fragments and its bias are requested before the epilogue / barriers in front of it.
    const unsigned w_c0 = wptr(S.conv0_w, 0, KI, 0), w_q1 = wptr(S.q1_w, MODE * NRT, NKS, 0);
    const unsigned w_d1a = wptr(Br.d1_w, 0, NKS, 0), w_d1b = wptr(Br.d1_w, NRT, NKS, 0), w_d2 = wptr(Br.d2_w, 0, NKS, 0);
    TmPre wn;
    Bias bn;
    // ---- x0 = relu(conv0(X)) ----
    {
        Tile x0;
        tm_preload<KI>(wn, bl, w_c0, KI);
        bias_fill(x0, bias_load(S.conv0_b));
        tm_linear<KI>(x0, wn, bl, w_c0, KI, from_bx, lane);
        tm_preload<NKS>(wn, bl, w_q1, NKS);
        bn = bias_load(S.q1_b + MODE * C);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) relu32(x0[rt]);
        ln_publish(x0);
        if constexpr (MODE == 1) {                             // the stage residual comes back as RSHMAG.dense2's start value: x0 + bias
            bias_add(x0, bias_load(S.q2_b));
            stash_store(1, x0);
        }
    }
    // ---- z = GELU(dense1 half): u (grid) / v (block); parked with dense2's bias for the branch residual ----
    {
        Tile z;
        bias_fill(z, bn);
        tm_linear<NKS>(z, wn, bl, w_q1, NKS, from_bx, lane);
        tm_preload<NKS>(wn, bl, w_d1a, NKS);
        tm_gelu(z);
        float rstd[P], shift[P];
        ln_stats_all(z, rstd, shift);                          // its barrier also says: everyone is done reading bx
        {
            Tile yv;
            normalize(yv, z, rstd, shift);
            publish(yv);
        }
        bias_add(z, bias_load(Br.d2_b));
        stash_store(0, z);
        bn = bias_load(Br.d1_b);
        barrier();
    }
    // ---- branch dense1: a half ----
    Tile ga;
    bias_fill(ga, bn);
    tm_linear<NKS>(ga, wn, bl, w_d1a, NKS, from_bx, lane);
    tm_preload<NKS>(wn, bl, w_d1b, NKS);
    bn = bias_load(Br.d1_b + C);
    tm_gelu(ga);
    // ---- b half, gating LayerNorm (affine) over all C channels, transposed token tiles ----
    {
        Tile gb;
        bias_fill(gb, bn);
        tm_linear<NKS>(gb, wn, bl, w_d1b, NKS, from_bx, lane);
        tm_preload<NKS>(wn, bl, w_d2, NKS);                    // dense2's first fragments travel through the token mix
        tm_gelu(gb);
        float rstd[P], shift[P];
        ln_stats_all(gb, rstd, shift);                         // every wave is past dense1's reads of bx: the tiles may land on it
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 gg = bh.vec((unsigned)(Br.gln_g + c0 + 32 * rt + 8 * g) * 4u), be = bh.vec((unsigned)(Br.gln_b + c0 + 32 * rt + 8 * g) * 4u);
#pragma unroll
                for (int grp = 0; grp < G; ++grp)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v0 = fmaf(fmaf(gb[rt][2 * grp][4 * g + r], rstd[2 * grp], shift[2 * grp]), gg[r], be[r]);
                        const float v1 = fmaf(fmaf(gb[rt][2 * grp + 1][4 * g + r], rstd[2 * grp + 1], shift[2 * grp + 1]), gg[r], be[r]);
                        h2 hh, ll;
                        split_pair<BALF_TM_SPLIT_MIX>(v0, v1, hh, ll);
                        // tokens 2 n, 2 n + 1 of channel row c: 4 bytes in each plane (XOR-swizzled 16-byte chunks, stage1_f16.h)
                        unsigned char *row = tile + grp * (2 * kTmPlane) + s1_bt_wr32(32 * rt + 8 * g + 4 * h + r, n);
                        *reinterpret_cast<h2 *>(row) = hh;
                        *reinterpret_cast<h2 *>(row + kTmPlane) = ll;
                    }
            }
    }
    // ---- token mix of this wave's channels (wave-local, per group) and the gate: ga *= Wmix . tile + bias + 1 ----
    {
        const auto mbv = __builtin_amdgcn_raw_buffer_load_b64(bl.rsrc, (unsigned)n * 8u, (unsigned)Br.mix_b * 4u, 0);   // bias of tokens 2 n, 2 n + 1
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            HL wv[4];                                           // mixing matrix, output-token tile pp: B operand (natural K order)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                wv[s].hi = bl.frag((unsigned)Br.mix_w * 4u + (unsigned)((pp * 4 + s) * 2048));
                wv[s].lo = bl.frag((unsigned)Br.mix_w * 4u + (unsigned)((pp * 4 + s) * 2048 + 1024));
            }
            // (a plain temporary: __builtin_bit_cast applied to a vector ELEMENT reads the vector's first element, stage2_f16.h)
            const unsigned mbu = mbv[pp];
            const float mb1 = __builtin_bit_cast(float, mbu) + 1.0f;
#pragma unroll
            for (int grp = 0; grp < G; ++grp)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    HL a[4];                                    // tile rows 32 rt + n: tokens 16 s + 8 h .. + 7 (A operand)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const unsigned char *row = tile + grp * (2 * kTmPlane) + s1_bt_rd(32 * rt + n, 2 * s + h);
                        a[s].hi = *reinterpret_cast<const h8 *>(row);
                        a[s].lo = *reinterpret_cast<const h8 *>(row + kTmPlane);
                    }
                    f16v m;
#pragma unroll
                    for (int r = 0; r < 16; ++r) m[r] = mb1;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        m = mfma32(a[s].lo, wv[s].hi, m);
                        m = mfma32(a[s].hi, wv[s].lo, m);
                        m = mfma32(a[s].hi, wv[s].hi, m);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) ga[rt][2 * grp + pp][r] *= m[r];
                }
        }
    }
    // this wave's tile region is its own K-steps of bx; every wave is past the gating-LN barrier, so nobody reads bx any more
    publish(ga);
    // ---- branch dense2 + residual: the start value is the parked z + bias ----
    Tile o;
    stash_load(0, o);
    barrier();
    if constexpr (MODE == 0) {
        tm_linear<NKS>(o, wn, bl, w_d2, NKS, from_bx, lane);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const HL v = tm_split8(o[rt][p], s);
                    char *up = reinterpret_cast<char *>(A.U) + pix(p) * (long)C * 4 + (WKS * wave + 2 * rt + s) * 64 + h * 16;
                    tm_store_nt(reinterpret_cast<h8 *>(up), v.hi);
                    tm_store_nt(reinterpret_cast<h8 *>(up + 32), v.lo);
                }
    } else {
        const unsigned w_q2u = wptr(S.q2_w, 0, 2 * NKS, 0), w_q2v = wptr(S.q2_w, 0, 2 * NKS, NKS);
        const unsigned w_r1 = wptr(S.r1_w, 0, NKS, 0), w_r2 = wptr(S.r2_w, 0, NKS, 0);
        tm_linear<NKS>(o, wn, bl, w_d2, NKS, from_bx, lane);   // o = v'
        tm_preload<NKS>(wn, bl, w_q2v, 2 * NKS);
        barrier();                                             // everyone is done with dense2's B operand
        publish(o);                                            // v' (dead from here on)
        // RSHMAG.dense2 over cat[u', v'], the v' half first: start value = the parked x0 + bias, K-steps NKS .. 2 NKS - 1
        Tile x1;
        stash_load(1, x1);
        // this wave's share of u' (its own K-steps, all pixel tiles): written by the grid kernel just before, served from L2 /
        // the Infinity Cache; requested in front of the v' half, which covers the round trip
        HL ub[WKS][P];
#pragma unroll
        for (int k = 0; k < WKS; ++k)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const char *up = reinterpret_cast<const char *>(A.U) + pix(p) * (long)C * 4 + (WKS * wave + k) * 64 + h * 16;
                ub[k][p].hi = tm_load_nt(reinterpret_cast<const h8 *>(up));
                ub[k][p].lo = tm_load_nt(reinterpret_cast<const h8 *>(up + 32));
            }
        barrier();
        tm_linear<NKS>(x1, wn, bl, w_q2v, 2 * NKS, from_bx, lane);
        tm_preload<NKS>(wn, bl, w_q2u, 2 * NKS);
        barrier();                                             // everyone is done with v'
#pragma unroll
        for (int k = 0; k < WKS; ++k)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                bxw[((k * P + p) * 2 + 0) * 64 + lane] = ub[k][p].hi;
                bxw[((k * P + p) * 2 + 1) * 64 + lane] = ub[k][p].lo;
            }
        barrier();
        tm_linear<NKS>(x1, wn, bl, w_q2u, 2 * NKS, from_bx, lane);
        tm_preload<NKS>(wn, bl, w_r1, NKS);
        if constexpr (FUSED) {
            nhwc_store(A.R, x1);                               // x1 itself: the tail kernel adds x0 and the scaled RCAB branch
        } else {
            // R = x1 + x0, x0 = the parked value - bias, one pixel tile at a time
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f4 v = tm_load_nt(reinterpret_cast<const f4 *>(stash + kTmStashFloats + (((rt * P + p) * 4 + g) * 64 + lane) * 4));
                        const f4 bq2 = bh.vec((unsigned)(S.q2_b + c0 + 32 * rt + 8 * g) * 4u);
                        f4 r;
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] = x1[rt][p][4 * g + i] + (v[i] - bq2[i]);
                        tm_store_nt(reinterpret_cast<f4 *>(A.R + pix(p) * C + c0 + 32 * rt + 8 * g + 4 * h), r);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bn = bias_load(S.r1_b);
        ln_publish(x1);
        Tile m1;
        bias_fill(m1, bn);
        tm_linear<NKS>(m1, wn, bl, w_r1, NKS, from_bx, lane);
        if constexpr (!FUSED) {
            tm_preload<NKS>(wn, bl, w_r2, NKS);
            bn = bias_load(S.r2_b);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) lrelu32(m1[rt]);
        if constexpr (FUSED) {
            // conv2 is linear: its channel means follow from those of its input (SE kernel), the tail kernel recomputes t
            chan_sums(m1);
        } else {
            barrier();                                         // everyone is done with conv1's B operand
            publish(m1);
            barrier();
            Tile t;
            bias_fill(t, bn);
            tm_linear<NKS>(t, wn, bl, w_r2, NKS, from_bx, lane);
            nhwc_store(A.T, t);
            chan_sums(t);
        }
    }
}
