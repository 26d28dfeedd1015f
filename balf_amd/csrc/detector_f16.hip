// BALF detector forward on gfx950, f16-MFMA path with split operands (BALF_PREC_FP16).
//
// Same network, same kernel decomposition and the same "pixel on the lane" register chain as the fp32
// path (detector.hip; reference: /root/reference/balf/model/mlp_ma_decoder.py:223-285,
// /root/reference/balf/model/decoder.py:16-30), but every Linear runs on v_mfma_f32_16x16x32_f16:
//
//   * each operand value v is split into two halves, hi = f16(v) and lo = f16(v - hi), and a product is
//     accumulated (in fp32) as  hi*hi' + lo*hi' + hi*lo'  -- three MFMAs per tile, ~2^-20 relative
//     operand error, i.e. fp32-grade results (plain f16 operands measure 3.5e-3 max-abs error on the
//     score map, outside the 1e-4 tolerance);
//   * three 16x16x32 f16 MFMAs (3 x 16 cycles) replace eight 16x16x4 f32 MFMAs (8 x 32 cycles), and --
//     unlike the f32 MFMA -- they execute beside VALU work instead of in place of it;
//   * LayerNorm / GELU / softmax / accumulation / NMS stay fp32.
//
// Fragment layouts (16x16x32 f16): A lane (row = l&15, q = l>>4) holds k = 8q + j, j = 0..7; B lane
// (col = l&15, q) holds k = 8q + j; C/D as the f32 MFMA (col = l&15, row = 4q + r).  A K-step covers 32
// input channels = two accumulator tiles (2s, 2s+1) of the producing Linear; k-slot (q, j) carries
// channel 32s + 16(j>>2) + 4q + (j&3), which is the register the accumulator layout left it in, so the
// chain still needs no shuffle.  Weights are packed on the host in that order (weights.hip).
//
// Activations that are consumed as B operands straight from HBM (stage inputs X2..X4 and the grid-branch
// output U) are stored pre-split in "fragment format": per pixel, per K-step 128 B = [hi: q0..q3 x 8
// halves][lo: q0..q3 x 8 halves] -- the same bytes per pixel as fp32 NHWC.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "det_common.h"
#include "split16.h"

namespace balf {
namespace {

// two accumulator tiles (channels 16*2s + 4q + r and 16*(2s+1) + 4q + r) -> one K-step B fragment
template <int MIX = BALF_SPLIT_MIX>
__device__ __forceinline__ HL split8(const f4 &t0, const f4 &t1) {
    HL o;
    h2 h, l;
    split_pair<MIX>(t0[0], t0[1], h, l); o.hi[0] = h[0]; o.hi[1] = h[1]; o.lo[0] = l[0]; o.lo[1] = l[1];
    split_pair<MIX>(t0[2], t0[3], h, l); o.hi[2] = h[0]; o.hi[3] = h[1]; o.lo[2] = l[0]; o.lo[3] = l[1];
    split_pair<MIX>(t1[0], t1[1], h, l); o.hi[4] = h[0]; o.hi[5] = h[1]; o.lo[4] = l[0]; o.lo[5] = l[1];
    split_pair<MIX>(t1[2], t1[3], h, l); o.hi[6] = h[0]; o.hi[7] = h[1]; o.lo[6] = l[0]; o.lo[7] = l[1];
    return o;
}

// pre-split activation row in HBM ("fragment format"): pixel base + ks*128 + {0: hi, 64: lo} + q*16 bytes
__device__ __forceinline__ HL load_frag_px(const float *base, long pix, int C, int ks, int q) {
    if (BALF_ABLATE_LOADLAT) pix &= 4095;
    const char *p = reinterpret_cast<const char *>(base) + pix * (long)C * 4 + ks * 128 + q * 16;
    HL o;
    o.hi = *reinterpret_cast<const h8 *>(p);
    o.lo = *reinterpret_cast<const h8 *>(p + 64);
    return o;
}

__device__ __forceinline__ void store_frag_px(float *base, long pix, int C, int ks, int q, const HL &v) {
    char *p = reinterpret_cast<char *>(base) + pix * (long)C * 4 + ks * 128 + q * 16;
    *reinterpret_cast<h8 *>(p) = v.hi;
    *reinterpret_cast<h8 *>(p + 64) = v.lo;
}

// ------------------------------------------------------------------------------------------------
// GEMM on split-f16 fragments.  Weight tile (nt, ks) = 2 KiB: [hi: 64 lanes x 16 B][lo: 64 lanes x 16 B].
// Two register stages, each loaded one compute block ahead (see detector.hip for the rationale).
// ------------------------------------------------------------------------------------------------
template <int NTT, int NT0, int NTC, int P, typename BL>
__device__ __forceinline__ void gemm16_chunk(f4 (&acc)[NTT][P], const float *w, int wnt0, int KStot, int ks0, int ksn,
                                             int lane, BL bload) {
    const char *wbase = reinterpret_cast<const char *>(w) + ((size_t)(wnt0 + NT0) * KStot + ks0) * 2048;
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned nstride = (unsigned)KStot * 2048u;
    auto wload = [&](int nt, int kk) {
        const char *p = wbase + ((unsigned)nt * nstride + (unsigned)kk * 2048u) + lane_off;
        HL o;
        o.hi = *reinterpret_cast<const h8 *>(p);
        o.lo = *reinterpret_cast<const h8 *>(p + 1024);
        return o;
    };
    auto compute = [&](const HL (&a)[NTC], const HL (&b)[P]) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) if (!BALF_DROP_WLO) acc[NT0 + nt][p] = mfma16(a[nt].lo, b[p].hi, acc[NT0 + nt][p]);
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[NT0 + nt][p] = mfma16(a[nt].hi, b[p].lo, acc[NT0 + nt][p]);
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[NT0 + nt][p] = mfma16(a[nt].hi, b[p].hi, acc[NT0 + nt][p]);
    };
    HL a0[NTC], a1[NTC], b0[P], b1[P];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) a0[nt] = wload(nt, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) b0[p] = bload(0, p);
    if (ksn == 1) {              // K = 32
        compute(a0, b0);
        return;
    }
    for (int kk = 0; kk < ksn; kk += 2) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) a1[nt] = wload(nt, kk + 1);
#pragma unroll
        for (int p = 0; p < P; ++p) b1[p] = bload(kk + 1, p);
        __builtin_amdgcn_sched_barrier(0);
        compute(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = (kk + 2 < ksn) ? kk + 2 : kk;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) a0[nt] = wload(nt, k2);
#pragma unroll
        for (int p = 0; p < P; ++p) b0[p] = bload(k2, p);
        __builtin_amdgcn_sched_barrier(0);
        compute(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NTT, int NT0, int CH, int P, typename BL>
__device__ __forceinline__ void gemm16_from(f4 (&acc)[NTT][P], const float *w, int wnt0, int KStot, int ks0, int ksn,
                                            int lane, BL bload) {
    if constexpr (NT0 < NTT) {
        constexpr int NTC = (NTT - NT0) < CH ? (NTT - NT0) : CH;
        gemm16_chunk<NTT, NT0, NTC, P>(acc, w, wnt0, KStot, ks0, ksn, lane, bload);
        gemm16_from<NTT, NT0 + NTC, CH, P>(acc, w, wnt0, KStot, ks0, ksn, lane, bload);
    }
}

template <int NTT, int P, typename BL>
__device__ __forceinline__ void gemm16(f4 (&acc)[NTT][P], const float *w, int wnt0, int KStot, int ks0, int ksn,
                                       int lane, BL bload) {
    constexpr int CH = (P >= 4) ? 2 : 4;
    gemm16_from<NTT, 0, CH, P>(acc, w, wnt0, KStot, ks0, ksn, lane, bload);
}


// LDS-only workgroup barrier: waits for this wave's LDS traffic, never for its loads or stores (__syncthreads() emits
// vmcnt(0)).  ONE asm statement with a memory clobber: the raw s_barrier builtin is IntrNoMem, so the compiler could
// otherwise move LDS accesses across it.
__device__ __forceinline__ void lds_barrier() {
    if (BALF_ABLATE_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#if BALF_STAMPS
__device__ unsigned long long g_stamp_sum[16][40];
__device__ unsigned long long g_stamp_cnt[16];
#define STAMP_DECL unsigned long long st_prev = 0; (void)st_prev
#define STAMP(i)                                                                                        \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long st_now;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (threadIdx.x == 0) {                                                                         \
            if ((i) > 0) atomicAdd(&g_stamp_sum[STAMP_KID][(i)], st_now - st_prev);                     \
            else atomicAdd(&g_stamp_cnt[STAMP_KID], 1ull);                                              \
        }                                                                                               \
        st_prev = st_now;                                                                               \
    } while (0)
// The same for straight-line kernels, without memory traffic between the stamps (an atomic in flight would be waited
// for by the next vmcnt wait of the code under test): lane i of wave 0 keeps the cycles of phase i in a register, one
// batch of atomics at the end.
#define STAMPV_DECL unsigned long long sv_prev = 0; unsigned sv_vec = 0; (void)sv_prev
#define STAMPV(i)                                                                                       \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long sv_now;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv_now)::"memory");                  \
        {                                                                                               \
            const unsigned sv_d = __builtin_amdgcn_readfirstlane((unsigned)(sv_now - sv_prev));         \
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(sv_vec) : "s"(sv_d), "n"(i));              \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        sv_prev = sv_now;                                                                               \
    } while (0)
#define STAMPV_FLUSH()                                                                                  \
    do {                                                                                                \
        if ((threadIdx.x & 63) == 0) {      /* where the waves of a workgroup land: SIMD id per wave slot */ \
            unsigned hw;                                                                                \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                            \
            atomicAdd(&g_stamp_sum[8 + STAMP_KID][((threadIdx.x >> 6) & 7) * 4 + ((hw >> 4) & 3)], 1ull); \
        }                                                                                               \
        if (threadIdx.x < 40) {                                                                         \
            if (threadIdx.x > 0) atomicAdd(&g_stamp_sum[STAMP_KID][threadIdx.x], (unsigned long long)sv_vec); \
            else atomicAdd(&g_stamp_cnt[STAMP_KID], 1ull);                                              \
        }                                                                                               \
    } while (0)
// the head kernel's phases: an array of its own (the sixteen rows above are taken by the stage kernels)
__device__ unsigned long long g_head_stamp_sum[40];
__device__ unsigned long long g_head_stamp_cnt;
#define STAMPV_FLUSH_HEAD()                                                                             \
    do {                                                                                                \
        if (threadIdx.x < 40) {                                                                         \
            if (threadIdx.x > 0) atomicAdd(&g_head_stamp_sum[threadIdx.x], (unsigned long long)sv_vec); \
            else atomicAdd(&g_head_stamp_cnt, 1ull);                                                    \
        }                                                                                               \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMPV_DECL
#define STAMPV(i)
#define STAMPV_FLUSH()
#define STAMPV_FLUSH_HEAD()
#endif
// add the value of lane (l + n) mod 16 of the same 16-lane row (DPP row_ror): 4 steps = sum over the row in every lane
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, r);
}

// ------------------------------------------------------------------------------------------------
// Stage-4 tail + detector head: a workgroup = 64 pixels of the 1/8-resolution map.  (Round 1's head kernel gave every
// wave its own 16 pixels and let it stream both weight matrices -- 256 KB + 80 KB of split-f16 fragments -- through L2 for
// them: 21 KB of weight traffic per pixel, waves parked or issue-stalled 91 % of the time.)  conv2 (256 -> 256) is split
// by OUTPUT channels: wave w computes row tiles 4w .. 4w+3 for all four pixel
// tiles, so the workgroup reads conv2 once (256 KB per 64 pixels instead of per 16) and every 8 KB of weights feeds
// 48 MFMAs.  x2 = t*s + r is staged once as shared B fragments in LDS (64 KB), the conv2 output goes back through the
// same buffer (wave w owns K-steps 2w, 2w+1 of it), the 65-way head Linear + BatchNorm + softmax + pixel shuffle
// stay per pixel tile.  Reference: Down.forward tail, mlp_ma_decoder.py:241-244; DetectorHead, decoder.py:16-30.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void head_kernel16_ns(HeadArgs A) {
    constexpr int C = 256, KS = 8, HT = kHeadNPad / 16;
    STAMPV_DECL;
    STAMPV(0);
    __shared__ __attribute__((aligned(16))) h8 xs[4 * KS * 2 * 64];          // [tile][K-step][hi|lo][lane]: 64 KiB
    const int lane = threadIdx.x & 63, q = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const long hw = (long)A.h * A.w;
    const long pixel = ((long)blockIdx.x * 4 + wave) * 16 + li;               // this wave's own pixel tile
    const int n = (int)(pixel / hw);
    const long o = pixel - (long)n * hw;
    const int i = (int)(o / A.w), j = (int)(o - (long)i * A.w);

    float rng = 0.0f;                                                     // max |v| over everything this lane splits (status block)
    // ---- x2 = t * s + r of the wave's 16 pixels -> shared B fragments ----
    {
        h8 *slot = xs + wave * (KS * 2 * 64);
#pragma unroll
        for (int half = 0; half < 2; ++half) {                                 // two batches of 4 K-steps: 16 loads in flight
            f4 tv[4][2], rv[4][2];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int c0 = 32 * (4 * half + k4) + 4 * q;
                tv[k4][0] = ldg4(A.T + pixel * C + c0);      tv[k4][1] = ldg4(A.T + pixel * C + c0 + 16);
                rv[k4][0] = ldg4(A.R + pixel * C + c0);      rv[k4][1] = ldg4(A.R + pixel * C + c0 + 16);
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int kk = 4 * half + k4, c0 = 32 * kk + 4 * q;
                const f4 v0 = tv[k4][0] * ldg4(A.scale + (long)n * C + c0) + rv[k4][0];
                const f4 v1 = tv[k4][1] * ldg4(A.scale + (long)n * C + c0 + 16) + rv[k4][1];
#pragma unroll
                for (int r = 0; r < 4; ++r) rng = range_max(rng, v0[r], v1[r]);
                const HL v = split8(v0, v1);
                slot[(kk * 2 + 0) * 64 + lane] = v.hi;
                slot[(kk * 2 + 1) * 64 + lane] = v.lo;
            }
        }
    }
    STAMPV(1);   // t, r, scale loaded; x2 split and staged
    // ---- conv2: row tiles 4w .. 4w+3 x four pixel tiles; weight fragments one K-step ahead in registers ----
    // (bias and the first K-step's fragments are requested in FRONT of the barrier that publishes x2: an L2 round trip less
    // behind it)
    f4 f[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 b = ldg4(blob + A.off.conv2_b + 16 * (4 * wave + t) + 4 * q);
#pragma unroll
        for (int p = 0; p < 4; ++p) f[t][p] = b;
    }
    const char *wb2 = reinterpret_cast<const char *>(blob + A.off.conv2_w) + ((size_t)(4 * wave) * KS) * 2048 + lane * 16;
    auto wload2 = [&](HL (&a)[4], int kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const char *p = wb2 + ((size_t)t * KS + kk) * 2048;
            a[t].hi = *reinterpret_cast<const h8 *>(p);
            a[t].lo = *reinterpret_cast<const h8 *>(p + 1024);
        }
    };
    HL cr[3][4];                                        // a ring of three K-steps of this wave's four row tiles (96 registers)
    wload2(cr[0], 0);
    wload2(cr[1], 1);
    __syncthreads();
    STAMPV(2);   // barrier
    {
        auto wload = wload2;
        auto compute = [&](const HL (&a)[4], int kk) {
            HL b[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                b[p].hi = xs[(p * KS + kk) * 2 * 64 + lane];
                b[p].lo = xs[(p * KS + kk) * 2 * 64 + 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) if (!BALF_DROP_WLO) f[t][p] = mfma16(a[t].lo, b[p].hi, f[t][p]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) f[t][p] = mfma16(a[t].hi, b[p].lo, f[t][p]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p) f[t][p] = mfma16(a[t].hi, b[p].hi, f[t][p]);
        };
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 2 < KS) wload(cr[(kk + 2) % 3], kk + 2);     // two K-steps (768 cycles of this wave's matrix work) ahead
            __builtin_amdgcn_sched_barrier(0);
            compute(cr[kk % 3], kk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    STAMPV(3);   // conv2 (384 MFMAs per wave, weights from L2)
    __syncthreads();                                  // everyone is done reading x2: the buffer takes the conv2 output
    STAMPV(4);   // barrier
    f4 z[HT][1], bn_al[HT], bn_be[HT];
    // (the head Linear's bias, weight pointers and first weight fragments: requested here, in front of the relu / split / barrier
    // section, which covers their L2 round trip)
    f4 za[4], zb;                                       // row tile `wave` x pixel tiles 0..3; row tile 4 x pixel tile `wave`
    {
        const f4 ba = ldg4(blob + A.head_b + 16 * wave + 4 * q);
#pragma unroll
        for (int p = 0; p < 4; ++p) za[p] = ba;
        zb = ldg4(blob + A.head_b + 64 + 4 * q);
    }
    const char *wb = reinterpret_cast<const char *>(blob + A.head_w) + lane * 16;
    auto wload = [&](HL (&a)[2], int kk) {              // [0]: row tile `wave`, [1]: row tile 4
        const char *p0 = wb + ((size_t)wave * KS + kk) * 2048, *p1 = wb + ((size_t)4 * KS + kk) * 2048;
        a[0].hi = *reinterpret_cast<const h8 *>(p0); a[0].lo = *reinterpret_cast<const h8 *>(p0 + 1024);
        a[1].hi = *reinterpret_cast<const h8 *>(p1); a[1].lo = *reinterpret_cast<const h8 *>(p1 + 1024);
    };
    HL ar[4][2];                                        // a ring of four K-steps (64 registers: conv2's accumulators die below)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wload(ar[kk], kk);
    // relu -> B fragments of the head Linear: this wave's 4 row tiles are K-steps 2w, 2w+1 of every pixel tile
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f4 v0 = f[2 * s2][p], v1 = f[2 * s2 + 1][p];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = max0(v0[r]); v1[r] = max0(v1[r]); rng = range_max(rng, v0[r], v1[r]); }
            const HL v = split8(v0, v1);
            xs[(p * KS + 2 * wave + s2) * 2 * 64 + lane] = v.hi;
            xs[(p * KS + 2 * wave + s2) * 2 * 64 + 64 + lane] = v.lo;
        }
    STAMPV(5);   // relu, split, staged
    __syncthreads();
    STAMPV(6);   // barrier

    // ---- head Linear 256 -> 65 (padded to 80 = five row tiles), split by OUTPUT rows like conv2 ----
    // (Until round 5 every wave ran all five row tiles on its own pixel tile and pulled all 80 KB of head weights through L2
    // for 16 pixels: 120 MFMAs behind eight exposed L2 round trips -- 16.5 k of the workgroup's 54 k cycles for 1.9 k cycles
    // of matrix work, profiles/r5_stamps_head.txt.)  Wave w computes row tile w for ALL four pixel tiles (96 MFMAs on 16 KB
    // of weights) and row tile 4 for its own pixel tile (24 MFMAs, 16 KB that the four waves read together); the logits then
    // cross the waves through LDS -- the B-fragment buffer is free once every wave has left the Linear -- so that BatchNorm,
    // softmax and the pixel shuffle find a pixel's 80 values in the four lanes that held them before.
    {
        auto compute = [&](const HL (&a)[2], int kk) {
            HL b[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                b[p].hi = xs[(p * KS + kk) * 2 * 64 + lane];
                b[p].lo = xs[(p * KS + kk) * 2 * 64 + 64 + lane];
            }
            HL bw;                                           // the wave's own pixel tile (wave-uniform index: a plain LDS read)
            bw.hi = xs[(wave * KS + kk) * 2 * 64 + lane];
            bw.lo = xs[(wave * KS + kk) * 2 * 64 + 64 + lane];
#pragma unroll
            for (int p = 0; p < 4; ++p) if (!BALF_DROP_WLO) za[p] = mfma16(a[0].lo, b[p].hi, za[p]);
            if (!BALF_DROP_WLO) zb = mfma16(a[1].lo, bw.hi, zb);
#pragma unroll
            for (int p = 0; p < 4; ++p) za[p] = mfma16(a[0].hi, b[p].lo, za[p]);
            zb = mfma16(a[1].hi, bw.lo, zb);
#pragma unroll
            for (int p = 0; p < 4; ++p) za[p] = mfma16(a[0].hi, b[p].hi, za[p]);
            zb = mfma16(a[1].hi, bw.hi, zb);
        };
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            compute(ar[kk & 3], kk);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 4 < KS) wload(ar[kk & 3], kk + 4);     // four K-steps (~1 300 cycles of matrix work of the CU's waves) ahead
            __builtin_amdgcn_sched_barrier(0);
        }
        // BatchNorm's scale and shift of the lane's 20 channels: requested in front of the exchange below
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            bn_al[t] = ldg4(blob + A.head_alpha + 16 * t + 4 * q);
            bn_be[t] = ldg4(blob + A.head_beta + 16 * t + 4 * q);
        }
        __syncthreads();                                    // everyone is done with the B fragments: the buffer takes the logits
        // zs[pixel 0..63][channel 0..79], 84 floats per pixel (16-byte accesses of 16 lanes a pixel apart: conflict-free)
        float *zs = reinterpret_cast<float *>(xs);
        constexpr int ZP = 84;
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f4 *>(zs + (16 * p + li) * ZP + 16 * wave + 4 * q) = za[p];
        *reinterpret_cast<f4 *>(zs + (16 * wave + li) * ZP + 64 + 4 * q) = zb;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < HT; ++t) z[t][0] = *reinterpret_cast<const f4 *>(zs + (16 * wave + li) * ZP + 16 * t + 4 * q);
    }
    STAMPV(7);   // head Linear (120 MFMAs per wave) + logits through LDS
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 al = bn_al[t], be = bn_be[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * q + r;
            z[t][0][r] = z[t][0][r] * al[r] + be[r];
            if (c < kHeadN) {
                mx = fmaxf(mx, z[t][0][r]);
                if (A.logits) A.logits[((long)n * kHeadN + c) * hw + o] = z[t][0][r];
            }
        }
    }
    STAMPV(8);   // BatchNorm, maximum, logits stored
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * q + r;
            const float e = (c < kHeadN) ? expf(z[t][0][r] - mx) : 0.0f;
            z[t][0][r] = e;
            sum += e;
        }
    sum = quarter_allreduce(sum);
    // status block (include/balf_hip.h): a denominator that is not a positive finite number means a non-finite logit -- an operand
    // somewhere upstream left the range of its f16 halves; an input of this kernel at or beyond the largest f16 says it is close
    if (!(sum > 0.0f && sum < INFINITY)) status_raise(A.status, 0 /* BALF_STATUS_SCORE */);
    if (rng >= kF16Max) status_raise(A.status, 1 /* BALF_STATUS_RANGE */);
    const float inv = 1.0f / sum;
    const int Wp = 8 * A.w;
    STAMPV(9);   // 20 exponentials per lane, sums, status
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 pr = z[t][0] * inv;
        float *dst = A.prob + ((long)n * 8 * A.h + 8 * i + 2 * t + (q >> 1)) * Wp + 8 * j + 4 * (q & 1);
        *reinterpret_cast<f4 *>(dst) = pr;
    }
    STAMPV(10);  // normalise, pixel shuffle, prob stored
    STAMPV_FLUSH_HEAD();
}

#include "stage1_f16.h"
#include "stage2_f16.h"
#include "stage_cs_f16.h"
#include "stage3_tail_f16.h"

// Stages 1-3 leave x1 + the hidden-layer channel sums and a tail kernel forms the next stage's input; stage 4 stores t and
// r for the head kernel.
template <int C> constexpr bool stage_fused() { return C == 32 || C == 64 || cs_fused<C>(); }

// ---- dynamic-LDS sizes of every kernel this file launches, and their one-time registration ----
template <int C, int MODE> constexpr int cs_launch_lds() {
    return cs_lds_bytes<C>() * cs_groups<C>() + cs_lut_bytes<MODE>();
}
static_assert(s1_lds_bytes<0>() <= 160 * 1024 && s1_lds_bytes<1>() <= 160 * 1024 && s1_lds_bytes<2>() <= 160 * 1024, "stage-1 LDS image");
static_assert(s2_lds_bytes<0>() <= 160 * 1024 && s2_lds_bytes<1>() <= 160 * 1024 && s2_lds_bytes<2>() <= 160 * 1024, "stage-2 LDS image");
static_assert(cs_launch_lds<128, 0>() <= 160 * 1024 && cs_launch_lds<128, 1>() <= 160 * 1024 && cs_launch_lds<256, 0>() <= 160 * 1024 &&
              cs_launch_lds<256, 1>() <= 160 * 1024, "channel-split LDS image");

template <typename K>
bool allow_lds(K k, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

// hipFuncSetAttribute is a driver round trip (tens of microseconds each, 17 kernels): done once per device of the process
// instead of before every launch -- what a single-image call (the reference's only calling pattern, demo_match.py:29) feels.
int ensure_kernel_attributes() {
    static std::mutex mu;
    static unsigned long long done = 0;              // bit d: device d has its attributes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return BALF_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(mu);
    if (done >> dev & 1) return BALF_OK;
    bool ok = true;
    ok = ok && allow_lds(stage1_kernel16<0, false>, s1_lds_bytes<0>()) && allow_lds(stage1_kernel16<0, true>, s1_lds_bytes<0>());
    ok = ok && allow_lds(stage1_kernel16<1, false>, s1_lds_bytes<1>()) && allow_lds(stage1_kernel16<1, true>, s1_lds_bytes<1>());
    ok = ok && allow_lds(stage1_kernel16<2, false>, s1_lds_bytes<2>()) && allow_lds(stage1_kernel16<2, true>, s1_lds_bytes<2>());
    ok = ok && allow_lds(stage2_kernel16<0>, s2_lds_bytes<0>()) && allow_lds(stage2_kernel16<1>, s2_lds_bytes<1>()) &&
         allow_lds(stage2_kernel16<2>, s2_lds_bytes<2>());
    ok = ok && allow_lds(stage_cs_kernel16<128, 64, 0>, cs_launch_lds<128, 0>()) && allow_lds(stage_cs_kernel16<128, 64, 1>, cs_launch_lds<128, 1>()) &&
         allow_lds(stage3_tail_kernel16, kT3LdsBytes);
    ok = ok && allow_lds(stage_cs_kernel16<256, 128, 0>, cs_launch_lds<256, 0>()) && allow_lds(stage_cs_kernel16<256, 128, 1>, cs_launch_lds<256, 1>());
    if (!ok) return BALF_ERR_LAUNCH;
    done |= 1ull << dev;
    return BALF_OK;
}

// grid of the persistent stage-1 kernels: one workgroup per CU at most (256 CUs on MI355X; any multiple of 8 is correct),
// every wave walks its own list of token groups
inline unsigned s1_blocks(long groups, int waves) {
    long b = (groups + waves - 1) / waves;
    b = (b + 7) / 8 * 8;
    return (unsigned)(b < 256 ? b : 256);
}

template <int C, int CIN>
int run_stage16(int *status, const float *blob, int s, const float *X, const InputU8 &u8, int B, int H, int W, float *U, float *T, float *R,
                float *partial, float *chunk, float *scale, hipStream_t st) {
    StageArgs a{blob, kLayout.st[s], X, u8.p, u8.ch, u8.h, u8.w, u8.top, u8.left, B, H, W, U, T, R, partial};
    const int rows_per_group = (C == 64) ? 2 : 1;           // partial-sum rows per token group (stage 2: one per wave half)
    const int per_img = (H / 8) * (W / 8) * rows_per_group;
    const long groups = (long)B * (H / 8) * (W / 8);
    if constexpr (C == 64) {
        // persistent token-split kernels (stage2_f16.h): one workgroup per CU at most, every wave PAIR walks its own list of groups
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(stage2_kernel16<0>, dim3(s1_blocks(groups, s2_waves<0>() / 2)), dim3(s2_waves<0>() * 64), s2_lds_bytes<0>(), st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(stage2_kernel16<1>, dim3(s1_blocks(groups, s2_waves<1>() / 2)), dim3(s2_waves<1>() * 64), s2_lds_bytes<1>(), st, a));
    } else if constexpr (C == 32) {
        auto g0 = u8.ch ? stage1_kernel16<0, true> : stage1_kernel16<0, false>;
        auto g1 = u8.ch ? stage1_kernel16<1, true> : stage1_kernel16<1, false>;
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(g0, dim3(s1_blocks(groups, s1_waves<0>())), dim3(s1_waves<0>() * 64), s1_lds_bytes<0>(), st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(g1, dim3(s1_blocks(groups, s1_waves<1>())), dim3(s1_waves<1>() * 64), s1_lds_bytes<1>(), st, a));
    } else {
        // channel-split kernels: one workgroup of C/32 waves per token group
        constexpr int G = cs_groups<C>();
        if (groups % G != 0) return BALF_ERR_ARG;          // (H, W multiples of 64: per_img is a multiple of 4 at C <= 128)
        auto c0k = stage_cs_kernel16<C, CIN, 0>;
        auto c1k = stage_cs_kernel16<C, CIN, 1>;
        constexpr int l0 = cs_launch_lds<C, 0>(), l1 = cs_launch_lds<C, 1>();
        const dim3 grid((unsigned)(groups / G)), block(cs_waves<C>() * G * 64);
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(c0k, grid, block, l0, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(c1k, grid, block, l1, st, a));
    }
    BALF_PROF(4 * s + 2, st, {
        hipLaunchKernelGGL(se_reduce_kernel<C>, dim3(B * kSeChunks), dim3(256), 0, st, partial, per_img, chunk);
        hipLaunchKernelGGL(se_kernel<C>, dim3(B), dim3(256), 0, st, blob, kLayout.st[s], chunk,
                           1.0f / ((float)H * (float)W), scale, stage_fused<C>() ? 1 : 0, status);
    });
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

// Tail of stages 2-3 (stage2_kernel16<2> / stage3_tail_kernel16, both persistent): the stage input X, x1 (in R) and the SE scale
// -> the next stage's input.
template <int C>
int run_tail23(int *status, const float *blob, int s, const float *X, const float *R, const float *scale, int B, int H, int W, float *out,
               hipStream_t st) {
    static_assert(C == 64 || C == 128, "stage 4 has no tail kernel: its consumer is the head kernel");
    StageArgs a{blob, kLayout.st[s], X, nullptr, 0, 0, 0, 0, 0, B, H, W, nullptr, nullptr, const_cast<float *>(R), nullptr, scale, out, status};
    const long groups = (long)B * (H / 8) * (W / 8);
    if constexpr (C == 64)      // every wave PAIR walks its own list of groups (stage2_f16.h)
        BALF_PROF(4 * s + 3, st, hipLaunchKernelGGL(stage2_kernel16<2>, dim3(s1_blocks(groups, s2_waves<2>() / 2)), dim3(s2_waves<2>() * 64), s2_lds_bytes<2>(), st, a));
    else                        // every wave owns half a token group (stage3_tail_f16.h)
        BALF_PROF(4 * s + 3, st, hipLaunchKernelGGL(stage3_tail_kernel16, dim3(s1_blocks(2 * groups, kT3Waves)), dim3(kT3Waves * 64), kT3LdsBytes, st, a));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

// Stage-1 tail (stage1_kernel16<2>): x1 (in R), the image and the SE scale -> the next stage's input.
int run_tail16(int *status, const float *blob, const float *X, const InputU8 &u8, const float *R, const float *scale, int B, int H, int W,
               float *out, hipStream_t st) {
    StageArgs a{blob, kLayout.st[0], X, u8.p, u8.ch, u8.h, u8.w, u8.top, u8.left, B, H, W, nullptr, nullptr,
                const_cast<float *>(R), nullptr, scale, out, status};
    auto k = u8.ch ? stage1_kernel16<2, true> : stage1_kernel16<2, false>;
    const long groups = (long)B * (H / 8) * (W / 8);
    BALF_PROF(3, st, hipLaunchKernelGGL(k, dim3(s1_blocks(groups, s1_waves<2>())), dim3(s1_waves<2>() * 64), s1_lds_bytes<2>(), st, a));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

}  // namespace

#if BALF_STAMPS
extern "C" int balf_debug_head_stamps(unsigned long long *sums /*[40]*/, unsigned long long *cnt /*[1]*/, int reset) {
    if (hipMemcpyFromSymbol(sums, HIP_SYMBOL(g_head_stamp_sum), sizeof(g_head_stamp_sum)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_head_stamp_cnt), sizeof(g_head_stamp_cnt)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[40];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_head_stamp_sum), z, sizeof(g_head_stamp_sum)) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_head_stamp_cnt), z, sizeof(g_head_stamp_cnt)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int balf_debug_stamps(unsigned long long *sums /*[16*40]*/, unsigned long long *cnt /*[16]*/, int reset) {
    if (hipMemcpyFromSymbol(sums, HIP_SYMBOL(g_stamp_sum), sizeof(g_stamp_sum)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_stamp_cnt), sizeof(g_stamp_cnt)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long z[16 * 40];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_sum), z, sizeof(g_stamp_sum)) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cnt), z, sizeof(g_stamp_cnt)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int forward_f16(const float *blob, const float *x_nchw_dev, const InputU8 &u8, int B, int Hp, int Wp, float *logits_dev,
                float *prob_dev, char *ws, const Plan &pl, int *status, hipStream_t st) {
    if (const int rc = ensure_kernel_attributes(); rc != BALF_OK) return rc;
    balf_prof::Chain prof_chain;       // the launches below follow each other on `st` with nothing in between

    float *U = reinterpret_cast<float *>(ws + pl.off_U), *T = reinterpret_cast<float *>(ws + pl.off_T),
          *R = reinterpret_cast<float *>(ws + pl.off_R), *partial = reinterpret_cast<float *>(ws + pl.off_partial),
          *chunk = reinterpret_cast<float *>(ws + pl.off_chunk), *scale = reinterpret_cast<float *>(ws + pl.off_scale);
    float *X2 = reinterpret_cast<float *>(ws + pl.off_X[0]), *X3 = reinterpret_cast<float *>(ws + pl.off_X[1]),
          *X4 = reinterpret_cast<float *>(ws + pl.off_X[2]);
    const int h8 = Hp / 8, w8 = Wp / 8;

    for (int b0 = 0; b0 < B; b0 += pl.mb) {
        const int nb = (B - b0 < pl.mb) ? (B - b0) : pl.mb;
        const float *x = x_nchw_dev ? x_nchw_dev + (size_t)b0 * 3 * Hp * Wp : nullptr;
        InputU8 u8b = u8;
        if (u8.ch) u8b.p = u8.p + (size_t)b0 * u8.h * u8.w * u8.ch;
        int rc;
        if ((rc = run_stage16<32, 3>(status, blob, 0, x, u8b, nb, Hp, Wp, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_tail16(status, blob, x, u8b, R, scale, nb, Hp, Wp, X2, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<64, 32>(status, blob, 1, X2, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 2, Wp / 2, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_tail23<64>(status, blob, 1, X2, R, scale, nb, Hp / 2, Wp / 2, X3, st)) != BALF_OK) return rc;
#if BALF_DEBUG_STOP
        if (const char *e = getenv("BALF_DEBUG_STOP_STAGE"); e && atoi(e) == 2) return BALF_OK;
#endif
        if ((rc = run_stage16<128, 64>(status, blob, 2, X3, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 4, Wp / 4, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_tail23<128>(status, blob, 2, X3, R, scale, nb, Hp / 4, Wp / 4, X4, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<256, 128>(status, blob, 3, X4, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, h8, w8, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        HeadArgs ha{blob, kLayout.st[3], kLayout.head_w, kLayout.head_b, kLayout.head_alpha, kLayout.head_beta,
                    T, R, scale, nb, h8, w8,
                    logits_dev ? logits_dev + (size_t)b0 * kHeadN * h8 * w8 : nullptr,
                    prob_dev + (size_t)b0 * Hp * Wp, status};
        BALF_PROF(15, st,
                  hipLaunchKernelGGL(head_kernel16_ns,
                                     dim3((unsigned)((long)nb * h8 * w8 / 64)), dim3(256), 0, st, ha));
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}

}  // namespace balf
