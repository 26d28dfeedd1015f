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

// wave-private LDS slot: [ks][p][hi|lo][lane] x 16 B
template <int NT, int P>
__device__ __forceinline__ void store_slot16(h8 *slot, const f4 (&t)[NT][P], int lane) {
#pragma unroll
    for (int ks = 0; ks < NT / 2; ++ks)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const HL v = split8(t[2 * ks][p], t[2 * ks + 1][p]);
            slot[((ks * P + p) * 2 + 0) * 64 + lane] = v.hi;
            slot[((ks * P + p) * 2 + 1) * 64 + lane] = v.lo;
        }
}

// pre-split activation row in HBM ("fragment format"): pixel base + ks*128 + {0: hi, 64: lo} + q*16 bytes
#ifndef BALF_ABLATE_LOADLAT
#define BALF_ABLATE_LOADLAT 0     // timing experiment: activation rows come from a small L2-resident window (wrong results)
#endif
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

#ifndef BALF_ALIGN_WAVES
#define BALF_ALIGN_WAVES 0
#endif
template <int NTT, int P, typename BL>
__device__ __forceinline__ void gemm16(f4 (&acc)[NTT][P], const float *w, int wnt0, int KStot, int ks0, int ksn,
                                       int lane, BL bload) {
    constexpr int CH = (P >= 4) ? 2 : 4;
    if (BALF_ALIGN_WAVES && P < 4) __builtin_amdgcn_s_barrier();   // keep the 4 waves on the same weight lines
    gemm16_from<NTT, 0, CH, P>(acc, w, wnt0, KStot, ks0, ksn, lane, bload);
}


// ------------------------------------------------------------------------------------------------
// Cooperative GEMM for C >= 64: weights go HBM/L2 -> LDS ring by LDS-DMA (global_load_lds, no VGPRs), each
// 2 KiB weight tile is fetched ONCE per workgroup and read by all four waves from LDS.  Without it every
// wave streams the whole weight matrix through the CU's 64 B/clk vector-memory path (~170/P B/clk of
// demand when MFMA-bound), which is what bounds the per-wave version at C >= 64.
//   unit u = (chunk c of 4 weight row-tiles, K-step k): 8 KiB = one ring slot; wave w DMA-loads tile w
//   (hi + lo = 2 x 1 KiB wave-instructions).  Ring of 4 slots, 2 units in flight beyond the one in use:
//   iteration u:  wait own loads of unit u (counted vmcnt)  ->  s_barrier (everyone's unit-u loads have
//   landed; everyone is done reading slot (u-1)%4)  ->  DMA unit u+3 into slot (u-1)%4  ->  ds_read + MFMA.
// ------------------------------------------------------------------------------------------------
#ifndef BALF_RING_STRICT
#define BALF_RING_STRICT 0
#endif
constexpr int kRingSlots = 4;
constexpr int kRingNTC = 4;
constexpr int kRingSlotBytes = kRingNTC * 2048;
constexpr int kRingBytes = kRingSlots * kRingSlotBytes;

struct RingGemm {
    const char *wbase;     // first weight tile of the matrix (bytes)
    int wnt0, KStot, ks0, ksn;
    int chunks;            // groups of 4 weight row-tiles (units = chunks * ksn)
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

__device__ __forceinline__ void ring_issue(const RingGemm &g, unsigned char *ring, int u, int wave, int lane) {
    const int c = u / g.ksn, k = u - c * g.ksn;
    const char *src = g.wbase + ((size_t)(g.wnt0 + c * kRingNTC + wave) * g.KStot + g.ks0 + k) * 2048 + lane * 16;
    unsigned char *dst = ring + (u & (kRingSlots - 1)) * kRingSlotBytes + wave * 2048;   // wave-uniform
    __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void *)(src + 1024), (lds_void *)(dst + 1024), 16, 0, 0);
}

// The ring is CHAINED across the consecutive Linears of a kernel: units are numbered globally over the whole
// sequence of GEMMs, and iteration G always issues global unit G + 3, which may belong to one of the next
// GEMMs -- so the first tiles of the next Linear arrive while the current epilogue (GELU, LayerNorm, ...)
// runs, and short GEMMs (2 units at C = 64) do not pay a pipeline refill each.
struct RingChain {         // by value, indexed with compile-time constants only: stays in SGPRs
    RingGemm g[4];          // current GEMM and up to three successors
    int tot[4];             // units of each (0 = absent)
};

template <int NTT>
__device__ __forceinline__ RingChain make_chain(const RingGemm &g0, const RingGemm &g1, const RingGemm &g2,
                                                const RingGemm &g3, int n) {
    RingChain c;
    c.g[0] = g0; c.g[1] = g1; c.g[2] = g2; c.g[3] = g3;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.tot[i] = (i < n) ? c.g[i].chunks * c.g[i].ksn : 0;
    return c;
}

// issue chain-relative unit t (t >= 0 counted from unit 0 of chain.g[0]) into ring slot `slot`
__device__ __forceinline__ void chain_issue(const RingChain &c, unsigned char *ring, int t, int slot, int wave,
                                            int lane) {
    RingGemm d = c.g[0];
    bool valid = c.tot[0] > 0, walking = true;        // step through the chain until t falls inside a GEMM
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (walking) {
            if (valid && t >= c.tot[i]) {
                t -= c.tot[i];
                if (i < 3) { d = c.g[i + 1]; valid = c.tot[i + 1] > 0; }
                else valid = false;
            } else {
                walking = false;
            }
        }
    if (!valid) return;
    const int cc = t / d.ksn, k = t - cc * d.ksn;
    const char *src = d.wbase + ((size_t)(d.wnt0 + cc * kRingNTC + wave) * d.KStot + d.ks0 + k) * 2048 + lane * 16;
    unsigned char *dst = ring + slot * kRingSlotBytes + wave * 2048;                    // wave-uniform
    __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void *)(src + 1024), (lds_void *)(dst + 1024), 16, 0, 0);
}

// Protocol (global unit index G, 4 slots, slot = G & 3); iteration G:
//   s_waitcnt vmcnt  -> this wave's DMA of unit G has landed (units G+1, G+2 may still be in flight);
//   lgkmcnt(0)       -> this wave's ds_reads of unit G-1 have returned (hipcc sinks the MFMAs that consume
//                       them, and their wait, below the barrier; without this the re-fill of that slot raced
//                       with the reads when two workgroups shared a CU);
//   s_barrier        -> every wave's part of unit G is visible, nobody reads slot (G-1) & 3 any more;
//   issue DMA of unit G+3 into slot (G-1) & 3;  ds_read unit G;  MFMAs.
// (A variant that read unit G+1's fragments during unit G's MFMAs measured no faster and costs 32 VGPRs.)
// Other vector-memory operations in the queue.  vmcnt counts loads, stores and LDS-DMA together, in issue order, so a
// batch of E ordinary loads / stores issued between ring iterations a and a+1 sits BEHIND the DMAs of units <= a+3
// (issued three iterations ahead) and in front of all later ones: the waits for units a+1 .. a+3 may leave E more
// operations outstanding -- without that allowance they drain the batch (the u' rows requested just before RSHMAG.dense2,
// the R stores before the RCAB: an ablation build priced them at 5-7 % of the block kernels).  A GEMM is told E and how
// many of the three waits were already USED by the GEMMs since the batch (compile-time tags: VmTag<E, USED>).
#ifndef BALF_RING_VMEXTRA
#define BALF_RING_VMEXTRA 1
#endif
template <int E, int USED> struct VmTag { static constexpr int e = E, used = USED; };
using VmNone = VmTag<0, 3>;

// s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier -- ONE asm statement with a memory clobber: the raw s_barrier builtin is
// IntrNoMem, so the compiler could otherwise move LDS accesses across it
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory");
}

// PER = DMA instructions per ring unit and wave; `after` = units issued after the awaited one (two at most are in flight)
template <int PER, int E>
__device__ __forceinline__ void ring_wait(int after, bool extra) {
    if (BALF_RING_STRICT) { wait_vm_barrier<0>(); return; }
    if (E > 0 && BALF_RING_VMEXTRA && extra) {
        if (after <= 0) wait_vm_barrier<E>();
        else if (after == 1) wait_vm_barrier<PER + E>();
        else wait_vm_barrier<2 * PER + E>();
    } else {
        if (after <= 0) wait_vm_barrier<0>();
        else if (after == 1) wait_vm_barrier<PER>();
        else wait_vm_barrier<2 * PER>();
    }
}

__device__ __forceinline__ void ring_wait_barrier(int after) { ring_wait<2, 0>(after, false); }

template <int NTC>
__device__ __forceinline__ void ring_read(HL (&a)[NTC], const unsigned char *ring, int g, int lane) {
    const unsigned char *slot = ring + (g & (kRingSlots - 1)) * kRingSlotBytes + lane * 16;
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        a[nt].hi = *reinterpret_cast<const h8 *>(slot + nt * 2048);
        a[nt].lo = *reinterpret_cast<const h8 *>(slot + nt * 2048 + 1024);
    }
}

#ifndef BALF_ABLATE_MFMA
#define BALF_ABLATE_MFMA 0
#endif
template <int NTT, int CI, int P>
__device__ __forceinline__ void ring_mfma(f4 (&acc)[NTT][P], const HL (&a)[kRingNTC], const HL (&b)[P]) {
    if (BALF_ABLATE_MFMA) {                      // timing experiment: keep the operands alive, skip the MFMAs
#pragma unroll
        for (int nt = 0; nt < kRingNTC; ++nt) asm volatile("" ::"v"(a[nt].hi), "v"(a[nt].lo));
#pragma unroll
        for (int p = 0; p < P; ++p) asm volatile("" ::"v"(b[p].hi), "v"(b[p].lo));
        return;
    }
#pragma unroll
    for (int nt = 0; nt < kRingNTC; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) if (!BALF_DROP_WLO) acc[CI * kRingNTC + nt][p] = mfma16(a[nt].lo, b[p].hi, acc[CI * kRingNTC + nt][p]);
#pragma unroll
    for (int nt = 0; nt < kRingNTC; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[CI * kRingNTC + nt][p] = mfma16(a[nt].hi, b[p].lo, acc[CI * kRingNTC + nt][p]);
#pragma unroll
    for (int nt = 0; nt < kRingNTC; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[CI * kRingNTC + nt][p] = mfma16(a[nt].hi, b[p].hi, acc[CI * kRingNTC + nt][p]);
}

template <int NTT, int CI, int P, typename VM, typename BL>
__device__ __forceinline__ void chain_chunk(f4 (&acc)[NTT][P], const RingChain &c, unsigned char *ring, int gu,
                                            int lane, int wave, BL bload) {
    if constexpr (CI * kRingNTC < NTT) {
        const int ksn = c.g[0].ksn;
        const int later = c.tot[1] + c.tot[2] + c.tot[3];
        for (int k = 0; k < ksn; ++k) {
            const int u = CI * ksn + k;
            ring_wait<2, VM::e>(c.tot[0] - 1 - u + later, u + VM::used < 3);
            chain_issue(c, ring, u + kRingSlots - 1, (gu + u + kRingSlots - 1) & (kRingSlots - 1), wave, lane);
            HL a[kRingNTC], b[P];
            ring_read<kRingNTC>(a, ring, gu + u, lane);
#pragma unroll
            for (int p = 0; p < P; ++p) b[p] = bload(k, p);
            ring_mfma<NTT, CI, P>(acc, a, b);
        }
        chain_chunk<NTT, CI + 1, P, VM>(acc, c, ring, gu, lane, wave, bload);
    }
}

// Run chain.g[0].  `gu` = global index of its unit 0 (advanced here).  All four waves call this together;
// the first three units of the kernel's sequence were issued in the prologue.
template <int NTT, int P, typename VM = VmNone, typename BL>
__device__ __forceinline__ void gemm16_chain(f4 (&acc)[NTT][P], const RingChain &c, int &gu, int lane, int wave,
                                             unsigned char *ring, BL bload) {
    static_assert(NTT % kRingNTC == 0, "row tiles must come in groups of 4");
    chain_chunk<NTT, 0, P, VM>(acc, c, ring, gu, lane, wave, bload);
    gu += c.tot[0];
}

#ifndef BALF_RAW_LDS_BARRIER
#define BALF_RAW_LDS_BARRIER 1
#endif
#ifndef BALF_MIX_RING
#define BALF_MIX_RING 1
#endif
__device__ __forceinline__ void lds_barrier() {
    if (BALF_RAW_LDS_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}

// Stage 1 (C = 32): every Linear is ONE K-step of two weight row-tiles.  Their fragments are loaded one Linear
// ahead (during the previous epilogue) so that no L2 round trip sits between an epilogue and the next MFMAs.
struct WPre {
    HL a[2];
};

__device__ __forceinline__ WPre wpre_load(const RingGemm &d, int lane) {
    const char *p = d.wbase + ((size_t)d.wnt0 * d.KStot + d.ks0) * 2048 + lane * 16;
    WPre w;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        w.a[nt].hi = *reinterpret_cast<const h8 *>(p + (size_t)nt * d.KStot * 2048);
        w.a[nt].lo = *reinterpret_cast<const h8 *>(p + (size_t)nt * d.KStot * 2048 + 1024);
    }
    return w;
}

template <int P, typename BL>
__device__ __forceinline__ void gemm16_single(f4 (&acc)[2][P], const WPre &w, BL bload) {
    HL b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) b[p] = bload(0, p);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) if (!BALF_DROP_WLO) acc[nt][p] = mfma16(w.a[nt].lo, b[p].hi, acc[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[nt][p] = mfma16(w.a[nt].hi, b[p].lo, acc[nt][p]);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[nt][p] = mfma16(w.a[nt].hi, b[p].hi, acc[nt][p]);
}

#ifndef BALF_ABLATE_STORE
#define BALF_ABLATE_STORE 0
#endif
#ifndef BALF_ABLATE_MIX
#define BALF_ABLATE_MIX 0
#endif
#ifndef BALF_ABLATE_SPLIT
#define BALF_ABLATE_SPLIT 0
#endif
// In-kernel stamps (diagnostic build -DBALF_STAMPS=1 only): wave 0 / lane 0 of every workgroup adds the cycles
// between consecutive STAMP(i) points to g_stamp_sum[kernel][i]; balf_debug_stamps() reads them back.
#ifndef BALF_STAMPS
#define BALF_STAMPS 0
#endif
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
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMPV_DECL
#define STAMPV(i)
#define STAMPV_FLUSH()
#endif
// add the value of lane (l + n) mod 16 of the same 16-lane row (DPP row_ror): 4 steps = sum over the row in every lane
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, r);
}

constexpr int kBtPitch16 = kTokens + 8;        // halves per channel row of the transposed token tile

#ifndef BALF_COOP_MIN_C
#define BALF_COOP_MIN_C 64
#endif
template <int C>
constexpr bool use_ring() { return C >= BALF_COOP_MIN_C; }

// Per-channel parameters of a stage kernel cached in LDS (so that no ordinary VMEM load sits between the
// chained LDS-DMA prefetches and their consumers: vmcnt completes in order).  Offsets in floats:
template <int C> constexpr int par_floats() { return 10 * C + 64; }
enum ParOff { kParConv0B = 0, kParQ1B = 1, kParD1B = 2, kParGlnG = 4, kParGlnB = 5, kParD2B = 6, kParMixB = 7,
              kParQ2B = 7, kParR1B = 8, kParR2B = 9 };      // x C  (kParQ2B.. are + 64 past kParMixB)

template <int C, int P>
constexpr int stage_lds_bytes16() {
    constexpr int slots = 4 * (C / 32) * P * 2048;
    constexpr int bt = 2 * P * C * kBtPitch16 * 2;
    return (slots > bt ? slots : bt) + 4 * C * 4 + (use_ring<C>() ? kRingBytes : 0) + par_floats<C>() * 4;
}

template <int C, int CIN, int MODE>
__global__ __launch_bounds__(256, StageP<C>::OCC) void stage_branch_kernel16(StageArgs A) {
    constexpr bool PK = (MODE == 0) && (C == 32);      // packed VALU math only where it measured faster
    constexpr int STAMP_KID = (C == 32 ? 0 : C == 64 ? 1 : C == 128 ? 2 : 3) * 2 + MODE; (void)STAMP_KID;
    STAMP_DECL;
    constexpr int P = StageP<C>::P;
    constexpr int NT = C / 16, KS = C / 32;
    STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int main_bytes =
        stage_lds_bytes16<C, P>() - 4 * C * 4 - (use_ring<C>() ? kRingBytes : 0) - par_floats<C>() * 4;
    _Float16 *bT = reinterpret_cast<_Float16 *>(smem_raw);                 // [hi|lo][p][c][pitch]
    // (smem_raw + main_bytes: 4 * C floats, formerly the cross-wave channel-sum scratch; kept so the LDS image is unchanged)
    unsigned char *ring = smem_raw + main_bytes + 4 * C * 4;               // weight ring (C >= 64)
    float *par = reinterpret_cast<float *>(smem_raw + main_bytes + 4 * C * 4 + (use_ring<C>() ? kRingBytes : 0));

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    h8 *slot = reinterpret_cast<h8 *>(smem_raw) + wave * (KS * P * 2 * 64);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    const int H = A.H, W = A.W;
    const int cols = W / 8 / P;
    const int per_img = (H / 8) * cols;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the blocks that share an L2).  Give
    // each XCD a contiguous range of work items so that neighbouring items -- which read the same cache lines
    // of the strided stage-1 gather -- hit one L2 instead of eight.  Speed only: any placement is correct.
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xl = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int item = (xl < xr ? xl * (xq + 1) : xr * (xq + 1) + (xl - xr) * xq) + xj;
    const int n = item / per_img;
    const int rem = item - n * per_img;
    const int iy0 = rem / cols, ix0 = (rem - iy0 * cols) * P;
    const int tok = 16 * wave + li, ty = tok >> 3, tx = tok & 7;
    long pix[P];
    int py[P], px_[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int y, x;
        if (MODE == 0) { y = ty * (H / 8) + iy0; x = tx * (W / 8) + ix0 + p; }
        else           { y = 8 * iy0 + ty;       x = 8 * (ix0 + p) + tx; }
        pix[p] = ((long)n * H + y) * W + x;
        py[p] = y; px_[p] = x;
    }

    // the Linears of this kernel in execution order (the weight ring prefetches across them)
    constexpr int NG = (MODE == 0) ? 6 : 10;
    // ring-wait allowances (VmTag): u' loads per wave, R stores per wave, ring units of one C -> C Linear
    constexpr int EU = KS * P * 2, ER = NT * P, UG = (NT / kRingNTC > 0 ? NT / kRingNTC : 1) * KS;
    const char *bb = reinterpret_cast<const char *>(blob);
    RingGemm seq[NG + 3];
    {
        constexpr int CH = NT / kRingNTC > 0 ? NT / kRingNTC : 1;
        constexpr int KI = CIN / 32 > 0 ? CIN / 32 : 1;
        int i = 0;
        seq[i++] = RingGemm{bb + (size_t)S.conv0_w * 4, 0, KI, 0, KI, CH};
        seq[i++] = RingGemm{bb + (size_t)S.q1_w * 4, MODE * NT, KS, 0, KS, CH};
        seq[i++] = RingGemm{bb + (size_t)Br.d1_w * 4, 0, KS, 0, KS, CH};
        seq[i++] = RingGemm{bb + (size_t)Br.d1_w * 4, NT, KS, 0, KS, CH};
        seq[i++] = RingGemm{bb + (size_t)Br.mix_w * 4, 0, 2, 0, 2, BALF_MIX_RING ? 1 : 0};   // 64x64 token-mix matrix: 2 units
        seq[i++] = RingGemm{bb + (size_t)Br.d2_w * 4, 0, KS, 0, KS, CH};
        if (MODE == 1) {
            seq[i++] = RingGemm{bb + (size_t)S.q2_w * 4, 0, 2 * KS, KS, KS, CH};
            seq[i++] = RingGemm{bb + (size_t)S.q2_w * 4, 0, 2 * KS, 0, KS, CH};
            seq[i++] = RingGemm{bb + (size_t)S.r1_w * 4, 0, KS, 0, KS, CH};
            seq[i++] = RingGemm{bb + (size_t)S.r2_w * 4, 0, KS, 0, KS, CH};
        }
    }
    seq[NG] = seq[0]; seq[NG + 1] = seq[0]; seq[NG + 2] = seq[0];    // padding (tot = 0, never issued)
    int gu = 0;                                        // global ring unit counter (wave-uniform)

    // ---- kernel prologue: put every long-latency request in flight before the first wait ----
    //  (1) this lane's stage input (NCHW planes at stage 1, fragment rows staged into the slot otherwise),
    //  (2) the first three weight units of the ring, (3) the per-channel parameters for the LDS cache; only
    //  then the barrier that publishes the cache.  (Touching the u' rows here to warm L2 for the block
    //  kernel's later read was tried: FETCH_SIZE doubled for that tensor and the kernel got no faster.)
    float in[P][3];
    if constexpr (CIN == 3) {
        if (MODE == 0 && P == 4 && A.u8_ch == 0) {
            // grid kernel, float input: the lane's four pixels are adjacent in x (16-byte aligned: W and the
            // group offset are multiples of 4) -> one 16-byte load per colour plane instead of four scalar ones
            const long hw = (long)H * W;
            const long o = (long)py[0] * W + px_[0];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const f4 v = ldg4(A.X + ((long)n * 3 + k) * hw + o);
#pragma unroll
                for (int p = 0; p < P; ++p) in[p][k] = v[p];
            }
        } else {
#pragma unroll
            for (int p = 0; p < P; ++p) load_input3(A, blob + kLayout.u8_lut, n, py[p], px_[p], in[p]);
        }
    }
    if constexpr (use_ring<C>() && CIN != 3) {
        // stage input as the first Linear's B operand, staged through the wave's slot: an ordinary global
        // load issued inside the ring loop would have to be waited for with a vmcnt that drains the LDS-DMA queue
        HL xin[CIN / 32][P];
#pragma unroll
        for (int kk = 0; kk < CIN / 32; ++kk)
#pragma unroll
            for (int p = 0; p < P; ++p) xin[kk][p] = load_frag_px(A.X, pix[p], CIN, kk, q);
#pragma unroll
        for (int kk = 0; kk < CIN / 32; ++kk)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                slot[((kk * P + p) * 2 + 0) * 64 + lane] = xin[kk][p].hi;
                slot[((kk * P + p) * 2 + 1) * 64 + lane] = xin[kk][p].lo;
            }
    }
    if constexpr (use_ring<C>()) {
        const RingChain c0 = make_chain<NT>(seq[0], seq[1], seq[2], seq[3], NG);
#pragma unroll
        for (int t = 0; t < kRingSlots - 1; ++t) chain_issue(c0, ring, t, t, wave, lane);
    }
    {   // parameter cache (see par_floats)
        auto put = [&](int dst, int src, int n) {
            for (int i = threadIdx.x; i < n; i += 256) par[dst + i] = blob[src + i];
        };
        put(kParConv0B * C, S.conv0_b, C);
        put(kParQ1B * C, S.q1_b + MODE * C, C);
        put(kParD1B * C, Br.d1_b, 2 * C);
        put(kParGlnG * C, Br.gln_g, C);
        put(kParGlnB * C, Br.gln_b, C);
        put(kParD2B * C, Br.d2_b, C);
        put(kParMixB * C, Br.mix_b, 64);
        if (MODE == 1) {
            put(kParQ2B * C + 64, S.q2_b, C);
            put(kParR1B * C + 64, S.r1_b, C);
            put(kParR2B * C + 64, S.r2_b, C);
        }
        __syncthreads();
    }
    STAMP(1);   // prologue: inputs, ring prime, parameter cache

    WPre wpre;                                         // stage 1: weights of the next Linear
    if constexpr (!use_ring<C>() && NT == 2) wpre = wpre_load(seq[1], lane);
    // every Linear goes through G: chained LDS-ring version for C >= 64, per-wave streaming otherwise.
    // The Linears of this kernel in execution order (the ring prefetches across them):
    auto G = [&](auto idx, auto &acc, auto bload, auto vm) {
        constexpr int I = decltype(idx)::value;
        const RingGemm &d = seq[I];
        if constexpr (use_ring<C>()) {
            const RingChain c = make_chain<NT>(seq[I], seq[I + 1], seq[I + 2], seq[I + 3], NG - I);
            gemm16_chain<NT, P, decltype(vm)>(acc, c, gu, lane, wave, ring, bload);   // ring primed in the prologue
        } else if constexpr (NT == 2) {
            gemm16_single<P>(acc, wpre, bload);
            constexpr int NX = (I + 1 == 4) ? I + 2 : I + 1;          // entry 4 is the token-mix matrix
            if constexpr (NX < NG) {
                wpre = wpre_load(seq[NX], lane);
                __builtin_amdgcn_sched_barrier(0);                     // keep the loads ahead of the epilogue
            }
        } else {
            gemm16<NT, P>(acc, reinterpret_cast<const float *>(d.wbase), d.wnt0, d.KStot, d.ks0, d.ksn, lane, bload);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
    using I8 = std::integral_constant<int, 8>; using I9 = std::integral_constant<int, 9>;

    // ---- x0 = relu(conv0(X)) ----
    f4 x0[NT][P];
    auto stage1_x0 = [&](f4 (&dst)[NT][P]) {          // 3-input Linear + ReLU on the VALU (stage 1 only)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f4 bias = ldg4(par + kParConv0B * C + 16 * nt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *wr = blob + S.conv0_w + (16 * nt + 4 * q + r) * 3;
                const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    dst[nt][p][r] = fmaxf(bias[r] + in[p][0] * w0 + in[p][1] * w1 + in[p][2] * w2, 0.0f);
            }
        }
    };
    if constexpr (CIN == 3) {
        stage1_x0(x0);
    } else {
        init_bias(x0, par + kParConv0B * C, q);
        if constexpr (use_ring<C>()) {
            // (the B operand was staged into the wave's slot in the prologue)
            G(I0{}, x0, [&](int kk, int p) {
                HL o;
                o.hi = slot[((kk * P + p) * 2 + 0) * 64 + lane];
                o.lo = slot[((kk * P + p) * 2 + 1) * 64 + lane];
                return o;
            }, VmNone{});
        } else {
            G(I0{}, x0, [&](int kk, int p) { return load_frag_px(A.X, pix[p], CIN, kk, q); }, VmNone{});
        }
        relu(x0);
    }

    {
        f4 h[NT][P];
        layernorm_plain<PK>(x0, h);
        store_slot16(slot, h, lane);
    }
    STAMP(2);   // x0 (VALU or GEMM) + LN + slot
    auto from_slot = [&](int kk, int p) {
        HL o;
        o.hi = slot[((kk * P + p) * 2 + 0) * 64 + lane];
        o.lo = slot[((kk * P + p) * 2 + 1) * 64 + lane];
        return o;
    };
    f4 z[NT][P];
    init_bias(z, par + kParQ1B * C, q);
    G(I1{}, z, from_slot, VmNone{});
    gelu<PK>(z);
    STAMP(3);   // dense1 half + GELU

    {
        f4 h[NT][P];
        layernorm_plain<PK>(z, h);
        store_slot16(slot, h, lane);
    }
    STAMP(4);   // LN + slot
    f4 ga[NT][P];
    init_bias(ga, par + kParD1B * C, q);
    G(I2{}, ga, from_slot, VmNone{});
    gelu<PK>(ga);
    STAMP(5);   // branch dense1 (a half) + GELU
    {
        f4 gb[NT][P];
        init_bias(gb, par + kParD1B * C + C, q);
        G(I3{}, gb, from_slot, VmNone{});
        gelu<PK>(gb);
        layernorm<PK>(gb, gb, par + kParGlnG * C, par + kParGlnB * C, q);
        STAMP(6);   // branch dense1 (b half) + GELU + LN
        lds_barrier();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                h2 h01, l01, h23, l23;
                split_pair(gb[nt][p][0], gb[nt][p][1], h01, l01);
                split_pair(gb[nt][p][2], gb[nt][p][3], h23, l23);
                _Float16 *row = bT + (p * C + 16 * nt + 4 * q) * kBtPitch16 + tok;
                _Float16 *rowl = row + P * C * kBtPitch16;
                row[0] = h01[0]; row[kBtPitch16] = h01[1]; row[2 * kBtPitch16] = h23[0]; row[3 * kBtPitch16] = h23[1];
                rowl[0] = l01[0]; rowl[kBtPitch16] = l01[1]; rowl[2 * kBtPitch16] = l23[0]; rowl[3 * kBtPitch16] = l23[1];
            }
    }
    lds_barrier();
    STAMP(7);   // transposed tile written + barrier
    {
        // mix^T[c][g'] = sum_g bT[c][g] * Wmix[g'][g]: A = bT rows (channels), B = natural-order Wmix fragments
        HL w0, w1;
        if constexpr (use_ring<C>() && BALF_MIX_RING) {
            // the mixing matrix arrives through the weight ring as chain entry 4 (two units of four 16-token
            // row tiles); this wave needs row tile `wave` of each
            const RingChain c = make_chain<NT>(seq[4], seq[5], seq[6], seq[7], NG - 4);
            const int later = c.tot[1] + c.tot[2] + c.tot[3];
            auto rd = [&](HL &w, int g) {
                const unsigned char *sl = ring + (g & (kRingSlots - 1)) * kRingSlotBytes + wave * 2048 + lane * 16;
                w.hi = *reinterpret_cast<const h8 *>(sl);
                w.lo = *reinterpret_cast<const h8 *>(sl + 1024);
            };
            ring_wait_barrier(1 + later);
            chain_issue(c, ring, kRingSlots - 1, (gu + kRingSlots - 1) & (kRingSlots - 1), wave, lane);
            rd(w0, gu);
            ring_wait_barrier(later);
            chain_issue(c, ring, kRingSlots, (gu + kRingSlots) & (kRingSlots - 1), wave, lane);
            rd(w1, gu + 1);
            gu += 2;
        } else {
            const char *wm = reinterpret_cast<const char *>(blob + Br.mix_w) + (wave * 2) * 2048 + lane * 16;
            w0.hi = *reinterpret_cast<const h8 *>(wm);        w0.lo = *reinterpret_cast<const h8 *>(wm + 1024);
            w1.hi = *reinterpret_cast<const h8 *>(wm + 2048); w1.lo = *reinterpret_cast<const h8 *>(wm + 3072);
        }
        const float mb1 = par[kParMixB * C + tok] + 1.0f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const _Float16 *row = bT + (p * C + 16 * ct + li) * kBtPitch16 + 8 * q;
                const _Float16 *rowl = row + P * C * kBtPitch16;
                HL a0, a1;
                a0.hi = *reinterpret_cast<const h8 *>(row);      a0.lo = *reinterpret_cast<const h8 *>(rowl);
                a1.hi = *reinterpret_cast<const h8 *>(row + 32); a1.lo = *reinterpret_cast<const h8 *>(rowl + 32);
                f4 m = {0.0f, 0.0f, 0.0f, 0.0f};
                m = mfma16x3(a0, w0, m);
                m = mfma16x3(a1, w1, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[ct][p][r] *= (m[r] + mb1);
            }
    }
    STAMP(8);   // token mix (+ mix weights from the ring)
    HL ub32[(!use_ring<C>() && MODE == 1) ? KS : 1][P];   // stage 1 block kernel: u' rows requested now, used by the
    if constexpr (!use_ring<C>() && MODE == 1) {           // RSHMAG dense2 three Linears later (HBM latency hidden)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int p = 0; p < P; ++p) ub32[kk][p] = load_frag_px(A.U, pix[p], C, kk, q);
    }
    lds_barrier();
    store_slot16(slot, ga, lane);
    f4 o[NT][P];
    init_bias(o, par + kParD2B * C, q);
    G(I5{}, o, from_slot, VmNone{});
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

    STAMP(9);   // gate -> slot, dense2, residual
    if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (!BALF_ABLATE_STORE || o[2 * ks][p][0] == 1.2345e-33f) store_frag_px(A.U, pix[p], C, ks, q, split8(o[2 * ks][p], o[2 * ks + 1][p]));
        STAMP(10);  // u' store
        return;
    } else {
        store_slot16(slot, o, lane);
        f4 x1[NT][P];
        init_bias(x1, par + kParQ2B * C + 64, q);
        if constexpr (use_ring<C>()) {
            HL ub[KS][P];                              // u' fragments: in flight during the v' half of dense2
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) ub[kk][p] = load_frag_px(A.U, pix[p], C, kk, q);
            G(I6{}, x1, from_slot, VmTag<EU, 0>{});            // the u' loads may stay in flight
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    slot[((kk * P + p) * 2 + 0) * 64 + lane] = ub[kk][p].hi;
                    slot[((kk * P + p) * 2 + 1) * 64 + lane] = ub[kk][p].lo;
                }
            G(I7{}, x1, from_slot, VmTag<EU, (UG < 3 ? UG : 3)>{});
        } else {
            G(I6{}, x1, from_slot, VmNone{});
            G(I7{}, x1, [&](int kk, int p) { return ub32[kk][p]; }, VmNone{});
        }
        STAMP(10);  // dense2 of the RSHMAG over cat[u', v'] (u' from HBM)
        if constexpr (CIN == 3) stage1_x0(x0);         // recomputed (3 MACs/channel): frees 32 registers across
                                                       // the whole block branch (bit-identical to the first time)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x1[nt][p] += x0[nt][p];
                if (!BALF_ABLATE_STORE || x1[nt][p][0] == 1.2345e-33f) *reinterpret_cast<f4 *>(A.R + pix[p] * C + 16 * nt + 4 * q) = x1[nt][p] + x0[nt][p];
            }
        layernorm_plain<PK>(x1, x1);
        store_slot16(slot, x1, lane);
        STAMP(11);  // x0 recompute, R store, LN, slot
        f4 m1[NT][P];
        init_bias(m1, par + kParR1B * C + 64, q);
        G(I8{}, m1, from_slot, VmTag<ER, 0>{});             // the R stores may stay in flight
        lrelu(m1);
        store_slot16(slot, m1, lane);
        STAMP(12);  // conv1 + lrelu + slot
        f4 t[NT][P];
        init_bias(t, par + kParR2B * C + 64, q);
        G(I9{}, t, from_slot, VmTag<ER, (UG < 3 ? UG : 3)>{});
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (!BALF_ABLATE_STORE || t[nt][p][0] == 1.2345e-33f) *reinterpret_cast<f4 *>(A.T + pix[p] * C + 16 * nt + 4 * q) = t[nt][p];
                s += t[nt][p];
            }
            // channel sums of this wave's 16 * P pixels (fixed order): one partial row per WAVE, straight to HBM -- the
            // cross-wave sum used to cost a workgroup barrier at the very end of the kernel (6 % of the C = 64 kernel);
            // se_reduce_kernel adds the four rows with the rest
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(s[r]))));
            if (li == 0) *reinterpret_cast<f4 *>(A.partial + ((long)item * 4 + wave) * C + 16 * nt + 4 * q) = s;
        }
        STAMP(13);  // conv2, T store, channel sums
    }
}

// ------------------------------------------------------------------------------------------------
// N-split variant of the branch kernels for C = 256 (BALF_NS256).  The register state of 16 pixels x 256 channels
// x several live tensors pins the kernels above to ONE wave per SIMD at C = 256.  Here a workgroup is 8 waves: each
// 16-pixel tile is shared by a PAIR of waves and each wave of the pair holds half of the output channels of every
// Linear (8 of the 16 row tiles), so a wave needs half the registers and two waves fit on a SIMD.  The two halves
// meet in LDS: the B operand slot of a pixel tile is written half by each wave (a wave's 8 tiles are 4 complete
// K-steps of the next Linear), LayerNorm statistics are exchanged through a small LDS array, and a weight-ring unit
// carries the 4 row tiles of both halves (16 KiB).  Everything else -- token mix, gating, residuals, SE sums,
// stores -- is local to a wave's channels.
// ------------------------------------------------------------------------------------------------
#ifndef BALF_NS256
#define BALF_NS256 1
#endif
#ifndef BALF_NS128
#define BALF_NS128 1
#endif
#ifndef BALF_NS64
#define BALF_NS64 0      // measured slower at C = 64 (VALU-bound kernels; half the MFMAs per barrier)
#endif
// Ring geometry: a unit carries HT row tiles per half.  C = 256: HT = 4 (8 tiles, 16 KiB; every wave DMAs one tile, hi
// and lo).  C = 128: HT = 2 (4 tiles, 8 KiB; every wave DMAs half a tile), which keeps the workgroup under 80 KiB of
// LDS so that two of them (16 waves) share a CU.
constexpr int kNsRingSlots = 4;
template <int HT> constexpr int ns_slot_bytes() { return 2 * HT * 2048; }
template <int HT> constexpr int ns_ring_bytes() { return kNsRingSlots * ns_slot_bytes<HT>(); }
template <int C> constexpr int ns_ht() { return C >= 256 ? 4 : 2; }

template <int HT>
__device__ __forceinline__ void ring_wait_barrier_ns(int after /* units issued after the awaited one */) {
    ring_wait<(HT == 4 ? 2 : 1), 0>(after, false);       // DMA instructions per unit and wave: 2 (HT = 4) or 1 (HT = 2)
}

struct RingGemmNs {
    const char *wbase;
    int wnt0, KStot, ks0, ksn;
    int chunks;            // groups of 4 row tiles PER HALF (units = chunks * ksn)
    int half_tiles;        // row-tile distance between the two halves (0: both halves get the same tiles)
};
struct RingChainNs {
    RingGemmNs g[4];
    int tot[4];
};

__device__ __forceinline__ RingChainNs make_chain_ns(const RingGemmNs &g0, const RingGemmNs &g1, const RingGemmNs &g2,
                                                     const RingGemmNs &g3, int n) {
    RingChainNs c;
    c.g[0] = g0; c.g[1] = g1; c.g[2] = g2; c.g[3] = g3;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.tot[i] = (i < n) ? c.g[i].chunks * c.g[i].ksn : 0;
    return c;
}

// HT = 4: wave8 fetches row tile (wave8 & 3) of half (wave8 >> 2) of chain-relative unit t, both parts;
// HT = 2: tile slot j = wave8 >> 1 (half j >> 1, tile j & 1), part wave8 & 1 (hi or lo).
template <int HT>
__device__ __forceinline__ void chain_issue_ns(const RingChainNs &c, unsigned char *ring, int t, int slot, int wave8,
                                               int lane) {
    RingGemmNs d = c.g[0];
    bool valid = c.tot[0] > 0, walking = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (walking) {
            if (valid && t >= c.tot[i]) {
                t -= c.tot[i];
                if (i < 3) { d = c.g[i + 1]; valid = c.tot[i + 1] > 0; }
                else valid = false;
            } else {
                walking = false;
            }
        }
    if (!valid) return;
    const int cc = t / d.ksn, k = t - cc * d.ksn;
    const int j = HT == 4 ? wave8 : (wave8 >> 1);                       // tile slot of the unit
    int tile;
    if (d.half_tiles == 0) tile = d.wnt0 + (j & 3);                     // token-mix matrix: its 4 row tiles (HT = 4: twice)
    else tile = d.wnt0 + (j / HT) * d.half_tiles + cc * HT + (j % HT);
    const char *src = d.wbase + ((size_t)tile * d.KStot + d.ks0 + k) * 2048 + lane * 16;
    unsigned char *dst = ring + slot * ns_slot_bytes<HT>() + j * 2048;
    if constexpr (HT == 4) {
        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + 1024), (lds_void *)(dst + 1024), 16, 0, 0);
    } else {
        const int part = (wave8 & 1) * 1024;
        __builtin_amdgcn_global_load_lds((glb_void *)(src + part), (lds_void *)(dst + part), 16, 0, 0);
    }
}

template <int NTL, int HT, int CI, int P, typename VM, typename BL>
__device__ __forceinline__ void chain_chunk_ns(f4 (&acc)[NTL][P], const RingChainNs &c, unsigned char *ring, int gu,
                                               int lane, int wave8, int hh, BL bload) {
    if constexpr (CI * HT < NTL) {
        const int ksn = c.g[0].ksn;
        const int later = c.tot[1] + c.tot[2] + c.tot[3];
        for (int k = 0; k < ksn; ++k) {
            const int u = CI * ksn + k;
            ring_wait<(HT == 4 ? 2 : 1), VM::e>(c.tot[0] - 1 - u + later, u + VM::used < 3);
            chain_issue_ns<HT>(c, ring, u + kNsRingSlots - 1, (gu + u + kNsRingSlots - 1) & (kNsRingSlots - 1), wave8, lane);
            HL a[HT], b[P];
            const unsigned char *sl =
                ring + ((gu + u) & (kNsRingSlots - 1)) * ns_slot_bytes<HT>() + hh * (HT * 2048) + lane * 16;
#pragma unroll
            for (int nt = 0; nt < HT; ++nt) {
                a[nt].hi = *reinterpret_cast<const h8 *>(sl + nt * 2048);
                a[nt].lo = *reinterpret_cast<const h8 *>(sl + nt * 2048 + 1024);
            }
#pragma unroll
            for (int p = 0; p < P; ++p) b[p] = bload(k, p);
#pragma unroll
            for (int nt = 0; nt < HT; ++nt)
#pragma unroll
                for (int p = 0; p < P; ++p) if (!BALF_DROP_WLO) acc[CI * HT + nt][p] = mfma16(a[nt].lo, b[p].hi, acc[CI * HT + nt][p]);
#pragma unroll
            for (int nt = 0; nt < HT; ++nt)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[CI * HT + nt][p] = mfma16(a[nt].hi, b[p].lo, acc[CI * HT + nt][p]);
#pragma unroll
            for (int nt = 0; nt < HT; ++nt)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[CI * HT + nt][p] = mfma16(a[nt].hi, b[p].hi, acc[CI * HT + nt][p]);
        }
        chain_chunk_ns<NTL, HT, CI + 1, P, VM>(acc, c, ring, gu, lane, wave8, hh, bload);
    }
}

template <int NTL, int HT, int P, typename VM = VmNone, typename BL>
__device__ __forceinline__ void gemm16_chain_ns(f4 (&acc)[NTL][P], const RingChainNs &c, int &gu, int lane, int wave8,
                                                int hh, unsigned char *ring, BL bload) {
    chain_chunk_ns<NTL, HT, 0, P, VM>(acc, c, ring, gu, lane, wave8, hh, bload);
    gu += c.tot[0];
}

// LayerNorm statistics over all C channels of a pixel: this wave's half + the partner's, through LDS
// (lnx: [which 0/1][pair 4][half 2][P][16 pixels]); all 8 waves call this together (two barriers).
template <int NTL, int P>
__device__ __forceinline__ void ln_stats_ns(const f4 (&x)[NTL][P], float *lnx, int pg, int hh, int q, int li,
                                            float (&mean)[P], float (&rstd)[P]) {
    constexpr float inv_c = 1.0f / (32 * NTL);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) s += (x[nt][p][0] + x[nt][p][1]) + (x[nt][p][2] + x[nt][p][3]);
        s = quarter_allreduce(s);
        if (q == 0) lnx[((pg * 2 + hh) * P + p) * 16 + li] = s;
    }
    lds_barrier();
#pragma unroll
    for (int p = 0; p < P; ++p) {
        mean[p] = (lnx[((pg * 2) * P + p) * 16 + li] + lnx[((pg * 2 + 1) * P + p) * 16 + li]) * inv_c;
        float v = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = x[nt][p][r] - mean[p];
                v = fmaf(d, d, v);
            }
        v = quarter_allreduce(v);
        if (q == 0) lnx[128 * P + ((pg * 2 + hh) * P + p) * 16 + li] = v;
    }
    lds_barrier();
#pragma unroll
    for (int p = 0; p < P; ++p)
        rstd[p] = __builtin_amdgcn_rsqf((lnx[128 * P + ((pg * 2) * P + p) * 16 + li] +
                                         lnx[128 * P + ((pg * 2 + 1) * P + p) * 16 + li]) * inv_c + kLnEps);
}

template <int C>
constexpr int ns_lds_bytes() {
    constexpr int P = StageP<C>::P;
    constexpr int slots = 4 * (C / 32) * P * 2048;             // one B-operand slot per pixel tile (shared by a pair)
    constexpr int bt = 2 * P * C * kBtPitch16 * 2;
    return (slots > bt ? slots : bt) + 4 * C * 4 + ns_ring_bytes<ns_ht<C>()>() + par_floats<C>() * 4 + 256 * P * 4;
}

template <int C, int CIN, int MODE>
__global__ __launch_bounds__(512, (C >= 256 ? 2 : 4)) void stage_branch_kernel16_ns(StageArgs A) {   // waves per SIMD
    constexpr int P = StageP<C>::P, NT = C / 16, NTL = NT / 2, KS = C / 32, KSL = KS / 2, HT = ns_ht<C>();
    constexpr int kNsRingBytes = ns_ring_bytes<HT>(), kNsRingSlotBytes = ns_slot_bytes<HT>();
    static_assert(KSL >= 1 && NTL % HT == 0, "N-split geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int slots_b = 4 * KS * P * 2048, bt_b = 2 * P * C * kBtPitch16 * 2;
    constexpr int main_bytes = slots_b > bt_b ? slots_b : bt_b;
    _Float16 *bT = reinterpret_cast<_Float16 *>(smem_raw);                 // [hi|lo][p][c][pitch]
    // (smem_raw + main_bytes: 4 * C floats, formerly the cross-wave channel-sum scratch; kept so the LDS image is unchanged)
    unsigned char *ring = smem_raw + main_bytes + 4 * C * 4;
    float *par = reinterpret_cast<float *>(smem_raw + main_bytes + 4 * C * 4 + kNsRingBytes);
    float *lnx = par + par_floats<C>();

    const int lane = threadIdx.x & 63, wave8 = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    const int pg = wave8 >> 1, hh = wave8 & 1;         // pixel tile of the token group, channel half
    const int nt0 = hh * NTL, ks0 = hh * KSL;
    h8 *slot = reinterpret_cast<h8 *>(smem_raw) + pg * (KS * P * 2 * 64);  // [ks][p][hi|lo][lane], shared by the pair
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    const int H = A.H, W = A.W;
    const int cols = W / 8 / P;
    const int per_img = (H / 8) * cols;
    const int nwg = gridDim.x;
    const int xq = nwg >> 3, xr = nwg & 7, xl = blockIdx.x & 7, xj = blockIdx.x >> 3;
    const int item = (xl < xr ? xl * (xq + 1) : xr * (xq + 1) + (xl - xr) * xq) + xj;
    const int n = item / per_img;
    const int rem = item - n * per_img;
    const int iy0 = rem / cols, ix0 = (rem - iy0 * cols) * P;
    const int tok = 16 * pg + li, ty = tok >> 3, tx = tok & 7;
    long pix[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int y, x;
        if (MODE == 0) { y = ty * (H / 8) + iy0; x = tx * (W / 8) + ix0 + p; }
        else           { y = 8 * iy0 + ty;       x = 8 * (ix0 + p) + tx; }
        pix[p] = ((long)n * H + y) * W + x;
    }

    constexpr int NG = (MODE == 0) ? 6 : 10;
    constexpr int EU = KSL * P * 2, ER = NTL * P, UG = (NTL / HT) * KS;      // ring-wait allowances (VmTag)
    const char *bb = reinterpret_cast<const char *>(blob);
    RingGemmNs seq[NG + 3];
    {
        constexpr int CH = NTL / HT, KI = CIN / 32, HTD = NT / 2;
        int i = 0;
        seq[i++] = RingGemmNs{bb + (size_t)S.conv0_w * 4, 0, KI, 0, KI, CH, HTD};
        seq[i++] = RingGemmNs{bb + (size_t)S.q1_w * 4, MODE * NT, KS, 0, KS, CH, HTD};
        seq[i++] = RingGemmNs{bb + (size_t)Br.d1_w * 4, 0, KS, 0, KS, CH, HTD};
        seq[i++] = RingGemmNs{bb + (size_t)Br.d1_w * 4, NT, KS, 0, KS, CH, HTD};
        seq[i++] = RingGemmNs{bb + (size_t)Br.mix_w * 4, 0, 2, 0, 2, 1, 0};        // 64x64 token-mix matrix: 2 units
        seq[i++] = RingGemmNs{bb + (size_t)Br.d2_w * 4, 0, KS, 0, KS, CH, HTD};
        if (MODE == 1) {
            seq[i++] = RingGemmNs{bb + (size_t)S.q2_w * 4, 0, 2 * KS, KS, KS, CH, HTD};
            seq[i++] = RingGemmNs{bb + (size_t)S.q2_w * 4, 0, 2 * KS, 0, KS, CH, HTD};
            seq[i++] = RingGemmNs{bb + (size_t)S.r1_w * 4, 0, KS, 0, KS, CH, HTD};
            seq[i++] = RingGemmNs{bb + (size_t)S.r2_w * 4, 0, KS, 0, KS, CH, HTD};
        }
    }
    seq[NG] = seq[0]; seq[NG + 1] = seq[0]; seq[NG + 2] = seq[0];
    int gu = 0;

    // ---- prologue: stage input (K-steps split between the pair) -> shared slot, ring prime, parameter cache ----
    {
        constexpr int KI = CIN / 32;
        constexpr int KIH = KI >= 2 ? KI / 2 : 1;
        const int kbase = KI >= 2 ? hh * KIH : 0;
        if (KI >= 2 || hh == 0) {
            HL xin[KIH][P];
#pragma unroll
            for (int kk = 0; kk < KIH; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) xin[kk][p] = load_frag_px(A.X, pix[p], CIN, kbase + kk, q);
#pragma unroll
            for (int kk = 0; kk < KIH; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    slot[(((kbase + kk) * P + p) * 2 + 0) * 64 + lane] = xin[kk][p].hi;
                    slot[(((kbase + kk) * P + p) * 2 + 1) * 64 + lane] = xin[kk][p].lo;
                }
        }
    }
    {
        const RingChainNs c0 = make_chain_ns(seq[0], seq[1], seq[2], seq[3], NG);
#pragma unroll
        for (int t = 0; t < kNsRingSlots - 1; ++t) chain_issue_ns<HT>(c0, ring, t, t, wave8, lane);
    }
    {
        auto put = [&](int dst, int src, int n_) {
            for (int i = threadIdx.x; i < n_; i += 512) par[dst + i] = blob[src + i];
        };
        put(kParConv0B * C, S.conv0_b, C);
        put(kParQ1B * C, S.q1_b + MODE * C, C);
        put(kParD1B * C, Br.d1_b, 2 * C);
        put(kParGlnG * C, Br.gln_g, C);
        put(kParGlnB * C, Br.gln_b, C);
        put(kParD2B * C, Br.d2_b, C);
        put(kParMixB * C, Br.mix_b, 64);
        if (MODE == 1) {
            put(kParQ2B * C + 64, S.q2_b, C);
            put(kParR1B * C + 64, S.r1_b, C);
            put(kParR2B * C + 64, S.r2_b, C);
        }
        __syncthreads();
    }

    auto G = [&](auto idx, auto &acc, auto bload, auto vm) {
        constexpr int I = decltype(idx)::value;
        const RingChainNs c = make_chain_ns(seq[I], seq[I + 1], seq[I + 2], seq[I + 3], NG - I);
        gemm16_chain_ns<NTL, HT, P, decltype(vm)>(acc, c, gu, lane, wave8, hh, ring, bload);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
    using I8 = std::integral_constant<int, 8>; using I9 = std::integral_constant<int, 9>;
    auto from_slot = [&](int kk, int p) {
        HL o;
        o.hi = slot[((kk * P + p) * 2 + 0) * 64 + lane];
        o.lo = slot[((kk * P + p) * 2 + 1) * 64 + lane];
        return o;
    };
    // this wave's tiles are K-steps ks0 .. ks0+KSL-1 of the next Linear; the barrier in front keeps the partner's (and
    // this wave's) reads of the previous contents ahead of the overwrite, the ring's first barrier publishes it
    auto to_slot = [&](const f4 (&t)[NTL][P]) {
        lds_barrier();
        store_slot16(slot + ks0 * (P * 2 * 64), t, lane);
    };
    auto ln_plain = [&](const f4 (&xin_)[NTL][P], f4 (&yout)[NTL][P]) {
        float mean[P], rstd[P];
        ln_stats_ns<NTL, P>(xin_, lnx, pg, hh, q, li, mean, rstd);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float shift = -mean[p] * rstd[p];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) yout[nt][p][r] = fmaf(xin_[nt][p][r], rstd[p], shift);
        }
    };

    // ---- x0 = relu(conv0(X)) ----
    f4 x0[NTL][P];
    init_bias(x0, par + kParConv0B * C + 16 * nt0, q);
    G(I0{}, x0, from_slot, VmNone{});
    relu(x0);
    {
        f4 h[NTL][P];
        ln_plain(x0, h);
        to_slot(h);
    }
    f4 z[NTL][P];
    init_bias(z, par + kParQ1B * C + 16 * nt0, q);
    G(I1{}, z, from_slot, VmNone{});
    gelu<false, (C == 128 && MODE == 1)>(z);
    {
        f4 h[NTL][P];
        ln_plain(z, h);
        to_slot(h);
    }
    f4 ga[NTL][P];
    init_bias(ga, par + kParD1B * C + 16 * nt0, q);
    G(I2{}, ga, from_slot, VmNone{});
    gelu<false, (C == 128 && MODE == 1)>(ga);
    {
        f4 gb[NTL][P];
        init_bias(gb, par + kParD1B * C + C + 16 * nt0, q);
        G(I3{}, gb, from_slot, VmNone{});
        gelu<false, (C == 128 && MODE == 1)>(gb);
        {
            float mean[P], rstd[P];
            ln_stats_ns<NTL, P>(gb, lnx, pg, hh, q, li, mean, rstd);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const f4 gg = ldg4(par + kParGlnG * C + 16 * (nt0 + nt) + 4 * q), be = ldg4(par + kParGlnB * C + 16 * (nt0 + nt) + 4 * q);
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gb[nt][p][r] = (gb[nt][p][r] - mean[p]) * rstd[p] * gg[r] + be[r];
            }
        }
        lds_barrier();                                  // slots (aliased by bT) are no longer read
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                h2 h01, l01, h23, l23;
                split_pair(gb[nt][p][0], gb[nt][p][1], h01, l01);
                split_pair(gb[nt][p][2], gb[nt][p][3], h23, l23);
                _Float16 *row = bT + (p * C + 16 * (nt0 + nt) + 4 * q) * kBtPitch16 + tok;
                _Float16 *rowl = row + P * C * kBtPitch16;
                row[0] = h01[0]; row[kBtPitch16] = h01[1]; row[2 * kBtPitch16] = h23[0]; row[3 * kBtPitch16] = h23[1];
                rowl[0] = l01[0]; rowl[kBtPitch16] = l01[1]; rowl[2 * kBtPitch16] = l23[0]; rowl[3 * kBtPitch16] = l23[1];
            }
    }
    lds_barrier();
    {
        // the mixing matrix arrives through the ring as chain entry 4 (two units); this wave needs row tile pg of each
        HL w0, w1;
        const RingChainNs c = make_chain_ns(seq[4], seq[5], seq[6], seq[7], NG - 4);
        const int later = c.tot[1] + c.tot[2] + c.tot[3];
        auto rd = [&](HL &w, int g) {
            const unsigned char *sl = ring + (g & (kNsRingSlots - 1)) * kNsRingSlotBytes + pg * 2048 + lane * 16;
            w.hi = *reinterpret_cast<const h8 *>(sl);
            w.lo = *reinterpret_cast<const h8 *>(sl + 1024);
        };
        ring_wait_barrier_ns<HT>(1 + later);
        chain_issue_ns<HT>(c, ring, kNsRingSlots - 1, (gu + kNsRingSlots - 1) & (kNsRingSlots - 1), wave8, lane);
        rd(w0, gu);
        ring_wait_barrier_ns<HT>(later);
        chain_issue_ns<HT>(c, ring, kNsRingSlots, (gu + kNsRingSlots) & (kNsRingSlots - 1), wave8, lane);
        rd(w1, gu + 1);
        gu += 2;
        const float mb1 = par[kParMixB * C + tok] + 1.0f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ct = 0; ct < NTL; ++ct) {
                const _Float16 *row = bT + (p * C + 16 * (nt0 + ct) + li) * kBtPitch16 + 8 * q;
                const _Float16 *rowl = row + P * C * kBtPitch16;
                HL a0, a1;
                a0.hi = *reinterpret_cast<const h8 *>(row);      a0.lo = *reinterpret_cast<const h8 *>(rowl);
                a1.hi = *reinterpret_cast<const h8 *>(row + 32); a1.lo = *reinterpret_cast<const h8 *>(rowl + 32);
                f4 m = {0.0f, 0.0f, 0.0f, 0.0f};
                m = mfma16x3(a0, w0, m);
                m = mfma16x3(a1, w1, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[ct][p][r] *= (m[r] + mb1);
            }
    }
    to_slot(ga);                                        // (its barrier also ends the reads of bT)
    f4 o[NTL][P];
    init_bias(o, par + kParD2B * C + 16 * nt0, q);
    G(I5{}, o, from_slot, VmNone{});
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

    if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks)
                store_frag_px(A.U, pix[p], C, ks0 + ks, q, split8(o[2 * ks][p], o[2 * ks + 1][p]));
        return;
    } else {
        HL ub[KSL][P];                                  // this wave's half of the u' K-steps, in flight during G(I6)
#pragma unroll
        for (int kk = 0; kk < KSL; ++kk)
#pragma unroll
            for (int p = 0; p < P; ++p) ub[kk][p] = load_frag_px(A.U, pix[p], C, ks0 + kk, q);
        to_slot(o);
        f4 x1[NTL][P];
        init_bias(x1, par + kParQ2B * C + 64 + 16 * nt0, q);
        G(I6{}, x1, from_slot, VmTag<EU, 0>{});                // the u' loads may stay in flight
        lds_barrier();
#pragma unroll
        for (int kk = 0; kk < KSL; ++kk)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                slot[(((ks0 + kk) * P + p) * 2 + 0) * 64 + lane] = ub[kk][p].hi;
                slot[(((ks0 + kk) * P + p) * 2 + 1) * 64 + lane] = ub[kk][p].lo;
            }
        G(I7{}, x1, from_slot, VmTag<EU, (UG < 3 ? UG : 3)>{});
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x1[nt][p] += x0[nt][p];
                *reinterpret_cast<f4 *>(A.R + pix[p] * C + 16 * (nt0 + nt) + 4 * q) = x1[nt][p] + x0[nt][p];
            }
        ln_plain(x1, x1);
        to_slot(x1);
        f4 m1[NTL][P];
        init_bias(m1, par + kParR1B * C + 64 + 16 * nt0, q);
        G(I8{}, m1, from_slot, VmTag<ER, 0>{});                // the R stores may stay in flight
        lrelu(m1);
        to_slot(m1);
        f4 t[NTL][P];
        init_bias(t, par + kParR2B * C + 64 + 16 * nt0, q);
        G(I9{}, t, from_slot, VmTag<ER, (UG < 3 ? UG : 3)>{});
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            f4 ssum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                *reinterpret_cast<f4 *>(A.T + pix[p] * C + 16 * (nt0 + nt) + 4 * q) = t[nt][p];
                ssum += t[nt][p];
            }
            // one partial row per wave PAIR (each wave its half of the channels), straight to HBM: no final barrier
#pragma unroll
            for (int r = 0; r < 4; ++r) ssum[r] = row_ror_add<1>(row_ror_add<2>(row_ror_add<4>(row_ror_add<8>(ssum[r]))));
            if (li == 0) *reinterpret_cast<f4 *>(A.partial + ((long)item * 4 + pg) * C + 16 * (nt0 + nt) + 4 * q) = ssum;
        }
    }
}

// x2 = t*s + r, 2x2 max pool, written in fragment format for the next stage's MFMA B operand.
template <int C>
__global__ __launch_bounds__(256) void pool_kernel16(const float *__restrict__ T, const float *__restrict__ R,
                                                     const float *__restrict__ scale, int B, int H, int W,
                                                     float *__restrict__ out) {
    constexpr int G = C / 8;                       // (ks, q) units of 8 channels per pixel
    const long total = (long)B * (H / 2) * (W / 2) * G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % G), ks = g >> 2, q = g & 3;
        long pxy = i / G;
        const long opix = pxy;
        const int xo = (int)(pxy % (W / 2));
        pxy /= (W / 2);
        const int yo = (int)(pxy % (H / 2));
        const int n = (int)(pxy / (H / 2));
        const int c0 = 32 * ks + 4 * q, c1 = c0 + 16;
        const f4 s0 = ldg4(scale + (long)n * C + c0), s1 = ldg4(scale + (long)n * C + c1);
        const long base = (((long)n * H + 2 * yo) * W + 2 * xo) * C;
        f4 m0, m1;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long o = base + ((long)dy * W + dx) * C;
                const f4 v0 = ldg4(T + o + c0) * s0 + ldg4(R + o + c0);
                const f4 v1 = ldg4(T + o + c1) * s1 + ldg4(R + o + c1);
                if (dy == 0 && dx == 0) { m0 = v0; m1 = v1; }
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) { m0[r] = fmaxf(m0[r], v0[r]); m1[r] = fmaxf(m1[r], v1[r]); }
            }
        store_frag_px(out, opix, C, ks, q, split8(m0, m1));
    }
}

__global__ __launch_bounds__(256, 2) void head_kernel16(HeadArgs A) {
    constexpr int C = 256, NT = 16, KS = 8, HT = kHeadNPad / 16;
    __shared__ __attribute__((aligned(16))) h8 smem[4 * KS * 2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    h8 *slot = smem + wave * (KS * 2 * 64);
    const float *blob = A.blob;
    const long hw = (long)A.h * A.w;
    const long pixel = ((long)blockIdx.x * 4 + wave) * 16 + li;
    const int n = (int)(pixel / hw);
    const long o = pixel - (long)n * hw;
    const int i = (int)(o / A.w), j = (int)(o - (long)i * A.w);

    f4 f[NT][1];
    init_bias(f, blob + A.off.conv2_b, q);
    gemm16<NT, 1>(f, blob + A.off.conv2_w, 0, KS, 0, KS, lane, [&](int kk, int) {
        const int c0 = 32 * kk + 4 * q, c1 = c0 + 16;
        const f4 v0 = ldg4(A.T + pixel * C + c0) * ldg4(A.scale + (long)n * C + c0) + ldg4(A.R + pixel * C + c0);
        const f4 v1 = ldg4(A.T + pixel * C + c1) * ldg4(A.scale + (long)n * C + c1) + ldg4(A.R + pixel * C + c1);
        return split8(v0, v1);
    });
    relu(f);
    store_slot16(slot, f, lane);
    f4 z[HT][1];
    init_bias(z, blob + A.head_b, q);
    gemm16<HT, 1>(z, blob + A.head_w, 0, KS, 0, KS, lane, [&](int kk, int) {
        HL v;
        v.hi = slot[(kk * 2 + 0) * 64 + lane];
        v.lo = slot[(kk * 2 + 1) * 64 + lane];
        return v;
    });

    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 al = ldg4(blob + A.head_alpha + 16 * t + 4 * q), be = ldg4(blob + A.head_beta + 16 * t + 4 * q);
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
    const float inv = 1.0f / sum;
    const int Wp = 8 * A.w;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 pr = z[t][0] * inv;
        float *dst = A.prob + ((long)n * 8 * A.h + 8 * i + 2 * t + (q >> 1)) * Wp + 8 * j + 4 * (q & 1);
        *reinterpret_cast<f4 *>(dst) = pr;
    }
}

// ------------------------------------------------------------------------------------------------
// Stage-4 tail + detector head, second form (BALF_HEAD_NSPLIT): a workgroup = 64 pixels of the 1/8-resolution map.
// head_kernel16 above gives every wave its own 16 pixels and lets it stream both weight matrices (256 KB + 80 KB of
// split-f16 fragments) through L2 for them: 21 KB of weight traffic per pixel, waves parked or issue-stalled 91 % of the
// time.  Here conv2 (256 -> 256) is split by OUTPUT channels: wave w computes row tiles 4w .. 4w+3 for all four pixel
// tiles, so the workgroup reads conv2 once (256 KB per 64 pixels instead of per 16) and every 8 KB of weights feeds
// 48 MFMAs.  x2 = t*s + r is staged once as shared B fragments in LDS (64 KB), the conv2 output goes back through the
// same buffer (wave w owns K-steps 2w, 2w+1 of it), the 65-way head Linear + BatchNorm + softmax + pixel shuffle
// stay per pixel tile.  Reference: Down.forward tail, mlp_ma_decoder.py:241-244; DetectorHead, decoder.py:16-30.
// ------------------------------------------------------------------------------------------------
#ifndef BALF_HEAD_NSPLIT
#define BALF_HEAD_NSPLIT 1
#endif
__global__ __launch_bounds__(256, 2) void head_kernel16_ns(HeadArgs A) {
    constexpr int C = 256, KS = 8, HT = kHeadNPad / 16;
    __shared__ __attribute__((aligned(16))) h8 xs[4 * KS * 2 * 64];          // [tile][K-step][hi|lo][lane]: 64 KiB
    const int lane = threadIdx.x & 63, q = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *blob = A.blob;
    const long hw = (long)A.h * A.w;
    const long pixel = ((long)blockIdx.x * 4 + wave) * 16 + li;               // this wave's own pixel tile
    const int n = (int)(pixel / hw);
    const long o = pixel - (long)n * hw;
    const int i = (int)(o / A.w), j = (int)(o - (long)i * A.w);

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
                const HL v = split8(v0, v1);
                slot[(kk * 2 + 0) * 64 + lane] = v.hi;
                slot[(kk * 2 + 1) * 64 + lane] = v.lo;
            }
        }
    }
    __syncthreads();

    // ---- conv2: row tiles 4w .. 4w+3 x four pixel tiles; weight fragments one K-step ahead in registers ----
    f4 f[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 b = ldg4(blob + A.off.conv2_b + 16 * (4 * wave + t) + 4 * q);
#pragma unroll
        for (int p = 0; p < 4; ++p) f[t][p] = b;
    }
    {
        const char *wb = reinterpret_cast<const char *>(blob + A.off.conv2_w) + ((size_t)(4 * wave) * KS) * 2048 + lane * 16;
        auto wload = [&](HL (&a)[4], int kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const char *p = wb + ((size_t)t * KS + kk) * 2048;
                a[t].hi = *reinterpret_cast<const h8 *>(p);
                a[t].lo = *reinterpret_cast<const h8 *>(p + 1024);
            }
        };
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
        HL a0[4], a1[4];
        wload(a0, 0);
        for (int kk = 0; kk < KS; kk += 2) {
            wload(a1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(a0, kk);
            __builtin_amdgcn_sched_barrier(0);
            wload(a0, kk + 2 < KS ? kk + 2 : kk);
            __builtin_amdgcn_sched_barrier(0);
            compute(a1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                                  // everyone is done reading x2: the buffer takes the conv2 output
    // relu -> B fragments of the head Linear: this wave's 4 row tiles are K-steps 2w, 2w+1 of every pixel tile
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f4 v0 = f[2 * s2][p], v1 = f[2 * s2 + 1][p];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v0[r] = max0(v0[r]); v1[r] = max0(v1[r]); }
            const HL v = split8(v0, v1);
            xs[(p * KS + 2 * wave + s2) * 2 * 64 + lane] = v.hi;
            xs[(p * KS + 2 * wave + s2) * 2 * 64 + 64 + lane] = v.lo;
        }
    __syncthreads();

    // ---- head Linear 256 -> 65 (padded to 80) on the wave's own pixel tile, BatchNorm(eval), softmax, pixel shuffle ----
    f4 z[HT][1];
    init_bias(z, blob + A.head_b, q);
    {
        const h8 *slot = xs + wave * (KS * 2 * 64);
        gemm16<HT, 1>(z, blob + A.head_w, 0, KS, 0, KS, lane, [&](int kk, int) {
            HL v;
            v.hi = slot[(kk * 2 + 0) * 64 + lane];
            v.lo = slot[(kk * 2 + 1) * 64 + lane];
            return v;
        });
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        const f4 al = ldg4(blob + A.head_alpha + 16 * t + 4 * q), be = ldg4(blob + A.head_beta + 16 * t + 4 * q);
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
    const float inv = 1.0f / sum;
    const int Wp = 8 * A.w;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f4 pr = z[t][0] * inv;
        float *dst = A.prob + ((long)n * 8 * A.h + 8 * i + 2 * t + (q >> 1)) * Wp + 8 * j + 4 * (q & 1);
        *reinterpret_cast<f4 *>(dst) = pr;
    }
}

#ifndef BALF_S1_WAVE
#define BALF_S1_WAVE 1      // stage 1: persistent wave-owns-group kernels (stage1_f16.h); 0 = generic stage kernels
#endif
#include "stage1_f16.h"
#ifndef BALF_CS_MIN_C
#define BALF_CS_MIN_C 64    // stages with C >= this run the channel-split kernels (stage_cs_f16.h); 1024 = ring kernels everywhere
#endif
#include "stage_cs_f16.h"

// stages whose block kernel leaves x1 + hidden-layer sums and whose pool kernel is replaced by a tail kernel
template <int C> constexpr bool stage_fused() {
    return C == 32 ? (BALF_S1_WAVE != 0 && BALF_S1_FUSE != 0) : (C >= BALF_CS_MIN_C && C >= 64 && cs_fused<C>());
}

template <int C, int CIN>
int run_stage16(const float *blob, int s, const float *X, const InputU8 &u8, int B, int H, int W, float *U, float *T, float *R,
                float *partial, float *chunk, float *scale, hipStream_t st) {
    constexpr int P = StageP<C>::P;
    constexpr int lds = stage_lds_bytes16<C, P>();
    StageArgs a{blob, kLayout.st[s], X, u8.p, u8.ch, u8.h, u8.w, u8.top, u8.left, B, H, W, U, T, R, partial};
    int per_img = (H / 8) * (W / 8 / P);
    const int nwg = B * per_img;
    auto k0 = stage_branch_kernel16<C, CIN, 0>;
    auto k1 = stage_branch_kernel16<C, CIN, 1>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
                hipSuccess)
            return BALF_ERR_LAUNCH;
    }
    if constexpr (C == 32 && BALF_S1_WAVE != 0) {
        // persistent: one workgroup per CU (256 CUs on MI355X; any multiple of 8 is correct), waves loop over token groups
        auto g0 = u8.ch ? stage1_kernel16<0, true> : stage1_kernel16<0, false>;
        auto g1 = u8.ch ? stage1_kernel16<1, true> : stage1_kernel16<1, false>;
        constexpr int l0 = s1_lds_bytes<0>(), l1 = s1_lds_bytes<1>();
        static_assert(l0 <= 160 * 1024 && l1 <= 160 * 1024, "stage-1 LDS image");
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(g0), hipFuncAttributeMaxDynamicSharedMemorySize, l0) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(g1), hipFuncAttributeMaxDynamicSharedMemorySize, l1) !=
                hipSuccess)
            return BALF_ERR_LAUNCH;
        per_img = (H / 8) * (W / 8);                       // one partial-sum row per token group
        const long groups = (long)B * per_img;
        auto blocks = [&](int waves) {
            long b = (groups + waves - 1) / waves;
            b = (b + 7) / 8 * 8;
            return (unsigned)(b < 256 ? b : 256);
        };
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(g0, dim3(blocks(s1_waves<0>())), dim3(s1_waves<0>() * 64), l0, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(g1, dim3(blocks(s1_waves<1>())), dim3(s1_waves<1>() * 64), l1, st, a));
    } else if constexpr (C >= BALF_CS_MIN_C && C >= 64) {
        // channel-split kernels: one workgroup of C/32 waves per token group, no weight ring
        auto c0k = stage_cs_kernel16<C, CIN, 0>;
        auto c1k = stage_cs_kernel16<C, CIN, 1>;
        constexpr int G = cs_groups<C>();
        constexpr int clds0 = cs_lds_bytes<C>() * G + cs_lut_bytes<0>(), clds1 = cs_lds_bytes<C>() * G + cs_lut_bytes<1>();
        constexpr int clds = clds0 > clds1 ? clds0 : clds1;
        static_assert(clds <= 160 * 1024, "channel-split LDS image");
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(c0k), hipFuncAttributeMaxDynamicSharedMemorySize, clds) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(c1k), hipFuncAttributeMaxDynamicSharedMemorySize, clds) !=
                hipSuccess)
            return BALF_ERR_LAUNCH;
        per_img = (H / 8) * (W / 8);                       // one partial-sum row per token group
        const unsigned groups = (unsigned)((long)B * per_img);
        if (groups % G != 0) return BALF_ERR_ARG;              // (H, W multiples of 64: per_img is a multiple of 4 at C <= 128)
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(c0k, dim3(groups / G), dim3(cs_waves<C>() * G * 64), clds0, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(c1k, dim3(groups / G), dim3(cs_waves<C>() * G * 64), clds1, st, a));
    } else if constexpr ((C == 256 && BALF_NS256 != 0) || (C == 128 && BALF_NS128 != 0) || (C == 64 && BALF_NS64 != 0)) {
        constexpr int nlds = ns_lds_bytes<C>();
        static_assert(nlds <= (C >= 256 ? 160 : 80) * 1024, "N-split LDS image");
        auto n0 = stage_branch_kernel16_ns<C, CIN, 0>;
        auto n1 = stage_branch_kernel16_ns<C, CIN, 1>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(n0), hipFuncAttributeMaxDynamicSharedMemorySize, nlds) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(n1), hipFuncAttributeMaxDynamicSharedMemorySize, nlds) !=
                hipSuccess)
            return BALF_ERR_LAUNCH;
        per_img *= 4;                                      // one partial-sum row per wave pair
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(n0, dim3(nwg), dim3(512), nlds, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(n1, dim3(nwg), dim3(512), nlds, st, a));
    } else {
        per_img *= 4;                                      // generic kernels: one partial-sum row per wave
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(k0, dim3(nwg), dim3(256), lds, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(k1, dim3(nwg), dim3(256), lds, st, a));
    }
    BALF_PROF(4 * s + 2, st, {
        hipLaunchKernelGGL(se_reduce_kernel<C>, dim3(B * kSeChunks), dim3(256), 0, st, partial, per_img, chunk);
        hipLaunchKernelGGL(se_kernel<C>, dim3(B), dim3(256), 0, st, blob, kLayout.st[s], chunk,
                           1.0f / ((float)H * (float)W), scale, stage_fused<C>() ? 1 : 0);
    });
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

template <int C>
int run_pool16(int s, const float *T, const float *R, const float *scale, int B, int H, int W, float *out,
               hipStream_t st) {
    const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    BALF_PROF(4 * s + 3, st,
              hipLaunchKernelGGL(pool_kernel16<C>, dim3((unsigned)blocks), dim3(256), 0, st, T, R, scale, B, H, W, out));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

// Tail of stages 2-3 (stage_cs_kernel16<C, CIN, 2>): the stage input X, x1 (in R) and the SE scale -> the next stage's input.
template <int C, int CIN>
int run_tail_cs16(const float *blob, int s, const float *X, const float *R, const float *scale, int B, int H, int W, float *out,
                  hipStream_t st) {
    StageArgs a{blob, kLayout.st[s], X, nullptr, 0, 0, 0, 0, 0, B, H, W, nullptr, nullptr, const_cast<float *>(R), nullptr, scale, out};
    auto k = stage_cs_kernel16<C, CIN, 2>;
    constexpr int G = cs_groups<C>();
    constexpr int clds = cs_tail_lds_bytes<C>() * G;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, clds) != hipSuccess)
        return BALF_ERR_LAUNCH;
    const unsigned groups = (unsigned)((long)B * (H / 8) * (W / 8));
    if (groups % G != 0) return BALF_ERR_ARG;
    BALF_PROF(4 * s + 3, st, hipLaunchKernelGGL(k, dim3(groups / G), dim3(cs_waves<C>() * G * 64), clds, st, a));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

// Stage-1 tail (stage1_kernel16<2>): x1 (in R), the image and the SE scale -> the next stage's input.
int run_tail16(const float *blob, const float *X, const InputU8 &u8, const float *R, const float *scale, int B, int H, int W,
               float *out, hipStream_t st) {
    StageArgs a{blob, kLayout.st[0], X, u8.p, u8.ch, u8.h, u8.w, u8.top, u8.left, B, H, W, nullptr, nullptr,
                const_cast<float *>(R), nullptr, scale, out};
    auto k = u8.ch ? stage1_kernel16<2, true> : stage1_kernel16<2, false>;
    constexpr int lds = s1_lds_bytes<2>();
    static_assert(lds <= 160 * 1024, "stage-1 tail LDS image");
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return BALF_ERR_LAUNCH;
    const long groups = (long)B * (H / 8) * (W / 8);
    long b = (groups + s1_waves<2>() - 1) / s1_waves<2>();
    b = (b + 7) / 8 * 8;
    BALF_PROF(3, st, hipLaunchKernelGGL(k, dim3((unsigned)(b < 256 ? b : 256)), dim3(s1_waves<2>() * 64), lds, st, a));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

}  // namespace

#if BALF_STAMPS
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
                float *prob_dev, char *ws, const Plan &pl, hipStream_t st) {
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
        if ((rc = run_stage16<32, 3>(blob, 0, x, u8b, nb, Hp, Wp, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if (stage_fused<32>()) {
            if ((rc = run_tail16(blob, x, u8b, R, scale, nb, Hp, Wp, X2, st)) != BALF_OK) return rc;
        } else if ((rc = run_pool16<32>(0, T, R, scale, nb, Hp, Wp, X2, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<64, 32>(blob, 1, X2, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 2, Wp / 2, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if (stage_fused<64>()) {
            if ((rc = run_tail_cs16<64, 32>(blob, 1, X2, R, scale, nb, Hp / 2, Wp / 2, X3, st)) != BALF_OK) return rc;
        } else if ((rc = run_pool16<64>(1, T, R, scale, nb, Hp / 2, Wp / 2, X3, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<128, 64>(blob, 2, X3, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 4, Wp / 4, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if (stage_fused<128>()) {
            if ((rc = run_tail_cs16<128, 64>(blob, 2, X3, R, scale, nb, Hp / 4, Wp / 4, X4, st)) != BALF_OK) return rc;
        } else if ((rc = run_pool16<128>(2, T, R, scale, nb, Hp / 4, Wp / 4, X4, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<256, 128>(blob, 3, X4, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, h8, w8, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        HeadArgs ha{blob, kLayout.st[3], kLayout.head_w, kLayout.head_b, kLayout.head_alpha, kLayout.head_beta,
                    T, R, scale, nb, h8, w8,
                    logits_dev ? logits_dev + (size_t)b0 * kHeadN * h8 * w8 : nullptr,
                    prob_dev + (size_t)b0 * Hp * Wp};
        BALF_PROF(15, st,
                  hipLaunchKernelGGL(BALF_HEAD_NSPLIT ? head_kernel16_ns : head_kernel16,
                                     dim3((unsigned)((long)nb * h8 * w8 / 64)), dim3(256), 0, st, ha));
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}

}  // namespace balf
