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
#include "det_common.h"

namespace balf {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct HL {
    h8 hi, lo;
};

__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// three-product accumulate: (ah + al)(bh + bl) ~ ah bh + al bh + ah bl
__device__ __forceinline__ f4 mfma16x3(const HL &a, const HL &b, f4 c) {
    c = mfma16(a.lo, b.hi, c);
    c = mfma16(a.hi, b.lo, c);
    return mfma16(a.hi, b.hi, c);
}

__device__ __forceinline__ void split_pair(float v0, float v1, h2 &hi, h2 &lo) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);      // hi = rtz_f16(v); v - hi is exact in fp32
    hi = __builtin_bit_cast(h2, h);
    const fp16x2 l = __builtin_amdgcn_cvt_pkrtz(fmaf((float)hi[0], -1.0f, v0), fmaf((float)hi[1], -1.0f, v1));
    lo = __builtin_bit_cast(h2, l);
}

// two accumulator tiles (channels 16*2s + 4q + r and 16*(2s+1) + 4q + r) -> one K-step B fragment
__device__ __forceinline__ HL split8(const f4 &t0, const f4 &t1) {
    HL o;
    h2 h, l;
    split_pair(t0[0], t0[1], h, l); o.hi[0] = h[0]; o.hi[1] = h[1]; o.lo[0] = l[0]; o.lo[1] = l[1];
    split_pair(t0[2], t0[3], h, l); o.hi[2] = h[0]; o.hi[3] = h[1]; o.lo[2] = l[0]; o.lo[3] = l[1];
    split_pair(t1[0], t1[1], h, l); o.hi[4] = h[0]; o.hi[5] = h[1]; o.lo[4] = l[0]; o.lo[5] = l[1];
    split_pair(t1[2], t1[3], h, l); o.hi[6] = h[0]; o.hi[7] = h[1]; o.lo[6] = l[0]; o.lo[7] = l[1];
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
__device__ __forceinline__ HL load_frag_px(const float *base, long pix, int C, int ks, int q) {
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
            for (int p = 0; p < P; ++p) acc[NT0 + nt][p] = mfma16(a[nt].lo, b[p].hi, acc[NT0 + nt][p]);
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

constexpr int kBtPitch16 = kTokens + 8;        // halves per channel row of the transposed token tile

template <int C, int P>
constexpr int stage_lds_bytes16() {
    constexpr int slots = 4 * (C / 32) * P * 2048;
    constexpr int bt = 2 * P * C * kBtPitch16 * 2;
    return (slots > bt ? slots : bt) + 4 * C * 4;
}

template <int C, int CIN, int MODE>
__global__ __launch_bounds__(256, (C >= 256 ? 1 : 2)) void stage_branch_kernel16(StageArgs A) {
    constexpr int P = StageP<C>::P;
    constexpr int NT = C / 16, KS = C / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int main_bytes = stage_lds_bytes16<C, P>() - 4 * C * 4;
    _Float16 *bT = reinterpret_cast<_Float16 *>(smem_raw);                 // [hi|lo][p][c][pitch]
    float *red = reinterpret_cast<float *>(smem_raw + main_bytes);         // [4][C]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    h8 *slot = reinterpret_cast<h8 *>(smem_raw) + wave * (KS * P * 2 * 64);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    const int H = A.H, W = A.W;
    const int cols = W / 8 / P;
    const int per_img = (H / 8) * cols;
    const int n = blockIdx.x / per_img;
    const int rem = blockIdx.x - n * per_img;
    const int iy0 = rem / cols, ix0 = (rem - iy0 * cols) * P;
    const int tok = 16 * wave + li, ty = tok >> 3, tx = tok & 7;
    long pix[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int y, x;
        if (MODE == 0) { y = ty * (H / 8) + iy0; x = tx * (W / 8) + ix0 + p; }
        else           { y = 8 * iy0 + ty;       x = 8 * (ix0 + p) + tx; }
        pix[p] = ((long)n * H + y) * W + x;
    }

    // ---- x0 = relu(conv0(X)) ----
    f4 x0[NT][P];
    if constexpr (CIN == 3) {
        float in[P][3];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const long hw = (long)H * W;
            const long o = pix[p] - (long)n * hw;
#pragma unroll
            for (int k = 0; k < 3; ++k) in[p][k] = A.X[((long)n * 3 + k) * hw + o];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f4 bias = ldg4(blob + S.conv0_b + 16 * nt + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float *wr = blob + S.conv0_w + (16 * nt + 4 * q + r) * 3;
                const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    x0[nt][p][r] = fmaxf(bias[r] + in[p][0] * w0 + in[p][1] * w1 + in[p][2] * w2, 0.0f);
            }
        }
    } else {
        init_bias(x0, blob + S.conv0_b, q);
        gemm16<NT, P>(x0, blob + S.conv0_w, 0, CIN / 32, 0, CIN / 32, lane,
                      [&](int kk, int p) { return load_frag_px(A.X, pix[p], CIN, kk, q); });
        relu(x0);
    }

    {
        f4 h[NT][P];
        layernorm_plain(x0, h);
        store_slot16(slot, h, lane);
    }
    auto from_slot = [&](int kk, int p) {
        HL o;
        o.hi = slot[((kk * P + p) * 2 + 0) * 64 + lane];
        o.lo = slot[((kk * P + p) * 2 + 1) * 64 + lane];
        return o;
    };
    f4 z[NT][P];
    init_bias(z, blob + S.q1_b + MODE * C, q);
    gemm16<NT, P>(z, blob + S.q1_w, MODE * NT, KS, 0, KS, lane, from_slot);
    gelu(z);

    {
        f4 h[NT][P];
        layernorm_plain(z, h);
        store_slot16(slot, h, lane);
    }
    f4 ga[NT][P];
    init_bias(ga, blob + Br.d1_b, q);
    gemm16<NT, P>(ga, blob + Br.d1_w, 0, KS, 0, KS, lane, from_slot);
    gelu(ga);
    {
        f4 gb[NT][P];
        init_bias(gb, blob + Br.d1_b + C, q);
        gemm16<NT, P>(gb, blob + Br.d1_w, NT, KS, 0, KS, lane, from_slot);
        gelu(gb);
        layernorm(gb, gb, blob + Br.gln_g, blob + Br.gln_b, q);
        __syncthreads();
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
    __syncthreads();
    {
        // mix^T[c][g'] = sum_g bT[c][g] * Wmix[g'][g]: A = bT rows (channels), B = natural-order Wmix fragments
        const char *wm = reinterpret_cast<const char *>(blob + Br.mix_w) + (wave * 2) * 2048 + lane * 16;
        HL w0, w1;
        w0.hi = *reinterpret_cast<const h8 *>(wm);        w0.lo = *reinterpret_cast<const h8 *>(wm + 1024);
        w1.hi = *reinterpret_cast<const h8 *>(wm + 2048); w1.lo = *reinterpret_cast<const h8 *>(wm + 3072);
        const float mb1 = blob[Br.mix_b + tok] + 1.0f;
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
    __syncthreads();
    store_slot16(slot, ga, lane);
    f4 o[NT][P];
    init_bias(o, blob + Br.d2_b, q);
    gemm16<NT, P>(o, blob + Br.d2_w, 0, KS, 0, KS, lane, from_slot);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

    if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                store_frag_px(A.U, pix[p], C, ks, q, split8(o[2 * ks][p], o[2 * ks + 1][p]));
        return;
    } else {
        store_slot16(slot, o, lane);
        f4 x1[NT][P];
        init_bias(x1, blob + S.q2_b, q);
        gemm16<NT, P>(x1, blob + S.q2_w, 0, 2 * KS, KS, KS, lane, from_slot);
        gemm16<NT, P>(x1, blob + S.q2_w, 0, 2 * KS, 0, KS, lane,
                      [&](int kk, int p) { return load_frag_px(A.U, pix[p], C, kk, q); });
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x1[nt][p] += x0[nt][p];
                *reinterpret_cast<f4 *>(A.R + pix[p] * C + 16 * nt + 4 * q) = x1[nt][p] + x0[nt][p];
            }
        layernorm_plain(x1, x1);
        store_slot16(slot, x1, lane);
        f4 m1[NT][P];
        init_bias(m1, blob + S.r1_b, q);
        gemm16<NT, P>(m1, blob + S.r1_w, 0, KS, 0, KS, lane, from_slot);
        lrelu(m1);
        store_slot16(slot, m1, lane);
        f4 t[NT][P];
        init_bias(t, blob + S.r2_b, q);
        gemm16<NT, P>(t, blob + S.r2_w, 0, KS, 0, KS, lane, from_slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                *reinterpret_cast<f4 *>(A.T + pix[p] * C + 16 * nt + 4 * q) = t[nt][p];
                s += t[nt][p];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[r];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (li == 0) red[wave * C + 16 * nt + 4 * q + r] = v;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256)
            A.partial[(long)blockIdx.x * C + c] = (red[c] + red[C + c]) + (red[2 * C + c] + red[3 * C + c]);
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

template <int C, int CIN>
int run_stage16(const float *blob, int s, const float *X, int B, int H, int W, float *U, float *T, float *R,
                float *partial, float *chunk, float *scale, hipStream_t st) {
    constexpr int P = StageP<C>::P;
    constexpr int lds = stage_lds_bytes16<C, P>();
    StageArgs a{blob, kLayout.st[s], X, B, H, W, U, T, R, partial};
    const int per_img = (H / 8) * (W / 8 / P);
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
    BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(k0, dim3(nwg), dim3(256), lds, st, a));
    BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(k1, dim3(nwg), dim3(256), lds, st, a));
    BALF_PROF(4 * s + 2, st, {
        hipLaunchKernelGGL(se_reduce_kernel<C>, dim3(B * kSeChunks), dim3(256), 0, st, partial, per_img, chunk);
        hipLaunchKernelGGL(se_kernel<C>, dim3(B), dim3(256), 0, st, blob, kLayout.st[s], chunk,
                           1.0f / ((float)H * (float)W), scale);
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

}  // namespace

int forward_f16(const float *blob, const float *x_nchw_dev, int B, int Hp, int Wp, float *logits_dev, float *prob_dev,
                char *ws, const Plan &pl, hipStream_t st) {
    float *U = reinterpret_cast<float *>(ws + pl.off_U), *T = reinterpret_cast<float *>(ws + pl.off_T),
          *R = reinterpret_cast<float *>(ws + pl.off_R), *partial = reinterpret_cast<float *>(ws + pl.off_partial),
          *chunk = reinterpret_cast<float *>(ws + pl.off_chunk), *scale = reinterpret_cast<float *>(ws + pl.off_scale);
    float *X2 = reinterpret_cast<float *>(ws + pl.off_X[0]), *X3 = reinterpret_cast<float *>(ws + pl.off_X[1]),
          *X4 = reinterpret_cast<float *>(ws + pl.off_X[2]);
    const int h8 = Hp / 8, w8 = Wp / 8;

    for (int b0 = 0; b0 < B; b0 += pl.mb) {
        const int nb = (B - b0 < pl.mb) ? (B - b0) : pl.mb;
        const float *x = x_nchw_dev + (size_t)b0 * 3 * Hp * Wp;
        int rc;
        if ((rc = run_stage16<32, 3>(blob, 0, x, nb, Hp, Wp, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_pool16<32>(0, T, R, scale, nb, Hp, Wp, X2, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<64, 32>(blob, 1, X2, nb, Hp / 2, Wp / 2, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_pool16<64>(1, T, R, scale, nb, Hp / 2, Wp / 2, X3, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<128, 64>(blob, 2, X3, nb, Hp / 4, Wp / 4, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_pool16<128>(2, T, R, scale, nb, Hp / 4, Wp / 4, X4, st)) != BALF_OK) return rc;
        if ((rc = run_stage16<256, 128>(blob, 3, X4, nb, h8, w8, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        HeadArgs ha{blob, kLayout.st[3], kLayout.head_w, kLayout.head_b, kLayout.head_alpha, kLayout.head_beta,
                    T, R, scale, nb, h8, w8,
                    logits_dev ? logits_dev + (size_t)b0 * kHeadN * h8 * w8 : nullptr,
                    prob_dev + (size_t)b0 * Hp * Wp};
        BALF_PROF(15, st,
                  hipLaunchKernelGGL(head_kernel16, dim3((unsigned)((long)nb * h8 * w8 / 64)), dim3(256), 0, st, ha));
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}

}  // namespace balf
