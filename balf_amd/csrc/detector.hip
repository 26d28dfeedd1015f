// BALF detector forward on gfx950: fp32 path on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains).
//
// Replaces MLP_MA_DECODER.forward (/root/reference/balf/model/mlp_ma_decoder.py:278-285): four Down
// stages (:223-244) of the multi-axis gated MLP (:132-149, :57-70, :104-117, :36-42, :83-89) with
// the residual channel-attention block (:185-199, :166-171), then DetectorHead.forward
// (/root/reference/balf/model/decoder.py:16-30) with pixel_shuffle
// (/root/reference/balf/utils/tensor_op.py:15-27).
//
// Design ("pixel on the lane"): every Linear is computed transposed, D[out_ch][pixel] = W * act^T, so
// the MFMA accumulator leaves a pixel in lane&15 and its channels 16*t + 4*(lane>>4) + r in register
// r of tile t -- exactly the B-operand layout of the next Linear.  A wave therefore owns 16*P pixels
// and carries them through a whole chain of Linear / LayerNorm / GELU layers in registers; LayerNorm is
// a per-lane register reduction plus two cross-quarter exchanges.  Chained inputs are parked in a
// wave-private LDS slot only so that the K loop can be a runtime loop (small code) -- no barrier.
// The only cross-wave step is the 64x64 token mix (a sum over pixels = over lanes): its operand is
// transposed through LDS ([channel][token]) and consumed as the MFMA A operand.
//
// Kernels per stage:
//   stage_branch_kernel<.., MODE=0>  64-pixel groups = the 8x8 grid cells at one in-cell offset:
//        x0 = relu(conv0 X) -> LN -> dense1[:C] -> GELU = u -> grid gMLP -> u'   (written to HBM)
//   stage_branch_kernel<.., MODE=1>  64-pixel groups = 8x8 blocks:
//        x0 -> LN -> dense1[C:] -> GELU = v -> block gMLP -> v';  x1 = dense2(cat[u', v']) + x0;
//        t = conv2(lrelu(conv1(LN x1)));  writes t, r = x1 + x0 and per-workgroup channel sums of t
//   se_kernel        per image: mean(t) -> C/4 -> C MLP -> sigmoid                (deterministic order)
//   pool_kernel      stages 1-3: x2 = t*s + r, 2x2 max pool -> next stage's NHWC input
//   head_kernel      stage 4: x2 -> conv2 -> relu -> dense(256->65) -> BN -> softmax -> pixel shuffle
#include <atomic>

#include "det_common.h"

namespace balf {
namespace {
// ------------------------------------------------------------------------------------------------
// GEMM: acc[0 .. NTT)[P] += W[weight row tiles wnt0 .. wnt0 + NTT, K tiles kt0 .. kt0 + KTN) * in
//   w     : fragment-ordered weights (layout.h), KTtot = K/16 tiles per weight row-tile
//   bload : (kk, p) -> f4 holding input channels 16*(kt0'+kk) + 4*q + {0..3} of pixel tile p
// The output row tiles are processed in chunks of CH (bounds the live weight fragments); the whole sequence of
// steps (chunk, kk) is ONE software pipeline, unrolled at compile time, with the weight fragments of step v + D - 1 requested
// before the MFMAs of step v: a step is CH * P * 4 MFMAs of 32 cycles -- 512 cycles at C >= 128 -- and a fragment comes from L2
// in ~700 (a tenth of them from beyond L2), so the one-step lookahead of rounds 1-4 (D = 2) left every step waiting
// (matrix pipe 0.57-0.65 busy in every stage, profiles/r5_pmc.json).  Accumulation order per output is unchanged
// (kk ascending, then the four K-slots): results are bit-identical for every D.
// B fragments come from the wave's LDS slot (short latency: DB = 2 steps in flight) or from global memory (DB = D).
// ------------------------------------------------------------------------------------------------
#ifndef BALF_F32_D32
#define BALF_F32_D32 2
#endif
#ifndef BALF_F32_D64
#define BALF_F32_D64 2
#endif
#ifndef BALF_F32_D128
#define BALF_F32_D128 2
#endif
#ifndef BALF_F32_D256
#define BALF_F32_D256 4
#endif
template <int C> constexpr int gemm_depth() {
    return C == 32 ? BALF_F32_D32 : C == 64 ? BALF_F32_D64 : C == 128 ? BALF_F32_D128 : BALF_F32_D256;
}

template <int NTT, int P, int KTN, int D, bool BGLOBAL, typename BL>
__device__ __forceinline__ void gemm(f4 (&acc)[NTT][P], const float *w, int wnt0, int KTtot, int kt0, int lane, BL bload) {
    constexpr int CH = (P >= 4) ? 2 : 4;
    constexpr int NCH = (NTT + CH - 1) / CH, STEPS = NCH * KTN;
    constexpr int DB = BGLOBAL ? D : 2;
    static_assert(D >= 2 && DB <= D, "ring depths");
    // uniform byte pointer of weight tile (wnt0, kt0) + a 32-bit per-lane offset: lets the compiler
    // use the SGPR-base form of global_load (no 64-bit VALU address arithmetic per load)
    const char *wbase = reinterpret_cast<const char *>(reinterpret_cast<const f4 *>(w) + ((size_t)wnt0 * KTtot + kt0) * 64);
    const unsigned lane_off = (unsigned)lane * 16u;
    const unsigned nstride = (unsigned)KTtot * 1024u;          // bytes between weight row-tiles
    f4 a[D][CH], b[DB][P];
    auto ntc = [](int c) { return (NTT - c * CH) < CH ? (NTT - c * CH) : CH; };
    auto load_a = [&](int v) {
        const int c = v / KTN, kk = v % KTN;
#pragma unroll
        for (int nt = 0; nt < CH; ++nt)
            if (nt < ntc(c))
                a[v % D][nt] = *reinterpret_cast<const f4 *>(wbase + ((unsigned)(c * CH + nt) * nstride + (unsigned)kk * 1024u) + lane_off);
    };
    auto load_b = [&](int v) {
#pragma unroll
        for (int p = 0; p < P; ++p) b[v % DB][p] = bload(v % KTN, p);
    };
#pragma unroll
    for (int v = 0; v < D - 1; ++v) if (v < STEPS) load_a(v);
#pragma unroll
    for (int v = 0; v < DB - 1; ++v) if (v < STEPS) load_b(v);
#pragma unroll
    for (int v = 0; v < STEPS; ++v) {
        if (v + D - 1 < STEPS) load_a(v + D - 1);
        if (v + DB - 1 < STEPS) load_b(v + DB - 1);
        __builtin_amdgcn_sched_barrier(0);
        const int c = v / KTN;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < CH; ++nt)
                if (nt < ntc(c))
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[c * CH + nt][p] = mfma4(a[v % D][nt][j], b[v % DB][p][j], acc[c * CH + nt][p]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int C, int CIN, int MODE>
__global__ __launch_bounds__(256, StageP<C>::OCC) void stage_branch_kernel(StageArgs A) {
    constexpr bool PK = (MODE == 0) && (C == 32);      // packed VALU math only where it measured faster
    constexpr int P = StageP<C>::P;
    constexpr int NT = C / 16, GD = gemm_depth<C>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f4 *smem = reinterpret_cast<f4 *>(smem_raw);
    float *bT = reinterpret_cast<float *>(smem_raw);
    constexpr int main_bytes = stage_lds_bytes<C, P>() - 4 * C * 4;
    float *red = reinterpret_cast<float *>(smem_raw + main_bytes);                    // [4][C]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    f4 *slot = smem + wave * (NT * P * 64);
    const float *blob = A.blob;
    const StageOff &S = A.off;
    const BranchOff &Br = S.br[MODE];

    // ---- which 64-pixel groups does this workgroup own, and which pixel is this lane's ----
    const int H = A.H, W = A.W;
    const int cols = W / 8 / P;                       // work items per row of items
    const int per_img = (H / 8) * cols;
    const int n = blockIdx.x / per_img;
    const int rem = blockIdx.x - n * per_img;
    const int iy0 = rem / cols, ix0 = (rem - iy0 * cols) * P;
    const int tok = 16 * wave + li, ty = tok >> 3, tx = tok & 7;
    long pix[P];                                      // flat pixel index (n, y, x) of tile p
    int py[P], px_[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int y, x;
        if (MODE == 0) { y = ty * (H / 8) + iy0; x = tx * (W / 8) + ix0 + p; }
        else           { y = 8 * iy0 + ty;       x = 8 * (ix0 + p) + tx; }
        pix[p] = ((long)n * H + y) * W + x;
        py[p] = y; px_[p] = x;
    }

    // ---- x0 = relu(conv0(X)) ----
    f4 x0[NT][P];
    if constexpr (CIN == 3) {
        float in[P][3];
#pragma unroll
        for (int p = 0; p < P; ++p) load_input3(A, blob + kLayout.u8_lut, n, py[p], px_[p], in[p]);
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
        gemm<NT, P, CIN / 16, GD, true>(x0, blob + S.conv0_w, 0, CIN / 16, 0, lane,
                    [&](int kk, int p) { return ldg4(A.X + pix[p] * CIN + 16 * kk + 4 * q); });
        relu(x0);
    }

    // ---- z = GELU(dense1[MODE half](LN(x0))) : u (grid) or v (block) ----
    {
        f4 h[NT][P];
        layernorm_plain<PK>(x0, h);                        // q.norm affine folded into dense1
        store_slot(slot, h, lane);
    }
    auto from_slot = [&](int kk, int p) { return slot[(kk * P + p) * 64 + lane]; };
    f4 z[NT][P];
    init_bias(z, blob + S.q1_b + MODE * C, q);
    gemm<NT, P, NT, GD, false>(z, blob + S.q1_w, MODE * NT, NT, 0, lane, from_slot);
    gelu<PK>(z);

    // ---- gMLP branch on z ----
    {
        f4 h[NT][P];
        layernorm_plain<PK>(z, h);                         // branch .norm affine folded into its dense1
        store_slot(slot, h, lane);
    }
    f4 ga[NT][P];                                      // gate input a = first C outputs of dense1
    init_bias(ga, blob + Br.d1_b, q);
    gemm<NT, P, NT, GD, false>(ga, blob + Br.d1_w, 0, NT, 0, lane, from_slot);
    gelu<PK>(ga);
    {
        f4 gb[NT][P];                                  // b = last C outputs, normalised, then token-mixed
        init_bias(gb, blob + Br.d1_b + C, q);
        gemm<NT, P, NT, GD, false>(gb, blob + Br.d1_w, NT, NT, 0, lane, from_slot);
        gelu<PK>(gb);
        layernorm<PK>(gb, gb, blob + Br.gln_g, blob + Br.gln_b, q);
        __syncthreads();                               // every wave is done reading its slot
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) bT[(p * C + 16 * nt + 4 * q + r) * kBtPitch + tok] = gb[nt][p][r];
    }
    __syncthreads();
    {
        // mix^T[c][g'] = sum_g bT[c][g] * Wmix[g'][g]; this wave produces its own 16 tokens g'
        const f4 *wm = reinterpret_cast<const f4 *>(blob + Br.mix_w) + (wave * 4) * 64 + lane;
        const f4 wm0 = wm[0], wm1 = wm[64], wm2 = wm[128], wm3 = wm[192];
        const float mb1 = blob[Br.mix_b + tok] + 1.0f;
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const float *row = bT + (p * C + 16 * ct + li) * kBtPitch + 4 * q;
                const f4 a0 = *reinterpret_cast<const f4 *>(row), a1 = *reinterpret_cast<const f4 *>(row + 16),
                         a2 = *reinterpret_cast<const f4 *>(row + 32), a3 = *reinterpret_cast<const f4 *>(row + 48);
                f4 m = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 4; ++j) m = mfma4(a0[j], wm0[j], m);
#pragma unroll
                for (int j = 0; j < 4; ++j) m = mfma4(a1[j], wm1[j], m);
#pragma unroll
                for (int j = 0; j < 4; ++j) m = mfma4(a2[j], wm2[j], m);
#pragma unroll
                for (int j = 0; j < 4; ++j) m = mfma4(a3[j], wm3[j], m);
#pragma unroll
                for (int r = 0; r < 4; ++r) ga[ct][p][r] *= (m[r] + mb1);
            }
    }
    __syncthreads();                                   // bT is dead; slots are wave-private again
    store_slot(slot, ga, lane);
    f4 o[NT][P];
    init_bias(o, blob + Br.d2_b, q);
    gemm<NT, P, NT, GD, false>(o, blob + Br.d2_w, 0, NT, 0, lane, from_slot);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) o[nt][p] += z[nt][p];

    if constexpr (MODE == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                *reinterpret_cast<f4 *>(A.U + pix[p] * C + 16 * nt + 4 * q) = o[nt][p];
        return;
    } else {
        // ---- x1 = dense2(cat[u', v']) + x0 ----
        store_slot(slot, o, lane);                     // v'
        f4 x1[NT][P];
        init_bias(x1, blob + S.q2_b, q);
        gemm<NT, P, NT, GD, false>(x1, blob + S.q2_w, 0, 2 * NT, NT, lane, from_slot);
        gemm<NT, P, NT, GD, true>(x1, blob + S.q2_w, 0, 2 * NT, 0, lane,
                    [&](int kk, int p) { return ldg4(A.U + pix[p] * C + 16 * kk + 4 * q); });
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                x1[nt][p] += x0[nt][p];
                *reinterpret_cast<f4 *>(A.R + pix[p] * C + 16 * nt + 4 * q) = x1[nt][p] + x0[nt][p];
            }
        // ---- t = conv2(lrelu(conv1(LN(x1)))) ----
        layernorm_plain<PK>(x1, x1);                       // RCAB .norm affine folded into conv1
        store_slot(slot, x1, lane);
        f4 m1[NT][P];
        init_bias(m1, blob + S.r1_b, q);
        gemm<NT, P, NT, GD, false>(m1, blob + S.r1_w, 0, NT, 0, lane, from_slot);
        lrelu(m1);
        store_slot(slot, m1, lane);
        f4 t[NT][P];
        init_bias(t, blob + S.r2_b, q);
        gemm<NT, P, NT, GD, false>(t, blob + S.r2_w, 0, NT, 0, lane, from_slot);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f4 s = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                *reinterpret_cast<f4 *>(A.T + pix[p] * C + 16 * nt + 4 * q) = t[nt][p];
                s += t[nt][p];
            }
            // channel sums over this wave's 16*P pixels: reduce over the 16 lanes of the quarter
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

// ------------------------------------------------------------------------------------------------
// x2 = t * s + r, MaxPool2d(2) -> next stage input, NHWC                 (mlp_ma_decoder.py:232-236)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ T, const float *__restrict__ R,
                                                   const float *__restrict__ scale, int B, int H, int W,
                                                   float *__restrict__ out) {
    constexpr int G = C / 4;
    const long total = (long)B * (H / 2) * (W / 2) * G;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        long pxy = i / G;
        const int xo = (int)(pxy % (W / 2));
        pxy /= (W / 2);
        const int yo = (int)(pxy % (H / 2));
        const int n = (int)(pxy / (H / 2));
        const f4 s = ldg4(scale + (long)n * C + 4 * g);
        const long base = (((long)n * H + 2 * yo) * W + 2 * xo) * C + 4 * g;
        f4 m;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long o = base + ((long)dy * W + dx) * C;
                const f4 v = ldg4(T + o) * s + ldg4(R + o);
                if (dy == 0 && dx == 0) m = v;
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], v[r]);
            }
        *reinterpret_cast<f4 *>(out + i * 4) = m;
    }
}

__global__ __launch_bounds__(256, 2) void head_kernel(HeadArgs A) {
    constexpr int C = 256, NT = 16, HT = kHeadNPad / 16;
    __shared__ __attribute__((aligned(16))) f4 smem[4 * NT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, li = lane & 15;
    f4 *slot = smem + wave * (NT * 64);
    const float *blob = A.blob;
    const long hw = (long)A.h * A.w;
    const long pixel = ((long)blockIdx.x * 4 + wave) * 16 + li;        // over B*h*w (multiple of 64)
    const int n = (int)(pixel / hw);
    const long o = pixel - (long)n * hw;
    const int i = (int)(o / A.w), j = (int)(o - (long)i * A.w);

    f4 f[NT][1];
    init_bias(f, blob + A.off.conv2_b, q);
    gemm<NT, 1, NT, gemm_depth<256>(), true>(f, blob + A.off.conv2_w, 0, NT, 0, lane, [&](int kk, int) {
        const int c = 16 * kk + 4 * q;
        return ldg4(A.T + pixel * C + c) * ldg4(A.scale + (long)n * C + c) + ldg4(A.R + pixel * C + c);
    });
    relu(f);
    store_slot(slot, f, lane);
    f4 z[HT][1];
    init_bias(z, blob + A.head_b, q);
    gemm<HT, 1, NT, gemm_depth<256>(), false>(z, blob + A.head_w, 0, NT, 0, lane, [&](int kk, int) { return slot[kk * 64 + lane]; });

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
    if (!(sum > 0.0f && sum < INFINITY)) status_raise(A.status, 0 /* BALF_STATUS_SCORE */);
    const float inv = 1.0f / sum;
    const int Wp = 8 * A.w;
#pragma unroll
    for (int t = 0; t < 4; ++t) {          // channels 0..63: dy = 2t + (q>>1), dx = 4(q&1) + r
        const f4 pr = z[t][0] * inv;
        float *dst = A.prob + ((long)n * 8 * A.h + 8 * i + 2 * t + (q >> 1)) * Wp + 8 * j + 4 * (q & 1);
        *reinterpret_cast<f4 *>(dst) = pr;
    }
}

#include "stage1_f32.h"

#ifndef BALF_F32_S1_GENERIC
#define BALF_F32_S1_GENERIC 0     // 1: stage 1 on the generic kernel (rounds 1-4), for A/B timing
#endif

// hipFuncSetAttribute is a driver round trip (tens of microseconds): once per kernel and device of the process, not per launch
// (as ensure_kernel_attributes of detector_f16.hip; a single-image call on the fp32 path made four of them)
template <auto Kernel>
bool allow_lds32(int bytes) {
    static std::atomic<unsigned long long> done{0};              // bit d: device d has the attribute
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done.load(std::memory_order_acquire) >> dev & 1) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return false;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

template <int C, int CIN>
int run_stage(int *status, const float *blob, int s, const float *X, const InputU8 &u8, int B, int H, int W, float *U, float *T, float *R,
              float *partial, float *chunk, float *scale, hipStream_t st) {
    StageArgs a{blob, kLayout.st[s], X, u8.p, u8.ch, u8.h, u8.w, u8.top, u8.left, B, H, W, U, T, R, partial};
    int per_img;                                         // rows of partial channel sums per image
    if constexpr (C == 32 && !BALF_F32_S1_GENERIC) {
        // persistent wave-owns-group kernels (stage1_f32.h): one workgroup per CU at most, a multiple of 8 (XCD-aware schedule)
        per_img = (H / 8) * (W / 8);
        const long groups = (long)B * per_img;
        long wgs = (groups + kF1NW - 1) / kF1NW;
        wgs = (wgs + 7) / 8 * 8;
        if (wgs > 256) wgs = 256;
        if (!allow_lds32<stage1_kernel32<0>>(f1_lds_bytes<0>()) || !allow_lds32<stage1_kernel32<1>>(f1_lds_bytes<1>())) return BALF_ERR_LAUNCH;
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(stage1_kernel32<0>, dim3((unsigned)wgs), dim3(kF1NW * 64), f1_lds_bytes<0>(), st, a));
#if BALF_F32_DBG
        if (getenv("BALF_DEBUG_STOP_STAGE")) return BALF_OK;
#endif
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(stage1_kernel32<1>, dim3((unsigned)wgs), dim3(kF1NW * 64), f1_lds_bytes<1>(), st, a));
    } else {
        constexpr int P = StageP<C>::P;
        constexpr int lds = stage_lds_bytes<C, P>();
        per_img = (H / 8) * (W / 8 / P);
        const int nwg = B * per_img;
        auto k0 = stage_branch_kernel<C, CIN, 0>;
        auto k1 = stage_branch_kernel<C, CIN, 1>;
        if (lds > 48 * 1024) {
            if (!allow_lds32<stage_branch_kernel<C, CIN, 0>>(lds) || !allow_lds32<stage_branch_kernel<C, CIN, 1>>(lds)) return BALF_ERR_LAUNCH;
        }
        BALF_PROF(4 * s + 0, st, hipLaunchKernelGGL(k0, dim3(nwg), dim3(256), lds, st, a));
        BALF_PROF(4 * s + 1, st, hipLaunchKernelGGL(k1, dim3(nwg), dim3(256), lds, st, a));
    }
    BALF_PROF(4 * s + 2, st, {
        hipLaunchKernelGGL(se_reduce_kernel<C>, dim3(B * kSeChunks), dim3(256), 0, st, partial, per_img, chunk);
        hipLaunchKernelGGL(se_kernel<C>, dim3(B), dim3(256), 0, st, blob, kLayout.st[s], chunk,
                           1.0f / ((float)H * (float)W), scale, 0, status);
    });
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

template <int C>
int run_pool(int s, const float *T, const float *R, const float *scale, int B, int H, int W, float *out,
             hipStream_t st) {
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    BALF_PROF(4 * s + 3, st,
              hipLaunchKernelGGL(pool_kernel<C>, dim3((unsigned)blocks), dim3(256), 0, st, T, R, scale, B, H, W, out));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

}  // namespace

int forward_f32(const float *blob, const float *x_nchw_dev, const InputU8 &u8, int B, int Hp, int Wp, float *logits_dev,
                float *prob_dev, char *ws, const Plan &pl, int *status, hipStream_t st) {
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
        if ((rc = run_stage<32, 3>(status, blob, 0, x, u8b, nb, Hp, Wp, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
#if BALF_F32_DBG
        if (getenv("BALF_DEBUG_STOP_STAGE")) return BALF_OK;       // (tests/experiments/f32_s1_debug.py: the tap in U must survive)
#endif
        if ((rc = run_pool<32>(0, T, R, scale, nb, Hp, Wp, X2, st)) != BALF_OK) return rc;
        if ((rc = run_stage<64, 32>(status, blob, 1, X2, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 2, Wp / 2, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_pool<64>(1, T, R, scale, nb, Hp / 2, Wp / 2, X3, st)) != BALF_OK) return rc;
        if ((rc = run_stage<128, 64>(status, blob, 2, X3, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, Hp / 4, Wp / 4, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        if ((rc = run_pool<128>(2, T, R, scale, nb, Hp / 4, Wp / 4, X4, st)) != BALF_OK) return rc;
        if ((rc = run_stage<256, 128>(status, blob, 3, X4, InputU8{nullptr, 0, 0, 0, 0, 0}, nb, h8, w8, U, T, R, partial, chunk, scale, st)) != BALF_OK) return rc;
        HeadArgs ha{blob, kLayout.st[3], kLayout.head_w, kLayout.head_b, kLayout.head_alpha, kLayout.head_beta,
                    T, R, scale, nb, h8, w8,
                    logits_dev ? logits_dev + (size_t)b0 * kHeadN * h8 * w8 : nullptr,
                    prob_dev + (size_t)b0 * Hp * Wp, status};
        BALF_PROF(15, st,
                  hipLaunchKernelGGL(head_kernel, dim3((unsigned)((long)nb * h8 * w8 / 64)), dim3(256), 0, st, ha));
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}

}  // namespace balf

extern "C" size_t balf_forward_workspace_bytes(int B, int Hp, int Wp) {
    if (B <= 0 || Hp <= 0 || Wp <= 0 || Hp % 64 || Wp % 64 || (long)Hp * Wp > (1L << 25)) return 0;
    return balf::make_plan(B, Hp, Wp).total;
}

extern "C" int balf_forward_micro_batch(int B, int Hp, int Wp) {
    if (B <= 0 || Hp <= 0 || Wp <= 0 || Hp % 64 || Wp % 64 || (long)Hp * Wp > (1L << 25)) return 0;   // (a shape balf_forward refuses)
    return balf::make_plan(B, Hp, Wp).mb;
}

static int forward_common(const void *packed_dev, int precision, const float *x_nchw_dev, const balf::InputU8 &u8, int B,
                          int Hp, int Wp, float *logits_dev, float *prob_dev, void *workspace_dev,
                          size_t workspace_bytes, int *status_dev, void *stream) {
    if (!packed_dev || !prob_dev || !workspace_dev) return BALF_ERR_ARG;
    if (precision != BALF_PREC_FP32 && precision != BALF_PREC_FP16) return BALF_ERR_ARG;
    if (B <= 0 || Hp <= 0 || Wp <= 0) return BALF_ERR_ARG;
    if (Hp % 64 || Wp % 64) return BALF_ERR_SHAPE;
    if ((long)B * Hp * Wp * 32 > 0x7fffffffffL) return BALF_ERR_SHAPE;
    // the kernels address a pixel's row inside ONE image with 32-bit byte offsets (up to 128 B per stage-1 pixel): 2^25 pixels
    // per padded image (5792 x 5792) is the limit; it is also what one micro-batch holds (BALF_MB_PIXELS)
    if ((long)Hp * Wp > (1L << 25)) return BALF_ERR_SHAPE;
    const balf::Plan pl = balf::make_plan(B, Hp, Wp);
    if (workspace_bytes < pl.total) return BALF_ERR_WORKSPACE;
    const float *blob = static_cast<const float *>(packed_dev);
    char *ws = static_cast<char *>(workspace_dev);
    return precision == BALF_PREC_FP32
               ? balf::forward_f32(blob, x_nchw_dev, u8, B, Hp, Wp, logits_dev, prob_dev, ws, pl, status_dev, (hipStream_t)stream)
               : balf::forward_f16(blob, x_nchw_dev, u8, B, Hp, Wp, logits_dev, prob_dev, ws, pl, status_dev, (hipStream_t)stream);
}

extern "C" int balf_forward_status(const void *packed_dev, int precision, const float *x_nchw_dev, int B, int Hp, int Wp,
                                   float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                                   int *status_dev, void *stream) {
    if (!x_nchw_dev) return BALF_ERR_ARG;
    return forward_common(packed_dev, precision, x_nchw_dev, balf::InputU8{nullptr, 0, 0, 0, 0, 0}, B, Hp, Wp, logits_dev,
                          prob_dev, workspace_dev, workspace_bytes, status_dev, stream);
}

extern "C" int balf_forward(const void *packed_dev, int precision, const float *x_nchw_dev, int B, int Hp, int Wp,
                            float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                            void *stream) {
    return balf_forward_status(packed_dev, precision, x_nchw_dev, B, Hp, Wp, logits_dev, prob_dev, workspace_dev,
                               workspace_bytes, nullptr, stream);
}

extern "C" int balf_forward_u8_status(const void *packed_dev, int precision, const unsigned char *image_dev, int channels,
                                      int B, int H, int W, float *logits_dev, float *prob_dev, void *workspace_dev,
                                      size_t workspace_bytes, int *status_dev, void *stream) {
    if (!image_dev || (channels != 1 && channels != 3) || H <= 0 || W <= 0) return BALF_ERR_ARG;
    // make_shape_even + mod_padding_symmetric(64) (test_utils.py:16-32): padded size and where the image lands
    const int He = H + (H & 1), We = W + (W & 1);
    const int Hp = He % 64 ? (He / 64 + 1) * 64 : He, Wp = We % 64 ? (We / 64 + 1) * 64 : We;
    const balf::InputU8 u8{image_dev, channels, H, W, (Hp - He) / 2, (Wp - We) / 2};
    return forward_common(packed_dev, precision, nullptr, u8, B, Hp, Wp, logits_dev, prob_dev, workspace_dev,
                          workspace_bytes, status_dev, stream);
}

extern "C" int balf_forward_u8(const void *packed_dev, int precision, const unsigned char *image_dev, int channels, int B,
                               int H, int W, float *logits_dev, float *prob_dev, void *workspace_dev,
                               size_t workspace_bytes, void *stream) {
    return balf_forward_u8_status(packed_dev, precision, image_dev, channels, B, H, W, logits_dev, prob_dev, workspace_dev,
                                  workspace_bytes, nullptr, stream);
}
