// Device helpers, argument structs and the workspace plan shared by the fp32 (detector.hip) and the
// split-f16 (detector_f16.hip) implementations of the detector forward.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "diag.h"
#include "layout.h"
#include "prof.h"

namespace balf {


typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 ldg4(const float *p) { return *reinterpret_cast<const f4 *>(p); }

// Tuning: pixel tiles (of 16) per wave and the register-allocation target (waves per SIMD) per stage.
#ifndef BALF_P32
#define BALF_P32 4
#endif
#ifndef BALF_P64
#define BALF_P64 2
#endif
#ifndef BALF_OCC32
#define BALF_OCC32 2
#endif
#ifndef BALF_OCC64
#define BALF_OCC64 2
#endif
#ifndef BALF_OCC128
#define BALF_OCC128 2
#endif
template <int C> struct StageP;               // pixel tiles (of 16) per wave
template <> struct StageP<32> { static constexpr int P = BALF_P32, OCC = BALF_OCC32; };
template <> struct StageP<64> { static constexpr int P = BALF_P64, OCC = BALF_OCC64; };
template <> struct StageP<128> { static constexpr int P = 1, OCC = BALF_OCC128; };
template <> struct StageP<256> { static constexpr int P = 1, OCC = 1; };

constexpr int kBtPitch = kTokens + 4;          // floats per channel row of the transposed token tile

// ------------------------------------------------------------------------------------------------
// register-tile helpers: a [C]-channel activation of 16*P pixels is f4 t[NT][P], NT = C/16
// ------------------------------------------------------------------------------------------------
template <int NT, int P>
__device__ __forceinline__ void init_bias(f4 (&t)[NT][P], const float *bias, int q) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f4 b = ldg4(bias + 16 * nt + 4 * q);
#pragma unroll
        for (int p = 0; p < P; ++p) t[nt][p] = b;
    }
}

// max(x, 0) as ONE instruction: a signed-integer max on the float's bits (negative floats, -0 included, are negative
// integers).  fmaxf on an MFMA result costs a second v_max_f32 that only quiets NaNs (IEEE mode), and hipcc folds
// v_med3_f32 with constants back into that pair.
__device__ __forceinline__ float max0(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// Exact (erf) GELU, nn.GELU() default (mlp_ma_decoder.py:52,99,126): x * Phi(x) = max(x, 0) - |x| Phi(-|x|), with
// Phi(-a) = 2^P(a), P = degree-5 polynomial fitted to log2(0.5 erfc(a / sqrt 2)) under the weight a Phi(-a) (the factor
// its error is multiplied by).  Max-abs error of the whole expression vs fp64 over [-12, 12]: 8.6e-7 (torch's own fp32
// GELU: 1.2e-6); the leading coefficient is negative, so large |x| underflow to the exact limits (0 and x).
// 8 VALU instructions, ONE transcendental (v_exp_f32 issues at a quarter of the fma rate) -- the Abramowitz-Stegun
// 7.1.26 form used before (BALF_GELU_AS=1) took 14 with two (v_rcp_f32, v_exp_f32): GELU was 45 % of the vector
// instructions of the stage-1 kernels.
#ifndef BALF_GELU_AS
#define BALF_GELU_AS 0
#endif
constexpr float kG0 = -1.000037633e+00f, kG1 = -1.150787766e+00f, kG2 = -4.599926517e-01f, kG3 = -5.182716455e-02f,
                kG4 = 7.084460191e-03f, kG5 = -4.732939498e-04f;
template <bool AS = (BALF_GELU_AS != 0)>
__device__ __forceinline__ float gelu1(float x) {
    const float ax = fabsf(x);
    if constexpr (AS) {
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
        const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
        float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);     // coefficients pre-scaled by 1/2
        p = fmaf(p, t, 0.5f * 1.421413741f);
        p = fmaf(p, t, 0.5f * -0.284496736f);
        p = fmaf(p, t, 0.5f * 0.254829592f);
        const float y = p * t * e;                     // Phi(-|x|)
        return fmaf(ax, 0.5f - y, 0.5f * x);           // = max(x, 0) - |x| y:  x >= 0: x(1 - y);  x < 0: x y
    } else {
        float p = fmaf(kG5, ax, kG4);
        p = fmaf(p, ax, kG3);
        p = fmaf(p, ax, kG2);
        p = fmaf(p, ax, kG1);
        p = fmaf(p, ax, kG0);
        float e = __builtin_amdgcn_exp2f(p);
        asm("" : "+v"(e));       // opaque: keeps hipcc from pairing two of these fmas into a v_pk_fma_f32, which has no
                                 // |x| modifier and costs two extra v_or_b32 per pair
        return fmaf(-ax, e, max0(x));
    }
}

// GELU on two values with the multiply/add work written as 2-wide vector math, which hipcc lowers to
// v_pk_mul_f32 / v_pk_fma_f32 (two fp32 lanes per VALU slot).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 gelu2(f2 x) {
    const f2 ax = {fabsf(x[0]), fabsf(x[1])};
#if BALF_GELU_AS
    const f2 den = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
    const f2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f2 ea = x * x * (-0.5f * 1.44269504088896340736f);
    const f2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
    f2 p = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    p = p * t + (0.5f * 1.421413741f);
    p = p * t + (0.5f * -0.284496736f);
    p = p * t + (0.5f * 0.254829592f);
    const f2 y = p * t * e;                         // Phi(-|x|)
    return ax * (0.5f - y) + x * 0.5f;
#else
    f2 p = ax * kG5 + kG4;
    p = p * ax + kG3;
    p = p * ax + kG2;
    p = p * ax + kG1;
    p = p * ax + kG0;
    const f2 e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    const f2 m = {max0(x[0]), max0(x[1])};
    return m - ax * e;
#endif
}

// PACKED: 2-wide vector math (v_pk_*): fewer VALU slots, but the register pairs it needs cost more than they
// save in the register-starved kernels (block branch, C = 256) -- measured per kernel, see DESIGN.md.
// AS: the longer Abramowitz-Stegun form -- kept for the 128-register N-split kernel of stage 3, which spills 35 registers
// with the polynomial form (more evaluations in flight) and is latency-bound, not issue-bound.
template <bool PACKED = true, bool AS = (BALF_GELU_AS != 0), int NT, int P>
__device__ __forceinline__ void gelu(f4 (&t)[NT][P]) {
    if (BALF_ABLATE_GELU) return;                // timing experiment only
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (PACKED) {
                const f2 lo = gelu2(f2{t[nt][p][0], t[nt][p][1]}), hi = gelu2(f2{t[nt][p][2], t[nt][p][3]});
                t[nt][p] = f4{lo[0], lo[1], hi[0], hi[1]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) t[nt][p][r] = gelu1<AS>(t[nt][p][r]);
            }
        }
}

// mask ? a : b per lane with mask = all ones / all zeros: one v_bfi_b32 (v_cndmask_b32 on a VCC mask measured 23 cycles)
__device__ __forceinline__ float lane_select(unsigned mask, float a, float b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    return __builtin_bit_cast(float, (ua & mask) | (ub & ~mask));
}

template <int NT, int P>
__device__ __forceinline__ void relu(f4 (&t)[NT][P]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[nt][p][r] = max0(t[nt][p][r]);
}

template <int NT, int P>
__device__ __forceinline__ void lrelu(f4 (&t)[NT][P]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = t[nt][p][r];
                // = v > 0 ? v : 0.2 v (slope < 1) in two instructions: the compiler's fmaxf first canonicalises v
                // (a third instruction); m is an arithmetic result and v has been read by the multiply before.  The
                // result is written into v's OWN register ("+v"), never into a fresh one: hipcc does not pad hazards
                // around inline asm, and a free register can be the SrcC of an MFMA (of an independent Linear the
                // scheduler hoisted) still in flight -- see split16.h
                const float m = 0.2f * v;
                asm("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(m));
                t[nt][p][r] = v;
            }
}

__device__ __forceinline__ float quarter_allreduce(float v) {   // the 4 lanes l, l^16, l^32, l^48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// LayerNorm over the channel axis (eps 1e-5, affine), y may alias x.  4-wide vector math (-> packed VALU).
template <bool PACKED = true, int NT, int P>
__device__ __forceinline__ void ln_stats(const f4 (&x)[NT][P], int p, float &mean, float &rstd) {
    constexpr float inv_c = 1.0f / (16 * NT);
    if constexpr (!PACKED) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) s += (x[nt][p][0] + x[nt][p][1]) + (x[nt][p][2] + x[nt][p][3]);
        mean = quarter_allreduce(s) * inv_c;
        float v = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = x[nt][p][r] - mean;
                v = fmaf(d, d, v);
            }
        rstd = __builtin_amdgcn_rsqf(quarter_allreduce(v) * inv_c + kLnEps);
        return;
    }
    f4 s4 = x[0][p];
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) s4 += x[nt][p];
    mean = quarter_allreduce((s4[0] + s4[1]) + (s4[2] + s4[3])) * inv_c;
    f4 v4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f4 d = x[nt][p] - mean;
        v4 += d * d;
    }
    rstd = __builtin_amdgcn_rsqf(quarter_allreduce((v4[0] + v4[1]) + (v4[2] + v4[3])) * inv_c + kLnEps);
}

template <bool PACKED = true, int NT, int P>
__device__ __forceinline__ void layernorm(const f4 (&x)[NT][P], f4 (&y)[NT][P], const float *g, const float *b,
                                          int q) {
    float mean[P], rstd[P];
#pragma unroll
    for (int p = 0; p < P; ++p) ln_stats<PACKED>(x, p, mean[p], rstd[p]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f4 gg = ldg4(g + 16 * nt + 4 * q), bb = ldg4(b + 16 * nt + 4 * q);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if constexpr (PACKED) y[nt][p] = (x[nt][p] - mean[p]) * (gg * rstd[p]) + bb;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) y[nt][p][r] = (x[nt][p][r] - mean[p]) * rstd[p] * gg[r] + bb[r];
        }
    }
}

// LayerNorm without the affine part: (x - mean) * rstd.  Used where gamma/beta were folded into the
// weights/bias of the Linear that consumes the result (weights.hip: fold_ln).
template <bool PACKED = true, int NT, int P>
__device__ __forceinline__ void layernorm_plain(const f4 (&x)[NT][P], f4 (&y)[NT][P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float mean, rstd;
        ln_stats<PACKED>(x, p, mean, rstd);
        const float shift = -mean * rstd;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (PACKED) y[nt][p] = x[nt][p] * rstd + shift;
            else
#pragma unroll
                for (int r = 0; r < 4; ++r) y[nt][p][r] = fmaf(x[nt][p][r], rstd, shift);
        }
    }
}

template <int NT, int P>
__device__ __forceinline__ void store_slot(f4 *slot, const f4 (&t)[NT][P], int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) slot[(nt * P + p) * 64 + lane] = t[nt][p];
}

// ------------------------------------------------------------------------------------------------
struct StageArgs {
    const float *blob;
    StageOff off;
    const float *X;       // stage input: NCHW [B,3,H,W] (stage 1) or NHWC [B,H,W,CIN]
    // stage 1 only: raw uint8 image instead of X (u8_ch = 1 gray [B,sh,sw] or 3 RGB [B,sh,sw,3]; 0 = use X).
    // Pixel (y, x) of the padded frame reads image pixel (y - u8_top, x - u8_left), zero outside, through the
    // float32(i / 255.0) table -- make_shape_even + mod_padding_symmetric + /255 of the callers
    // (/root/reference/balf/utils/test_utils.py:16-32, /root/reference/demo/demo_match.py:22) on the fly.
    const unsigned char *X8;
    int u8_ch, u8_h, u8_w, u8_top, u8_left;
    int B, H, W;          // resolution of this stage
    float *U;             // [B,H,W,C] grid-branch output u'
    float *T;             // [B,H,W,C] RCAB body output t
    float *R;             // [B,H,W,C] x1 + x0
    float *partial;       // [B, wgs_per_image, C] channel sums of t (stage 1, fused tail: of the RCAB's hidden layer)
    const float *scale;   // stage-1 tail kernel only: [B, C] squeeze-excite scale
    float *out;           // stage-1 tail kernel only: next stage's input, fragment format [B, H/2, W/2, C]
    int *status;          // tail kernels: optional status block (include/balf_hip.h: balf_forward_status), may be null
};

// Status words (include/balf_hip.h): plain stores of 1, system scope (the block may be pinned host memory); no atomics, the
// library never clears a word.
__device__ __forceinline__ void status_raise(int *status, int word) {
    if (status) __hip_atomic_store(status + word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// running maximum of |v| over the values a kernel is about to split into f16 halves (NaN is ignored here: it reaches the
// head kernel's check): two values per v_max3_f32
__device__ __forceinline__ float range_max(float m, float a, float b) {
    return __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
}
constexpr float kF16Max = 65504.0f;

struct InputU8 {           // optional raw-image input of the forward (ch = 0: none)
    const unsigned char *p;
    int ch, h, w, top, left;
};

// stage-1 input of padded-frame pixel (n, y, x): the three network input channels
__device__ __forceinline__ void load_input3(const StageArgs &A, const float *lut, int n, int y, int x, float (&v)[3]) {
    if (A.u8_ch == 0) {
        const long hw = (long)A.H * A.W;
        const long o = (long)y * A.W + x;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = A.X[((long)n * 3 + k) * hw + o];
    } else {
        const int yy = y - A.u8_top, xx = x - A.u8_left;
        const bool in = yy >= 0 && yy < A.u8_h && xx >= 0 && xx < A.u8_w;
        const unsigned char *px = A.X8 + (((long)n * A.u8_h + (in ? yy : 0)) * A.u8_w + (in ? xx : 0)) * A.u8_ch;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = in ? lut[px[A.u8_ch == 3 ? k : 0]] : 0.0f;
    }
}

template <int C, int P>
constexpr int stage_lds_bytes() {
    constexpr int slots = 4 * (C / 16) * P * 1024;
    constexpr int bt = P * C * kBtPitch * 4;
    return (slots > bt ? slots : bt) + 4 * C * 4;
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite: s[n, :] = sigmoid(W2 relu(W0 mean_hw(t) + b0) + b2)   (mlp_ma_decoder.py:166-171)
// ------------------------------------------------------------------------------------------------
// Level 1: [B, per_img, C] partial sums -> [B, kSeChunks, C] chunk sums (fixed order: deterministic).  A thread owns four
// adjacent channels (16-byte loads); the 256 / (C/4) row groups of a block stride over the chunk's rows.
constexpr int kSeChunks = 128;

template <int C>
__global__ __launch_bounds__(256) void se_reduce_kernel(const float *__restrict__ partial, int per_img,
                                                        float *__restrict__ chunk) {
    constexpr int LANES = C / 4;                       // threads per row
    constexpr int PARTS = 256 / LANES;                 // rows in flight per block (>= 4 for C <= 256)
    __shared__ f4 s_part[PARTS][LANES];
    const int n = blockIdx.x / kSeChunks, ch = blockIdx.x % kSeChunks;
    const int per_chunk = (per_img + kSeChunks - 1) / kSeChunks;
    const int i0 = ch * per_chunk, i1 = (i0 + per_chunk < per_img) ? i0 + per_chunk : per_img;
    const float *pp = partial + (long)n * per_img * C;
    const int c4 = threadIdx.x % LANES, part = threadIdx.x / LANES;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = i0 + part; i < i1; i += PARTS) acc += *reinterpret_cast<const f4 *>(pp + (long)i * C + 4 * c4);
    s_part[part][c4] = acc;
    __syncthreads();
    if (threadIdx.x < LANES) {
        f4 t = s_part[0][threadIdx.x];
        for (int k = 1; k < PARTS; ++k) t += s_part[k][threadIdx.x];
        *reinterpret_cast<f4 *>(chunk + ((long)n * kSeChunks + ch) * C + 4 * threadIdx.x) = t;
    }
}

// hidden_sums != 0: the chunk sums are those of the RCAB's hidden layer h = lrelu(conv1(.)) (stage 1 with the fused tail
// kernel, stage1_f16.h); conv2 is linear, so mean(t) = conv2_w mean(h) + conv2_b.
template <int C>
__global__ __launch_bounds__(256) void se_kernel(const float *blob, StageOff S, const float *chunk, float inv_hw,
                                                 float *scale, int hidden_sums, int *status) {
    __shared__ float s_mean[C];
    __shared__ float s_in[C];
    __shared__ float s_hid[C / 4];
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.0f;
        for (int k = 0; k < kSeChunks; ++k) acc += chunk[((long)n * kSeChunks + k) * C + c];
        (hidden_sums ? s_in : s_mean)[c] = acc * inv_hw;
    }
    __syncthreads();
    if (hidden_sums) {
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = blob[S.r2_b + c];
            for (int k = 0; k < C; ++k) acc += blob[S.r2_plain + c * C + k] * s_in[k];
            s_mean[c] = acc;
        }
        __syncthreads();
    }
    for (int h = threadIdx.x; h < C / 4; h += 256) {
        float acc = blob[S.se0_b + h];
        for (int c = 0; c < C; ++c) acc += blob[S.se0_w + h * C + c] * s_mean[c];
        s_hid[h] = fmaxf(acc, 0.0f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = blob[S.se2_b + c];
        for (int h = 0; h < C / 4; ++h) acc += blob[S.se2_w + c * (C / 4) + h] * s_hid[h];
        if (!(fabsf(acc) < INFINITY)) status_raise(status, 2 /* BALF_STATUS_SE */);
        scale[(long)n * C + c] = 1.0f / (1.0f + expf(-acc));
    }
}

// ------------------------------------------------------------------------------------------------
// stage-4 tail + detector head: one wave = 16 pixels of the 1/8-resolution map
// ------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float *blob;
    StageOff off;            // stage 4 (conv2)
    int head_w, head_b, head_alpha, head_beta;
    const float *T, *R, *scale;   // [B,h,w,256], [B,256]
    int B, h, w;             // 1/8 resolution
    float *logits;           // [B,65,h,w] or nullptr
    float *prob;             // [B,8h,8w]
    int *status;             // optional status block (include/balf_hip.h: balf_forward_status), may be null
};

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#ifndef BALF_MB_PIXELS
// padded input pixels per micro-batch: 16 images at 1088x1920 (14.8 GB of workspace).  Round 4: 8 -> 16 images per launch,
// +1 % on 32 x 1080p (the persistent kernels stage ~100 KB of weights per CU at every launch and end on a partial round of
// groups; 32 per launch buys nothing more).  balf_forward_micro_batch() reports it.
#define BALF_MB_PIXELS (32L * 1024 * 1024)
#endif
struct Plan {
    int mb;                 // images per micro-batch
    size_t off_U, off_T, off_R, off_X[3], off_partial, off_chunk, off_scale, total;
};

// Run-time knob: BALF_MB_MPIXELS = 1 .. 64 in the environment of the process overrides the compiled default, in
// units of 2^20 padded pixels (16: half the workspace -- 7.4 GB instead of 14.8 GB at 1088x1920 -- for ~1 % of the throughput).
// Read once: balf_forward_workspace_bytes, balf_forward_micro_batch and balf_forward must agree within a process.
inline long mb_pixels() {
    static const long v = [] {
        const char *e = getenv("BALF_MB_MPIXELS");
        const long m = e ? atol(e) : 0;
        return (m >= 1 && m <= 64) ? (m << 20) : (long)BALF_MB_PIXELS;
    }();
    return v;
}

inline Plan make_plan(int B, int Hp, int Wp) {
    Plan p{};
    const long px = (long)Hp * Wp;
    long mb = mb_pixels() / px;                        // <= 33.5 Mpx of stage-1 activations in flight (default)
    if (mb < 1) mb = 1;
    if (mb > B) mb = B;
    p.mb = (int)mb;
    size_t o = 0;
    auto take = [&](size_t floats) { size_t r = o; o = balf_align_up(o + floats * sizeof(float), 256); return r; };
    const size_t big = (size_t)mb * px * 32;           // every stage: H*W*C = px * 32 / 2^(s)
    p.off_U = take(big);
    p.off_T = take(big);
    p.off_R = take(big);
    p.off_X[0] = take((size_t)mb * (px / 4) * 32);
    p.off_X[1] = take((size_t)mb * (px / 16) * 64);
    p.off_X[2] = take((size_t)mb * (px / 64) * 128);
    p.off_partial = take((size_t)mb * (px / 64) * 32);  // groups/P * C <= px/64 * 32 for every stage
    p.off_chunk = take((size_t)mb * kSeChunks * 256);
    p.off_scale = take((size_t)mb * 256);
    p.total = o;
    return p;
}


// entry points of the two implementations (detector.hip / detector_f16.hip)
int forward_f32(const float *blob, const float *x, const InputU8 &u8, int B, int Hp, int Wp, float *logits, float *prob,
                char *ws, const Plan &pl, int *status, hipStream_t st);
int forward_f16(const float *blob, const float *x, const InputU8 &u8, int B, int Hp, int Wp, float *logits, float *prob,
                char *ws, const Plan &pl, int *status, hipStream_t st);

}  // namespace balf
