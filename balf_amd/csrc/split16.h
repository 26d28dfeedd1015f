// Split-f16 operand helpers shared by the f16-MFMA kernels (detector_f16.hip, hardnet.hip).
//
// A value v is carried as two halves, hi = f16(v) and lo = f16(v - hi); a product is accumulated in fp32 as
// hi*hi' + lo*hi' + hi*lo' (three v_mfma_f32_16x16x32_f16), ~2^-20 relative operand error.
//
// OPERAND RANGE.  hi is converted round-toward-zero and saturates at 65504; once |v| exceeds ~1.3e5 the residual
// v - hi no longer fits f16 either and lo becomes +-inf, which the three products turn into inf/NaN without any
// trap.  Below 6.1e-5 the halves go subnormal and precision degrades gracefully (2^-24 absolute).  Operands that
// pass through a LayerNorm are O(1); the un-normalised ones (stage inputs after max-pool, the gated a*(mix+1), the
// RCAB hidden layer after leaky-relu, the head input, the weights themselves) scale with the checkpoint.  The host
// side offers MLP_MA_DECODER.validate_fp16() to check a checkpoint against the fp32 path (tests/test_forward_gpu.py
// exercises activations up to ~1e4 and the out-of-range failure).
#pragma once
#include "common.h"
#include "diag.h"

namespace balf {

typedef float f4x __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

struct HL {
    h8 hi, lo;
};

__device__ __forceinline__ f4x mfma16(h8 a, h8 b, f4x c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// three-product accumulate: (ah + al)(bh + bl) ~ ah bh + al bh + ah bl
__device__ __forceinline__ f4x mfma16x3(const HL &a, const HL &b, f4x c) {
    c = mfma16(a.lo, b.hi, c);
    c = mfma16(a.hi, b.lo, c);
    return mfma16(a.hi, b.hi, c);
}

#ifndef BALF_SPLIT_MIX
#define BALF_SPLIT_MIX 1
#endif
template <int MIX = BALF_SPLIT_MIX>
__device__ __forceinline__ void split_pair(float v0, float v1, h2 &hi, h2 &lo) {
    if (BALF_ABLATE_SPLIT) {                       // timing experiment: one convert, no residual
        typedef __fp16 fp16x2_ __attribute__((ext_vector_type(2)));
        const fp16x2_ h_ = __builtin_amdgcn_cvt_pkrtz(v0, v1);
        hi = __builtin_bit_cast(h2, h_); lo = hi;
        return;
    }
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v0, v1);      // hi = rtz_f16(v); v - hi is exact in fp32
    hi = __builtin_bit_cast(h2, h);
    if constexpr (MIX == 2) {
    // MIX == 2 (the stage-1 kernels): residuals in fp32 (v_fma_mix_f32 reads hi as f16: 4.9 cycles each against 8.9 for the
    // f16-writing forms below), then one more packed convert -- lo is rounded toward zero instead of to nearest (2^-21
    // instead of 2^-22 of v); each residual is built in the register of its own input (see the hazard note below).
    // Measured per 8 x 1088x1920: stage-1 block 1.465 -> 1.42 ms, grid 1.12 -> 1.07; the channel-split kernels of stages 2-4
    // (at their register limits) lose 1 % with it and keep MIX == 1.
    const unsigned hu2 = __builtin_bit_cast(unsigned, h);
    float r0 = v0, r1 = v1;
    asm("v_fma_mix_f32 %0, %2, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(r0), "+v"(r1) : "v"(hu2));
    lo = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    } else if constexpr (MIX == 1) {
    // lo = f16(v - hi) straight from the packed halves: v_fma_mix{lo,hi}_f16 read hi as f16, v as f32, and write
    // one half of the destination each -- 3 instructions per pair instead of the 5 hipcc emits for the casts.
    // The result is built IN v0's REGISTER ("+v"): hipcc does not pad hazards around inline asm, and a free register
    // picked for a separate output can be the SrcC (bias) of an MFMA issued a few cycles earlier -- a VALU write within
    // 7 wait states of such an MFMA corrupts its accumulator input (measured: non-deterministic 1e-4 errors).  v0's
    // register holds a live VALU result up to this point, so no MFMA in flight can be reading it.
    const unsigned hu = __builtin_bit_cast(unsigned, h);
    unsigned lu = __builtin_bit_cast(unsigned, v0);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(lu) : "v"(hu), "v"(v1));
    lo = __builtin_bit_cast(h2, lu);
    } else {
    const fp16x2 l = __builtin_amdgcn_cvt_pkrtz(fmaf((float)hi[0], -1.0f, v0), fmaf((float)hi[1], -1.0f, v1));
    lo = __builtin_bit_cast(h2, l);
    }
}


}  // namespace balf
