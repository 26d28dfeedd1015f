// Packed-weight blob layout shared by the host packer (weights.hip) and the kernels (detector.hip).
//
// All offsets are in floats from the start of the blob and are multiples of 64 floats (256 B).
// Linear weights that feed the MFMA are stored in "A-fragment order" for v_mfma_f32_16x16x4_f32:
//   frag(nt, kt)[lane][j] = W[16*nt + (lane & 15)][16*kt + 4*(lane >> 4) + j]      j = 0..3
// laid out as float4[(nt * K/16 + kt) * 64 + lane], so one wave-wide 16-byte load fetches a whole
// 16(out) x 16(in) weight tile as 1 KiB of contiguous memory, and element j feeds MFMA k-step j
// (the k-slot `lane >> 4` of step j carries input channel 16*kt + 4*(lane>>4) + j, which is exactly
// the register the accumulator layout of the previous MFMA left that channel in).
#pragma once

namespace balf {

constexpr int kStages = 4;
constexpr int kC[kStages] = {32, 64, 128, 256};
constexpr int kCin[kStages] = {3, 32, 64, 128};
constexpr int kTokens = 64;       // 8x8 grid cells / 8x8 block positions
constexpr int kHeadN = 65;        // 64 cell positions + dustbin
constexpr int kHeadNPad = 80;     // padded to a multiple of 16 output rows
constexpr float kLnEps = 1e-5f;
constexpr float kBnEps = 1e-5f;
// GELU table of the stage-1 split-f16 kernels (stage1_f16.h): kGeluLutN intervals over [-kGeluLutL, kGeluLutL), one
// (a, b) pair per interval with gelu(x) ~ a + b x, plus the two exact asymptotes (entry 0: 0, entry N: x)
constexpr int kGeluLutN = 3072;
constexpr float kGeluLutL = 6.0f;
// the same table with 2048 intervals (16 KB instead of 24): the stage-2 block kernel (stage2_f16.h), whose LDS also holds
// 104 KB of weights and the token tiles; chord error 1.7e-6 instead of 7.6e-7
constexpr int kGeluLut2N = 2048;
// GELU table of the channel-split kernels (stage_cs_f16.h), which have 6 KB of LDS to spare, not 24: gelu(x) = x / 2 +
// E(|x|), E(a) = a erf(a / sqrt 2) / 2, chords of E over intervals whose width doubles where the curvature allows it:
// kGeluLogM intervals each over |x| in [0, 1), [1, 3), [3, 7) -- the binades of (|x| + 1) / 8, so the interval index is
// a bit field of that float -- plus the asymptote entry E = |x| / 2 for |x| >= 7.  Same chord error as the table above.
constexpr int kGeluLogM = 256;
constexpr int kGeluLogEntries = 3 * kGeluLogM + 1;

// Which stages' split-f16 weights are packed as fragments of v_mfma_f32_32x32x16_f16 (weights.hip: pack_frags32) instead
// of v_mfma_f32_16x16x32_f16 (pack_frags16); the fp32 blob is not affected.  Stage s's OUTPUT activation (the next stage's
// input X, fragment format in HBM) follows the format of the stage that CONSUMES it.
constexpr bool kFmt32[kStages] = {true, true, false, false};
constexpr bool kFmt32Head = false;

struct BranchOff {                // GridGmlpLayer / BlockGmlpLayer
    int ln_g, ln_b;               // .norm
    int d1_w, d1_b;               // .dense1  [2C, C]   frags
    int gln_g, gln_b;             // .{grid,block}_gating_unit.norm
    int mix_w, mix_b;             // .{grid,block}_gating_unit.dense [64, 64] frags
    int d2_w, d2_b;               // .dense2  [C, C]    frags
};

struct StageOff {
    int conv0_w, conv0_b;         // .conv.0 [C, Cin]: plain row-major for stage 1 (Cin = 3), frags otherwise
    int qln_g, qln_b;             // RSHMAG .norm
    int q1_w, q1_b;               // RSHMAG .dense1 [2C, C] frags
    BranchOff br[2];              // 0 = grid, 1 = block
    int q2_w, q2_b;               // RSHMAG .dense2 [C, 2C] frags
    int rln_g, rln_b;             // RCAB .norm
    int r1_w, r1_b, r2_w, r2_b;   // RCAB .conv1 / .conv2 [C, C] frags
    int se0_w, se0_b;             // calayer.excite.0 [C/4, C] plain
    int se2_w, se2_b;             // calayer.excite.2 [C, C/4] plain
    int conv2_w, conv2_b;         // .conv2 [C, C] frags (stage 4 only; dead weight elsewhere)
    int r2_plain;                 // RCAB .conv2 again, plain row-major fp32 [C, C]: mean(conv2(h)) = conv2(mean(h)) in the SE kernel
};

struct Layout {
    StageOff st[kStages];
    int head_w;                   // detector_head.dense [80(pad), 256] frags
    int head_b;                   // [80] dense bias
    int head_alpha, head_beta;    // [80] BatchNorm(eval) as z = lin * alpha + beta
    int u8_lut;                   // [256] float32(i / 255.0): uint8 image -> network input (demo_match.py:22)
    int gelu_lut;                 // [kGeluLutN + 1][2] chord table of the exact GELU (see kGeluLutN)
    int gelu_log;                 // [kGeluLogEntries][2] chord table of E(|x|) (see kGeluLogM)
    int gelu_lut2;                // [kGeluLut2N + 1][2] chord table of the exact GELU, coarser (see kGeluLut2N)
    int total;                    // floats
};

constexpr int align64(int v) { return (v + 63) / 64 * 64; }

constexpr Layout make_layout() {
    Layout L{};
    int o = 0;
    auto take = [&](int n) { int r = o; o = align64(o + n); return r; };
    for (int s = 0; s < kStages; ++s) {
        const int C = kC[s], Cin = kCin[s];
        StageOff &S = L.st[s];
        S.conv0_w = take(C * Cin); S.conv0_b = take(C);
        S.qln_g = take(C); S.qln_b = take(C);
        S.q1_w = take(2 * C * C); S.q1_b = take(2 * C);
        for (int b = 0; b < 2; ++b) {
            BranchOff &B = S.br[b];
            B.ln_g = take(C); B.ln_b = take(C);
            B.d1_w = take(2 * C * C); B.d1_b = take(2 * C);
            B.gln_g = take(C); B.gln_b = take(C);
            B.mix_w = take(kTokens * kTokens); B.mix_b = take(kTokens);
            B.d2_w = take(C * C); B.d2_b = take(C);
        }
        S.q2_w = take(2 * C * C); S.q2_b = take(C);
        S.rln_g = take(C); S.rln_b = take(C);
        S.r1_w = take(C * C); S.r1_b = take(C);
        S.r2_w = take(C * C); S.r2_b = take(C);
        S.se0_w = take(C / 4 * C); S.se0_b = take(C / 4);
        S.se2_w = take(C * (C / 4)); S.se2_b = take(C);
        S.conv2_w = take(C * C); S.conv2_b = take(C);
        S.r2_plain = take(C * C);
    }
    L.head_w = take(kHeadNPad * kC[3]);
    L.head_b = take(kHeadNPad);
    L.head_alpha = take(kHeadNPad);
    L.head_beta = take(kHeadNPad);
    L.u8_lut = take(256);
    L.gelu_lut = take(2 * (kGeluLutN + 1));
    L.gelu_log = take(2 * kGeluLogEntries);
    L.gelu_lut2 = take(2 * (kGeluLut2N + 1));
    L.total = o;
    return L;
}

constexpr Layout kLayout = make_layout();

// Number of floating-point state tensors: 40 per stage + 6 head entries.
constexpr int kTensorsPerStage = 40;
constexpr int kNumStateTensors = kStages * kTensorsPerStage + 6;

}  // namespace balf
