// Host side of the weight path: the state-dict table the library expects and the packer that turns
// the reference's nn.Linear / LayerNorm / BatchNorm tensors into the MFMA-fragment-ordered blob of
// layout.h.  Replaces MLP_MA_DECODER.load_state_dict as reached from
// /root/reference/balf/model/get_model.py:60-67 (tensor names and shapes are that state_dict's).
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "layout.h"

namespace {

using namespace balf;

struct Entry {
    std::string name;
    size_t numel;
};

const std::vector<Entry> &table() {
    static const std::vector<Entry> t = [] {
        std::vector<Entry> v;
        const std::string rsh = "residual_split_head_multi_axis_gmlp_layer";
        const std::string rcab = "residual_channel_attention_block";
        for (int s = 0; s < kStages; ++s) {
            const size_t C = kC[s], Cin = kCin[s];
            const std::string d = "down" + std::to_string(s + 1) + ".";
            auto lin = [&](const std::string &n, size_t o, size_t i) {
                v.push_back({d + n + ".weight", o * i});
                v.push_back({d + n + ".bias", o});
            };
            auto ln = [&](const std::string &n) {
                v.push_back({d + n + ".weight", C});
                v.push_back({d + n + ".bias", C});
            };
            lin("conv.0", C, Cin);
            ln(rsh + ".norm");
            lin(rsh + ".dense1", 2 * C, C);
            const char *br[2] = {"grid_gmlp_layer", "block_gmlp_layer"};
            const char *un[2] = {"grid_gating_unit", "block_gating_unit"};
            for (int b = 0; b < 2; ++b) {
                const std::string p = rsh + "." + br[b];
                ln(p + ".norm");
                lin(p + ".dense1", 2 * C, C);
                ln(p + "." + un[b] + ".norm");
                lin(p + "." + un[b] + ".dense", kTokens, kTokens);
                lin(p + ".dense2", C, C);
            }
            lin(rsh + ".dense2", C, 2 * C);
            ln(rcab + ".norm");
            lin(rcab + ".conv1", C, C);
            lin(rcab + ".conv2", C, C);
            lin(rcab + ".calayer.excite.0", C / 4, C);
            lin(rcab + ".calayer.excite.2", C, C / 4);
            lin("conv2", C, C);
        }
        v.push_back({"detector_head.dense.weight", (size_t)kHeadN * kC[3]});
        v.push_back({"detector_head.dense.bias", (size_t)kHeadN});
        v.push_back({"detector_head.norm.weight", (size_t)kHeadN});
        v.push_back({"detector_head.norm.bias", (size_t)kHeadN});
        v.push_back({"detector_head.norm.running_mean", (size_t)kHeadN});
        v.push_back({"detector_head.norm.running_var", (size_t)kHeadN});
        return v;
    }();
    return t;
}

// W [N, K] row-major -> A-fragment order, output rows padded with zeros up to Npad.
void pack_frags(float *dst, const float *W, int N, int K, int Npad) {
    const int KT = K / 16;
    for (int nt = 0; nt < Npad / 16; ++nt)
        for (int kt = 0; kt < KT; ++kt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int n = 16 * nt + (lane & 15);
                    const int k = 16 * kt + 4 * (lane >> 4) + j;
                    dst[(((size_t)nt * KT + kt) * 64 + lane) * 4 + j] = n < N ? W[(size_t)n * K + k] : 0.0f;
                }
}

void copy(float *dst, const float *src, size_t n) { memcpy(dst, src, n * sizeof(float)); }

// Split-f16 fragments for v_mfma_f32_16x16x32_f16 (detector_f16.hip).  Weight tile (nt, ks) occupies 2 KiB:
// [hi: 64 lanes x 8 halves][lo: 64 lanes x 8 halves], hi = f16(w) (round to nearest even), lo = f16(w - hi).
// `permuted` = the K order of the register chain: k-slot (q, j) of K-step ks holds input channel
// 32 ks + 16 (j >> 2) + 4 q + (j & 3); natural order (k = 32 ks + 8 q + j) is used for the token-mix matrix,
// whose K axis is the token index read linearly from LDS.  The region has the same size as the fp32 one.
void pack_frags16(float *dst_region, const float *W, int N, int K, int Npad, bool permuted) {
    _Float16 *dst = reinterpret_cast<_Float16 *>(dst_region);
    const int KS = K / 32;
    for (int nt = 0; nt < Npad / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = 16 * nt + (lane & 15), q = lane >> 4;
                    const int k = permuted ? 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3) : 32 * ks + 8 * q + j;
                    const float w = n < N ? W[(size_t)n * K + k] : 0.0f;
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const size_t tile = ((size_t)nt * KS + ks) * 1024;          // halves per 2 KiB tile
                    dst[tile + lane * 8 + j] = hi;
                    dst[tile + 512 + lane * 8 + j] = lo;
                }
}

// Split-f16 fragments for v_mfma_f32_32x32x16_f16 (the 32x32 kernels: layout.h kFmt32).  Weight tile (R, s) = 32 output
// rows x 16 inputs occupies 2 KiB: [hi: 64 lanes x 8 halves][lo: ...]; lane (m = lane & 31, h = lane >> 5), half j holds
//   permuted:  W[row(R, m)][16 s + 8 (j >> 2) + 4 h + (j & 3)]  -- the K order of the register chain: an accumulator's
//              registers 8 s .. 8 s + 7 of lane half h (channels 8 g + 4 h + r, g = 2 s, 2 s + 1) are K-step s' K-slots;
//   natural:   W[row(R, m)][16 s + 8 h + j]                     -- the token-mix matrix (K = token index read linearly
//              from the transposed tile in LDS).
// row(R, m) = 32 R + m, or for the token-mix matrix of stage 1 (`token_rows`) 2 m + R: output tile R holds the tokens the
// lane columns of pixel tile R carry (token t = 2 n + p, stage1_f16.h; stage 2: token t = 32 R + n, stage2_f16.h).
void pack_frags32(float *dst_region, const float *W, int N, int K, int Npad, bool permuted, bool token_rows) {
    _Float16 *dst = reinterpret_cast<_Float16 *>(dst_region);
    const int KS = K / 16;
    for (int R = 0; R < Npad / 32; ++R)
        for (int s = 0; s < KS; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = lane & 31, h = lane >> 5;
                    const int n = token_rows ? 2 * m + R : 32 * R + m;
                    const int k = permuted ? 16 * s + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * s + 8 * h + j;
                    const float w = n < N ? W[(size_t)n * K + k] : 0.0f;
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const size_t tile = ((size_t)R * KS + s) * 1024;            // halves per 2 KiB tile
                    dst[tile + lane * 8 + j] = hi;
                    dst[tile + 512 + lane * 8 + j] = lo;
                }
}

// LayerNorm(x; gamma, beta) followed by Linear(W, b):  W (n * gamma + beta) + b = (W diag gamma) n + (W beta + b)
// where n = (x - mean) * rstd.  Folding the affine part into the Linear removes 3 VALU ops per element
// from the kernels (the f32 MFMA shares its pipe with the VALU, so they are not free).
void fold_ln(const float *W, const float *b, const float *gamma, const float *beta, int N, int K,
             std::vector<float> &Wf, std::vector<float> &bf) {
    Wf.resize((size_t)N * K);
    bf.resize(N);
    for (int n = 0; n < N; ++n) {
        double acc = b[n];
        for (int k = 0; k < K; ++k) {
            Wf[(size_t)n * K + k] = W[(size_t)n * K + k] * gamma[k];
            acc += (double)W[(size_t)n * K + k] * (double)beta[k];
        }
        bf[n] = (float)acc;
    }
}

}  // namespace

extern "C" int balf_num_state_tensors(void) { return (int)table().size(); }

extern "C" const char *balf_state_tensor_name(int i) {
    if (i < 0 || i >= (int)table().size()) return nullptr;
    return table()[i].name.c_str();
}

extern "C" size_t balf_state_tensor_numel(int i) {
    if (i < 0 || i >= (int)table().size()) return 0;
    return table()[i].numel;
}

extern "C" size_t balf_packed_weights_bytes(int precision) {
    if (precision != BALF_PREC_FP32 && precision != BALF_PREC_FP16) return 0;
    return (size_t)kLayout.total * sizeof(float);       // both blobs share layout.h's offsets
}

extern "C" int balf_pack_weights(const float *const *tensors, int n_tensors, int precision, void *packed_host,
                                 size_t packed_bytes) {
    if (!tensors || !packed_host) return BALF_ERR_ARG;
    if (n_tensors != kNumStateTensors || (int)table().size() != kNumStateTensors) return BALF_ERR_ARG;
    if (precision != BALF_PREC_FP32 && precision != BALF_PREC_FP16) return BALF_ERR_ARG;
    if (packed_bytes < balf_packed_weights_bytes(precision)) return BALF_ERR_WORKSPACE;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return BALF_ERR_ARG;

    float *blob = static_cast<float *>(packed_host);
    memset(blob, 0, (size_t)kLayout.total * sizeof(float));
    const bool f16 = precision == BALF_PREC_FP16;
    // MFMA-operand weights: fp32 A fragments, or split-f16 fragments in the same region
    bool fmt32 = false;                 // (set per stage below) split-f16 fragments of the 32x32x16 MFMA
    auto pack = [&](float *dst, const float *W, int N, int K, int Npad) {
        if (f16 && fmt32) pack_frags32(dst, W, N, K, Npad, true, false);
        else if (f16) pack_frags16(dst, W, N, K, Npad, true);
        else pack_frags(dst, W, N, K, Npad);
    };
    for (int s = 0; s < kStages; ++s) {
        const int C = kC[s], Cin = kCin[s];
        fmt32 = kFmt32[s];
        const StageOff &S = kLayout.st[s];
        const float *const *t = tensors + s * kTensorsPerStage;
        if (s == 0) copy(blob + S.conv0_w, t[0], (size_t)C * Cin);
        else pack(blob + S.conv0_w, t[0], C, Cin, C);
        copy(blob + S.conv0_b, t[1], C);
        std::vector<float> Wf, bf;
        copy(blob + S.qln_g, t[2], C);                 // kept for reference; the kernels use the folded form
        copy(blob + S.qln_b, t[3], C);
        fold_ln(t[4], t[5], t[2], t[3], 2 * C, C, Wf, bf);
        pack(blob + S.q1_w, Wf.data(), 2 * C, C, 2 * C);
        copy(blob + S.q1_b, bf.data(), 2 * C);
        for (int b = 0; b < 2; ++b) {
            const BranchOff &B = S.br[b];
            const float *const *u = t + 6 + 10 * b;
            copy(blob + B.ln_g, u[0], C);
            copy(blob + B.ln_b, u[1], C);
            fold_ln(u[2], u[3], u[0], u[1], 2 * C, C, Wf, bf);
            pack(blob + B.d1_w, Wf.data(), 2 * C, C, 2 * C);
            copy(blob + B.d1_b, bf.data(), 2 * C);
            copy(blob + B.gln_g, u[4], C);
            copy(blob + B.gln_b, u[5], C);
            if (f16 && fmt32) pack_frags32(blob + B.mix_w, u[6], kTokens, kTokens, kTokens, false, /*token_rows=*/s == 0);
            else if (f16) pack_frags16(blob + B.mix_w, u[6], kTokens, kTokens, kTokens, false);
            else pack_frags(blob + B.mix_w, u[6], kTokens, kTokens, kTokens);
            copy(blob + B.mix_b, u[7], kTokens);
            pack(blob + B.d2_w, u[8], C, C, C);
            copy(blob + B.d2_b, u[9], C);
        }
        pack(blob + S.q2_w, t[26], C, 2 * C, C);
        copy(blob + S.q2_b, t[27], C);
        copy(blob + S.rln_g, t[28], C);
        copy(blob + S.rln_b, t[29], C);
        fold_ln(t[30], t[31], t[28], t[29], C, C, Wf, bf);
        pack(blob + S.r1_w, Wf.data(), C, C, C);
        copy(blob + S.r1_b, bf.data(), C);
        pack(blob + S.r2_w, t[32], C, C, C);
        copy(blob + S.r2_b, t[33], C);
        copy(blob + S.r2_plain, t[32], C * C);
        copy(blob + S.se0_w, t[34], (size_t)(C / 4) * C);
        copy(blob + S.se0_b, t[35], C / 4);
        copy(blob + S.se2_w, t[36], (size_t)C * (C / 4));
        copy(blob + S.se2_b, t[37], C);
        pack(blob + S.conv2_w, t[38], C, C, C);
        copy(blob + S.conv2_b, t[39], C);
    }
    const float *const *h = tensors + kStages * kTensorsPerStage;
    fmt32 = kFmt32Head;
    pack(blob + kLayout.head_w, h[0], kHeadN, kC[3], kHeadNPad);
    copy(blob + kLayout.head_b, h[1], kHeadN);
    for (int c = 0; c < kHeadN; ++c) {
        // BatchNorm2d in eval mode (/root/reference/balf/model/decoder.py:12,22):
        // (x - mean) / sqrt(var + eps) * weight + bias  ==  x * alpha + beta
        const double alpha = (double)h[2][c] / sqrt((double)h[5][c] + (double)kBnEps);
        blob[kLayout.head_alpha + c] = (float)alpha;
        blob[kLayout.head_beta + c] = (float)((double)h[3][c] - (double)h[4][c] * alpha);
    }
    for (int i = 0; i < 256; ++i) blob[kLayout.u8_lut + i] = (float)((double)i / 255.0);
    // GELU chord table (nn.GELU() default = erf form, mlp_ma_decoder.py:52,99,126).  The kernels' index arithmetic
    // (stage1_f16.h: gelu_lut1) puts x in interval i when N (x + L) / 2L lies in [i - 1/16, i + 15/16); each chord is
    // shifted by half its largest deviation (equi-oscillating).  Entry 0 (x < -L + 15h/16) is exactly 0 and entry N
    // (x >= L - h/16) exactly x: |gelu - asymptote| < 7e-9 out there.
    {
        auto g = [](double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); };
        const double L = (double)kGeluLutL;
        auto uniform_table = [&](float *lut, int N) {
            const double hh = 2.0 * L / N;
            for (int i = 0; i <= N; ++i) {
                double a = 0.0, b = (i == N) ? 1.0 : 0.0;
                if (i > 0 && i < N) {
                    const double x0 = -L + (i - 1.0 / 16.0) * hh, x1 = x0 + hh, xm = 0.5 * (x0 + x1);
                    b = (g(x1) - g(x0)) / (x1 - x0);
                    a = g(x0) - b * x0;
                    a += 0.5 * (g(xm) - (a + b * xm));
                }
                lut[2 * i] = (float)a;
                lut[2 * i + 1] = (float)b;
            }
        };
        uniform_table(blob + kLayout.gelu_lut, kGeluLutN);
        uniform_table(blob + kLayout.gelu_lut2, kGeluLut2N);
        // E(a) = a erf(a / sqrt 2) / 2 over a = |x|: entry k * M + j covers a in [2^k (1 + j / M) - 1, + 2^k / M)
        auto E = [](double a) { return 0.5 * a * erf(a * 0.70710678118654752440); };
        float *lg = blob + kLayout.gelu_log;
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < kGeluLogM; ++j) {
                const double w = (double)(1 << k) / kGeluLogM, a0 = (double)(1 << k) * (1.0 + (double)j / kGeluLogM) - 1.0;
                const double a1 = a0 + w, am = 0.5 * (a0 + a1);
                const double b = (E(a1) - E(a0)) / w;
                double a = E(a0) - b * a0;
                a += 0.5 * (E(am) - (a + b * am));
                lg[2 * (k * kGeluLogM + j)] = (float)a;
                lg[2 * (k * kGeluLogM + j) + 1] = (float)b;
            }
        lg[2 * 3 * kGeluLogM] = 0.0f;                    // |x| >= 7: E = |x| / 2 (to 1e-11)
        lg[2 * 3 * kGeluLogM + 1] = 0.5f;
    }
    return BALF_OK;
}
