// HardNet patch descriptor on gfx950 (the descriptor of the demo path).
//
// Reference: /root/reference/third_party/hardnet/hardnet_pytorch.py:31-72 -- per-patch (mean, unbiased std)
// normalisation of a 32x32 patch, six [conv3x3 -> BatchNorm(eval, affine=False) -> ReLU] layers
// (1->32, 32->32, 32->64 /2, 64->64, 64->128 /2, 128->128), a valid 8x8 convolution 128->128 + BatchNorm, and
// x / sqrt(sum x^2 + 1e-10).  Called per 1000 patches from demo/demo_match.py:72-93.
//
// Every convolution is an implicit GEMM  D[co][pixel] = sum_{tap,ci} W[co][tap,ci] * act[ci][pixel + tap]  on
// v_mfma_f32_16x16x32_f16 with split operands (split16.h): activations live in HBM and LDS as two f16 planes
// (hi, lo), channel-fastest ([y][x][c]), so that the B operand of a K-step (32 input channels of one tap) is a
// single 16-byte LDS read per plane, shifted per tap by an address offset only -- no im2col buffer.  The LDS
// image of the input band carries a one-pixel zero frame (the convolution padding) and XOR-swizzles the 16-byte
// chunk index with the pixel index, which makes the 16 pixels of an MFMA column tile hit 16 different bank
// groups without padding bytes.  BatchNorm is folded into the packed weights (scale) and a bias on the host.
//
//   hn_conv_kernel<L2>   patch normalisation + conv1 (K = 9 padded to one K-step) computed into LDS for the band,
//                        then conv2; a1 never exists in HBM
//   hn_conv_kernel<L3..L6>
//   hn_fc_kernel         the 8x8 valid convolution as a [128 x 8192] x [8192 x patches] GEMM, B operand straight
//                        from HBM, fused BatchNorm bias + L2 normalisation
//
// A workgroup is 4 waves; a wave owns 2 output-channel tiles x 4 pixel tiles (8 accumulator tiles).  Weights are
// read from global memory (L2-resident, <= 576 KiB per layer) in A-fragment order, one K-step ahead.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "prof.h"
#include "split16.h"

namespace balf {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kPS = 32;                       // patch side
constexpr int kDesc = 128;
constexpr int kFcK = 8 * 8 * 128;

// ---- packed blob ---------------------------------------------------------------------------------
// weights of layer l: [K-step][channel tile][plane hi|lo][lane][8 halves]; k = (ky*KW + kx)*CIN + ci.
struct HnConv { int cin, cout, ksz; };
constexpr HnConv kConv[7] = {{1, 32, 3}, {32, 32, 3}, {32, 64, 3}, {64, 64, 3}, {64, 128, 3}, {128, 128, 3}, {128, 128, 8}};
constexpr int kFeatIdx[7] = {0, 3, 6, 9, 12, 15, 19};

constexpr size_t hn_ksteps(int l) { return l == 0 ? 1 : (size_t)kConv[l].ksz * kConv[l].ksz * kConv[l].cin / 32; }
constexpr size_t hn_wbytes(int l) { return hn_ksteps(l) * (kConv[l].cout / 16) * 2048; }
constexpr size_t hn_woff(int l) { return l == 0 ? 0 : hn_woff(l - 1) + hn_wbytes(l - 1) + (size_t)kConv[l - 1].cout * 4; }
constexpr size_t hn_boff(int l) { return hn_woff(l) + hn_wbytes(l); }
constexpr size_t kHnBlobBytes = hn_boff(6) + 128 * 4;

struct HnState { std::string name; size_t numel; };
const std::vector<HnState> &hn_table() {
    static const std::vector<HnState> t = [] {
        std::vector<HnState> v;
        for (int l = 0; l < 7; ++l) {
            const std::string c = "features." + std::to_string(kFeatIdx[l]);
            const std::string b = "features." + std::to_string(kFeatIdx[l] + 1);
            v.push_back({c + ".weight", (size_t)kConv[l].cout * kConv[l].cin * kConv[l].ksz * kConv[l].ksz});
            v.push_back({b + ".running_mean", (size_t)kConv[l].cout});
            v.push_back({b + ".running_var", (size_t)kConv[l].cout});
        }
        return v;
    }();
    return t;
}

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ h8 zero8() {
    h8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.0f;
    return z;
}

// plain-f16 mode: round to nearest (the split mode truncates hi and carries the rest in lo)
__device__ __forceinline__ h2 round_pair(float v0, float v1) { return h2{(_Float16)v0, (_Float16)v1}; }

template <int CH>
__device__ __forceinline__ int swz(int q) {
    constexpr int SH = CH == 4 ? 2 : (CH == 8 ? 1 : 0);
    return (q >> SH) & (CH - 1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// In-kernel stamps (diagnostic build -DBALF_HN_STAMPS=1 only): thread 0 of every workgroup adds the cycles between
// consecutive HSTAMP(i) points to g_hn_stamp[kernel][i]; balf_debug_hn_stamps() reads them back (tools/hn_stamps.py).
#if BALF_HN_STAMPS
__device__ unsigned long long g_hn_stamp[8][8];
#define HSTAMP_DECL unsigned long long hs_prev = 0; (void)hs_prev
#define HSTAMP(i)                                                                                       \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long hs_now;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hs_now)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (threadIdx.x == 0) atomicAdd(&g_hn_stamp[Cfg::KID][(i)], (i) > 0 ? hs_now - hs_prev : 1ull);  \
        hs_prev = hs_now;                                                                               \
    } while (0)
#else
#define HSTAMP_DECL
#define HSTAMP(i)
#endif

template <int CIN_, int COUT_, int HIN_, int STRIDE_, int OROWS_, bool FUSE1_, bool SPLIT_ = true>
struct ConvCfg {
    static constexpr bool SPLIT = SPLIT_;                   // false: plain f16 operands (hi plane only, one product)
    static constexpr int NPL = SPLIT_ ? 2 : 1;              // activation planes
    static constexpr int KID = FUSE1_ ? 0 : (CIN_ == 32 ? 1 : (CIN_ == 64 ? (STRIDE_ == 1 ? 2 : 3) : 4));
    static constexpr int CIN = CIN_, COUT = COUT_, HIN = HIN_, STRIDE = STRIDE_, OROWS = OROWS_;
    static constexpr bool FUSE1 = FUSE1_;
    static constexpr int HOUT = HIN / STRIDE, WP = HIN + 2, IR = (OROWS - 1) * STRIDE + 3;
    static constexpr int NPIX = OROWS * HOUT, NT = NPIX / 16, MT = COUT / 16, WM = 2, WN = 4;
    static constexpr int MG = MT / WM, NG = NT / WN;
    static constexpr int KPT = CIN / 32, KS = 9 * KPT, CH = CIN / 8;
    static constexpr int NQ = IR * WP;                          // LDS pixels per plane
    static constexpr int PLANE_BYTES = NQ * CIN * 2;
    static constexpr int BANDS = HOUT / OROWS;
    static constexpr int INP_FLOATS = FUSE1 ? 34 * 34 : 0;
    static constexpr int LDS_BYTES = NPL * PLANE_BYTES + INP_FLOATS * 4 + 64;
    static constexpr int OCC = FUSE1 ? 3 : 2;       // register-allocation target (workgroups per CU); measured
    static_assert(MG * NG == 4, "4 waves per workgroup");
    static_assert(HOUT % OROWS == 0 && NPIX % 64 == 0, "band shape");
};

struct ConvArgs {
    const void *in;          // FUSE1: float patches [N][32][32]; else f16 planes [N][2][HIN][HIN][CIN]
    void *out;               // f16 planes [N][2][HOUT][HOUT][COUT]
    const char *w;           // this layer's fragments
    const float *bias;
    const char *w1;          // FUSE1: conv1 fragments + bias
    const float *bias1;
    int n_patches;
    const int *count;        // optional mask: patch p (global index p0 + local) is used iff (p % group) < count[p / group]
    int group, p0;
};

__device__ __forceinline__ bool hn_patch_used(const int *count, int group, int p) {
    return count == nullptr || (p % group) < count[p / group];
}

template <typename Cfg>
__global__ __launch_bounds__(256, Cfg::OCC) void hn_conv_kernel(ConvArgs a) {
    constexpr int CIN = Cfg::CIN, COUT = Cfg::COUT, HIN = Cfg::HIN, S = Cfg::STRIDE, HOUT = Cfg::HOUT, WP = Cfg::WP;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, MT = Cfg::MT, CH = Cfg::CH, KPT = Cfg::KPT, IR = Cfg::IR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *act = smem;                                       // [plane][q][CIN] halves, chunk-swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int patch = blockIdx.x / Cfg::BANDS, band = blockIdx.x % Cfg::BANDS;
    if (!hn_patch_used(a.count, a.group, a.p0 + patch)) return;      // masked slot: no work, its buffers stay stale
    const int r0 = band * Cfg::OROWS;                       // first output row
    const int r_in0 = r0 * S - 1;                           // input row of LDS row 0
    HSTAMP_DECL;
    HSTAMP(0);

    if constexpr (Cfg::FUSE1) {
        // ---- patch normalisation (hardnet_pytorch.py:58-63) ----
        float *inp = reinterpret_cast<float *>(smem + Cfg::NPL * Cfg::PLANE_BYTES); // [34][34], zero frame
        float *red = inp + 34 * 34;
        const float *src = static_cast<const float *>(a.in) + (size_t)patch * (kPS * kPS);
        const f4 v = *reinterpret_cast<const f4 *>(src + 4 * tid);
        for (int i = tid; i < 34 * 34; i += 256) inp[i] = 0.0f;
        float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) * (1.0f / 1024.0f);
        const f4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
        s = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
        if (lane == 0) red[4 + wave] = s;
        __syncthreads();
        const float sd = sqrtf((red[4] + red[5] + red[6] + red[7]) * (1.0f / 1023.0f)) + 1e-7f;
        const int py = (4 * tid) >> 5, px = (4 * tid) & 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) inp[(py + 1) * 34 + px + 1 + j] = d[j] / sd;
        __syncthreads();
        HSTAMP(1);      // patch load + normalisation

        // ---- conv1 + BN + ReLU for the band's 10 x 34 pixel frame, straight into the LDS image ----
        HL a1[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a1[m].hi = *reinterpret_cast<const h8 *>(a.w1 + (m * 2 + 0) * 1024 + lane * 16);
            if constexpr (Cfg::SPLIT) a1[m].lo = *reinterpret_cast<const h8 *>(a.w1 + (m * 2 + 1) * 1024 + lane * 16);
        }
        f4 bias1[2];                                    // loaded once: hipcc leaves a load written inside the
#pragma unroll                                          // tile loop there, and every tile then pays an L2 round trip
        for (int m = 0; m < 2; ++m) bias1[m] = *reinterpret_cast<const f4 *>(a.bias1 + 16 * m + 4 * g);
        constexpr int NQT = (Cfg::NQ + 15) / 16;
        for (int t = wave; t < NQT; t += 4) {
            const int q = 16 * t + n;
            const int iy = q / WP, ixp = q - iy * WP;
            const int y = r_in0 + iy, x = ixp - 1;
            const bool inside = q < Cfg::NQ && y >= 0 && y < HIN && x >= 0 && x < HIN;
            const int yc = inside ? y : 0, xc = inside ? x : 0;
            float tv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tv[j] = 0.0f;
            if (g == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tv[j] = inp[(yc + j / 3) * 34 + xc + j % 3];
            } else if (g == 1) {
                tv[0] = inp[(yc + 2) * 34 + xc + 2];
            }
            HL b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h2 hh, ll;
                if constexpr (Cfg::SPLIT) split_pair(tv[2 * j], tv[2 * j + 1], hh, ll);
                else { hh = round_pair(tv[2 * j], tv[2 * j + 1]); ll = hh; }
                b.hi[2 * j] = hh[0]; b.hi[2 * j + 1] = hh[1];
                b.lo[2 * j] = ll[0]; b.lo[2 * j + 1] = ll[1];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f4 c = {0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (Cfg::SPLIT) c = mfma16x3(a1[m], b, c);
                else c = mfma16(a1[m].hi, b.hi, c);
                const f4 bb = bias1[m];
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = inside ? fmaxf(c[r] + bb[r], 0.0f) : 0.0f;
                h2 h01, l01, h23, l23;
                if constexpr (Cfg::SPLIT) { split_pair(o[0], o[1], h01, l01); split_pair(o[2], o[3], h23, l23); }
                else { h01 = round_pair(o[0], o[1]); h23 = round_pair(o[2], o[3]); l01 = h01; l23 = h23; }
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 hv = {h01[0], h01[1], h23[0], h23[1]};
                const h4 lv = {l01[0], l01[1], l23[0], l23[1]};
                if (q < Cfg::NQ) {
                    const int off = q * (CIN * 2) + (((2 * m + (g >> 1)) ^ swz<CH>(q)) << 4) + (g & 1) * 8;
                    *reinterpret_cast<h4 *>(act + off) = hv;
                    if constexpr (Cfg::SPLIT) *reinterpret_cast<h4 *>(act + Cfg::PLANE_BYTES + off) = lv;
                }
            }
        }
    } else {
        // ---- copy the input band (both planes) into the swizzled LDS image, zero frame included ----
        // An image row of one plane is 128 16-byte chunks in every layer (HIN * CIN / 8), so the 256 threads take
        // two rows per pass and all index arithmetic is shifts and masks.  All global loads are issued before the
        // first LDS write, so a workgroup pays one HBM round trip.
        static_assert(HIN * CH == 128, "one image row of one plane = 128 chunks");
        const _Float16 *in = static_cast<const _Float16 *>(a.in) + (size_t)patch * Cfg::NPL * HIN * HIN * CIN;
        constexpr int ROWS = Cfg::NPL * IR, PASSES = (ROWS + 1) / 2;
        const int rsub = tid >> 7, xc = tid & 127;              // row within the pass, chunk within the row
        const int x = xc / CH, c = xc % CH;
        h8 stage[PASSES];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = 2 * ps + rsub;                      // plane * IR + iy
            const int plane = row >= IR ? 1 : 0;
            const int y = r_in0 + row - plane * IR;
            stage[ps] = zero8();
            if (row < ROWS && y >= 0 && y < HIN)
                stage[ps] = *reinterpret_cast<const h8 *>(in + ((size_t)(plane * HIN + y) * HIN) * CIN + xc * 8);
        }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = 2 * ps + rsub;
            const int plane = row >= IR ? 1 : 0;
            const int q = (row - plane * IR) * WP + x + 1;
            if (row < ROWS)
                *reinterpret_cast<h8 *>(act + plane * Cfg::PLANE_BYTES + q * (CIN * 2) + ((c ^ swz<CH>(q)) << 4)) = stage[ps];
        }
        // the two frame columns (the frame rows, if any, were written as zeros above)
        for (int i = tid; i < ROWS * 2 * CH; i += 256) {
            const int row = i / (2 * CH), e = i % (2 * CH);
            const int plane = row >= IR ? 1 : 0;
            const int q = (row - plane * IR) * WP + (e >= CH ? WP - 1 : 0);
            *reinterpret_cast<h8 *>(act + plane * Cfg::PLANE_BYTES + q * (CIN * 2) + (((e % CH) ^ swz<CH>(q)) << 4)) = zero8();
        }
    }
    __syncthreads();
    HSTAMP(2);          // conv1 into LDS (fused kernel) / band load

    // ---- implicit GEMM over 9 taps x CIN ----
    const int mg = wave / Cfg::NG, ng = wave % Cfg::NG;
    int q0[WN];
#pragma unroll
    for (int t = 0; t < WN; ++t) {
        const int p = 16 * (ng * WN + t) + n;
        const int oy = p / HOUT, ox = p - oy * HOUT;
        q0[t] = oy * S * WP + ox * S;
    }
    f4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int t = 0; t < WN; ++t) acc[m][t] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    f4 bias[WM];                                        // requested before the K loop, used after it
#pragma unroll
    for (int m = 0; m < WM; ++m) bias[m] = *reinterpret_cast<const f4 *>(a.bias + 16 * (mg * WM + m) + 4 * g);
    const char *wl = a.w + (size_t)(mg * WM) * 2048 + lane * 16;
    auto load_a = [&](int ks, HL (&dst)[WM]) {
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const char *p = wl + ((size_t)ks * MT + m) * 2048;
            dst[m].hi = *reinterpret_cast<const h8 *>(p);
            if constexpr (Cfg::SPLIT) dst[m].lo = *reinterpret_cast<const h8 *>(p + 1024);
        }
    };
    auto load_b = [&](int ks, HL (&dst)[WN]) {          // ks is a compile-time constant after unrolling
        const int tap = ks / KPT, kk = ks - tap * KPT;
        const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            const int q = q0[t] + ky * WP + kx;
#ifdef BALF_HN_ABLATE_BANK
            const int off = ((q * (CIN * 2)) & ~1023) + lane * 16;      // timing experiment: conflict-free, wrong data
#else
            const int off = q * (CIN * 2) + (((kk * 4 + g) ^ swz<CH>(q)) << 4);
#endif
            dst[t].hi = *reinterpret_cast<const h8 *>(act + off);
            if constexpr (Cfg::SPLIT) dst[t].lo = *reinterpret_cast<const h8 *>(act + Cfg::PLANE_BYTES + off);
        }
    };
    // weights (global, L2 latency) run two K-steps ahead, LDS fragments one
    HL abuf[3][WM], bbuf[2][WN];
    load_a(0, abuf[0]);
    if (Cfg::KS > 1) load_a(1, abuf[1]);
    load_b(0, bbuf[0]);
#pragma unroll
    for (int ks = 0; ks < Cfg::KS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        const int ac = ks % 3;
        if (ks + 2 < Cfg::KS) load_a(ks + 2, abuf[(ks + 2) % 3]);
        if (ks + 1 < Cfg::KS) load_b(ks + 1, bbuf[nxt]);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch distances exact
        if constexpr (Cfg::SPLIT) {
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int t = 0; t < WN; ++t) acc[m][t] = mfma16(abuf[ac][m].lo, bbuf[cur][t].hi, acc[m][t]);
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int t = 0; t < WN; ++t) acc[m][t] = mfma16(abuf[ac][m].hi, bbuf[cur][t].lo, acc[m][t]);
        }
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int t = 0; t < WN; ++t) acc[m][t] = mfma16(abuf[ac][m].hi, bbuf[cur][t].hi, acc[m][t]);
        __builtin_amdgcn_sched_barrier(0);
    }

    HSTAMP(3);          // K loop
    // ---- bias (folded BatchNorm) + ReLU, split, store both planes ----
    _Float16 *out = static_cast<_Float16 *>(a.out) + (size_t)patch * Cfg::NPL * HOUT * HOUT * COUT;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int co = 16 * (mg * WM + m) + 4 * g;
        const f4 bb = bias[m];
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            const int p = 16 * (ng * WN + t) + n;
            const int oy = r0 + p / HOUT, ox = p % HOUT;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[m][t][r] + bb[r], 0.0f);
            h2 h01, l01, h23, l23;
            if constexpr (Cfg::SPLIT) { split_pair(o[0], o[1], h01, l01); split_pair(o[2], o[3], h23, l23); }
            else { h01 = round_pair(o[0], o[1]); h23 = round_pair(o[2], o[3]); l01 = h01; l23 = h23; }
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 hv = {h01[0], h01[1], h23[0], h23[1]};
            const h4 lv = {l01[0], l01[1], l23[0], l23[1]};
            const size_t e = ((size_t)oy * HOUT + ox) * COUT + co;
            *reinterpret_cast<h4 *>(out + e) = hv;
            if constexpr (Cfg::SPLIT) *reinterpret_cast<h4 *>(out + (size_t)HOUT * HOUT * COUT + e) = lv;
        }
    }
    HSTAMP(4);          // epilogue
}

// The last layer: out[co][patch] = sum_k W7[co][k] a6[patch][k], k = (y*8 + x)*128 + c -- exactly the order a6 is
// stored in.  128 patches per workgroup; a wave owns all 8 channel tiles of its 32 patches, so the L2 norm never
// leaves the wave.  The weights (4 MiB, shared by every workgroup) stream through a double-buffered LDS stage of
// two K-steps; the B operand comes straight from HBM one stage ahead.
struct FcArgs {
    const _Float16 *in;      // [N][2][8192]
    float *desc;             // [N][128]
    const char *w;
    const float *bias;
    int n_patches;
    const int *count;        // optional mask, as ConvArgs
    int group;
};

constexpr int kFcStageKs = 2, kFcStageBytes = kFcStageKs * 8 * 2048, kFcLds = 2 * kFcStageBytes;

template <bool SPLIT>
__global__ __launch_bounds__(256) void hn_fc_kernel(FcArgs a) {
    constexpr int NPL = SPLIT ? 2 : 1;
    constexpr int MT = 8, WN = 2, KS = kFcK / 32, NST = KS / kFcStageKs;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int p0 = blockIdx.x * 128 + wave * 32;
    const _Float16 *bp[WN];
    bool used[WN];
    int any = 0;
#pragma unroll
    for (int t = 0; t < WN; ++t) {
        int p = p0 + 16 * t + n;
        used[t] = p < a.n_patches && hn_patch_used(a.count, a.group, p);
        any |= used[t];
        p = p < a.n_patches ? p : a.n_patches - 1;
        bp[t] = a.in + (size_t)p * NPL * kFcK + g * 8;
    }
    if (a.count != nullptr && !__syncthreads_or(any)) {      // a tile of masked slots: zero descriptors, no GEMM
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            const int p = p0 + 16 * t + n;
            if (p < a.n_patches)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    *reinterpret_cast<f4 *>(a.desc + (size_t)p * kDesc + 16 * m + 4 * g) = f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        return;
    }
    f4 acc[MT][WN];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < WN; ++t) acc[m][t] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    constexpr int CP = kFcStageBytes / (256 * 16);          // 16-byte pieces per thread per stage
    h8 wreg[CP];
    HL breg[kFcStageKs][WN];
    auto fetch_w = [&](int st) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            wreg[c] = *reinterpret_cast<const h8 *>(a.w + (size_t)st * kFcStageBytes + (size_t)(c * 256 + tid) * 16);
    };
    auto put_w = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CP; ++c)
            *reinterpret_cast<h8 *>(smem + buf * kFcStageBytes + (c * 256 + tid) * 16) = wreg[c];
    };
    auto fetch_b = [&](int st) {
#pragma unroll
        for (int k2 = 0; k2 < kFcStageKs; ++k2)
#pragma unroll
            for (int t = 0; t < WN; ++t) {
                breg[k2][t].hi = *reinterpret_cast<const h8 *>(bp[t] + (st * kFcStageKs + k2) * 32);
                if constexpr (SPLIT) breg[k2][t].lo = *reinterpret_cast<const h8 *>(bp[t] + kFcK + (st * kFcStageKs + k2) * 32);
            }
    };
    fetch_w(0);
    fetch_b(0);
    put_w(0);
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < NST; ++st) {
        const int buf = st & 1;
        HL bcur[kFcStageKs][WN];
#pragma unroll
        for (int k2 = 0; k2 < kFcStageKs; ++k2)
#pragma unroll
            for (int t = 0; t < WN; ++t) bcur[k2][t] = breg[k2][t];
        const int nst = st + 1 < NST ? st + 1 : st;
        fetch_w(nst);
        fetch_b(nst);
#pragma unroll
        for (int k2 = 0; k2 < kFcStageKs; ++k2)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                HL wa;
                const char *p = smem + buf * kFcStageBytes + ((k2 * MT + m) * 2048) + lane * 16;
                wa.hi = *reinterpret_cast<const h8 *>(p);
                wa.lo = *reinterpret_cast<const h8 *>(p + 1024);
#pragma unroll
                for (int t = 0; t < WN; ++t) {
                    if constexpr (SPLIT) acc[m][t] = mfma16x3(wa, bcur[k2][t], acc[m][t]);
                    else acc[m][t] = mfma16(wa.hi, bcur[k2][t].hi, acc[m][t]);
                }
            }
        put_w(buf ^ 1);
        __syncthreads();
    }
    // BatchNorm bias, then x / sqrt(sum x^2 + 1e-10) over the 128 channels of each patch (hardnet_pytorch.py:11-15)
    float part[WN];
#pragma unroll
    for (int t = 0; t < WN; ++t) part[t] = 0.0f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const f4 bb = *reinterpret_cast<const f4 *>(a.bias + 16 * m + 4 * g);
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            acc[m][t] += bb;
#pragma unroll
            for (int r = 0; r < 4; ++r) part[t] += acc[m][t][r] * acc[m][t][r];
        }
    }
#pragma unroll
    for (int t = 0; t < WN; ++t) {
        part[t] += __shfl_xor(part[t], 16);
        part[t] += __shfl_xor(part[t], 32);
        const int p = p0 + 16 * t + n;
        const float inv = 1.0f / sqrtf(part[t] + 1e-10f);
        if (p < a.n_patches) {
            // masked slots get exact zeros: their accumulators were fed stale workspace contents (possibly NaN/Inf)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                *reinterpret_cast<f4 *>(a.desc + (size_t)p * kDesc + 16 * m + 4 * g) =
                    used[t] ? acc[m][t] * inv : f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
}

template <bool SP> using CfgL2 = ConvCfg<32, 32, 32, 1, 8, true, SP>;
template <bool SP> using CfgL3 = ConvCfg<32, 64, 32, 2, 8, false, SP>;
template <bool SP> using CfgL4 = ConvCfg<64, 64, 16, 1, 8, false, SP>;
template <bool SP> using CfgL5 = ConvCfg<64, 128, 16, 2, 8, false, SP>;
template <bool SP> using CfgL6 = ConvCfg<128, 128, 8, 1, 8, false, SP>;

constexpr size_t kBufA = (size_t)2 * 32 * 32 * 32 * 2;      // a2 (largest tenant): 128 KiB per patch
constexpr size_t kBufB = (size_t)2 * 16 * 16 * 64 * 2;      // a3: 64 KiB per patch
constexpr size_t kBufA6 = (size_t)2 * kFcK * 2;             // a6: 32 KiB per patch, kept for the whole batch
constexpr int kChunkDefault = 4096;                         // patches per pass through the convolution kernels
int hn_chunk() {
    static const int c = [] {
        const char *e = getenv("BALF_HN_CHUNK");            // tuning aid
        const int v = e ? atoi(e) : 0;
        return v >= 64 ? v : kChunkDefault;
    }();
    return c;
}

template <typename Cfg>
int launch_conv(int slot, const ConvArgs &a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&hn_conv_kernel<Cfg>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES) != hipSuccess)
            return BALF_ERR_LAUNCH;
        attr_set = true;
    }
    BALF_PROF(slot, st, (hn_conv_kernel<Cfg><<<a.n_patches * Cfg::BANDS, 256, Cfg::LDS_BYTES, st>>>(a)));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

void pack_layer(char *blob, int l, const float *w, const float *mean, const float *var) {
    const HnConv c = kConv[l];
    const int taps = c.ksz * c.ksz, K = taps * c.cin, KS = (int)hn_ksteps(l), MT = c.cout / 16;
    _Float16 *dst = reinterpret_cast<_Float16 *>(blob + hn_woff(l));
    float *bias = reinterpret_cast<float *>(blob + hn_boff(l));
    std::vector<double> rstd(c.cout);
    for (int co = 0; co < c.cout; ++co) {
        rstd[co] = 1.0 / sqrt((double)var[co] + 1e-5);
        bias[co] = (float)(-(double)mean[co] * rstd[co]);
    }
    for (int ks = 0; ks < KS; ++ks)
        for (int mt = 0; mt < MT; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 16 * mt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
                    float v = 0.0f;
                    if (k < K) {
                        const int tap = k / c.cin, ci = k % c.cin;
                        v = (float)((double)w[((size_t)co * c.cin + ci) * taps + tap] * rstd[co]);
                    }
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t tile = ((size_t)ks * MT + mt) * 1024;
                    dst[tile + lane * 8 + j] = hi;
                    dst[tile + 512 + lane * 8 + j] = lo;
                }
}

}  // namespace
}  // namespace balf

using namespace balf;

#if BALF_HN_STAMPS
extern "C" int balf_debug_hn_stamps(unsigned long long *out, int reset) {
    if (reset) {
        static unsigned long long z[64];
        return hipMemcpyToSymbol(HIP_SYMBOL(g_hn_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hn_stamp), 64 * 8) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int balf_hardnet_num_state_tensors(void) { return (int)hn_table().size(); }

extern "C" const char *balf_hardnet_state_tensor_name(int i) {
    if (i < 0 || i >= (int)hn_table().size()) return nullptr;
    return hn_table()[i].name.c_str();
}

extern "C" size_t balf_hardnet_state_tensor_numel(int i) {
    if (i < 0 || i >= (int)hn_table().size()) return 0;
    return hn_table()[i].numel;
}

extern "C" size_t balf_hardnet_packed_weights_bytes(void) { return kHnBlobBytes; }

extern "C" int balf_hardnet_pack_weights(const float *const *tensors, int n_tensors, void *packed_host,
                                         size_t packed_bytes) {
    if (!tensors || !packed_host || n_tensors != (int)hn_table().size()) return BALF_ERR_ARG;
    if (packed_bytes < kHnBlobBytes) return BALF_ERR_WORKSPACE;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return BALF_ERR_ARG;
    memset(packed_host, 0, kHnBlobBytes);
    for (int l = 0; l < 7; ++l)
        pack_layer(static_cast<char *>(packed_host), l, tensors[3 * l], tensors[3 * l + 1], tensors[3 * l + 2]);
    return BALF_OK;
}

extern "C" size_t balf_hardnet_workspace_bytes(int n_patches) {
    if (n_patches <= 0) return 0;
    const size_t c = n_patches < hn_chunk() ? n_patches : hn_chunk();
    return c * (kBufA + kBufB) + (size_t)n_patches * kBufA6;
}

namespace {
template <bool SP>
int hardnet_forward_impl(const void *packed_dev, const float *patches_dev, int n_patches, int group, const int32_t *count_dev,
                         float *desc_dev, void *workspace_dev, hipStream_t st) {
    const char *blob = static_cast<const char *>(packed_dev);
    const int chunk = n_patches < hn_chunk() ? n_patches : hn_chunk();
    char *bufA = static_cast<char *>(workspace_dev);
    char *bufB = bufA + (size_t)chunk * kBufA;
    char *a6 = bufB + (size_t)chunk * kBufB;
    constexpr size_t a6_bytes = SP ? kBufA6 : kBufA6 / 2;              // one plane in plain-f16 mode
    auto bias = [&](int l) { return reinterpret_cast<const float *>(blob + hn_boff(l)); };
    for (int c0 = 0; c0 < n_patches; c0 += chunk) {
        const int n = n_patches - c0 < chunk ? n_patches - c0 : chunk;
        int rc;
        ConvArgs a{};
        a.n_patches = n;
        a.count = count_dev; a.group = group > 0 ? group : 1; a.p0 = c0;
        a.in = patches_dev + (size_t)c0 * kPS * kPS; a.out = bufA;
        a.w = blob + hn_woff(1); a.bias = bias(1); a.w1 = blob + hn_woff(0); a.bias1 = bias(0);
        if ((rc = launch_conv<CfgL2<SP>>(balf_prof::kHnConv2, a, st)) != BALF_OK) return rc;
        a.in = bufA; a.out = bufB; a.w = blob + hn_woff(2); a.bias = bias(2);
        if ((rc = launch_conv<CfgL3<SP>>(balf_prof::kHnConv3, a, st)) != BALF_OK) return rc;
        a.in = bufB; a.out = bufA; a.w = blob + hn_woff(3); a.bias = bias(3);
        if ((rc = launch_conv<CfgL4<SP>>(balf_prof::kHnConv4, a, st)) != BALF_OK) return rc;
        a.in = bufA; a.out = bufB; a.w = blob + hn_woff(4); a.bias = bias(4);
        if ((rc = launch_conv<CfgL5<SP>>(balf_prof::kHnConv5, a, st)) != BALF_OK) return rc;
        a.in = bufB; a.out = a6 + (size_t)c0 * a6_bytes; a.w = blob + hn_woff(5); a.bias = bias(5);
        if ((rc = launch_conv<CfgL6<SP>>(balf_prof::kHnConv6, a, st)) != BALF_OK) return rc;
    }
    // the final GEMM runs once over the whole batch (a chunk alone would fill 32 of the 256 CUs)
    FcArgs f{reinterpret_cast<const _Float16 *>(a6), desc_dev, blob + hn_woff(6), bias(6), n_patches, count_dev,
             group > 0 ? group : 1};
    BALF_PROF(balf_prof::kHnFc, st, (hn_fc_kernel<SP><<<balf_ceil_div(n_patches, 128), 256, kFcLds, st>>>(f)));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}
}  // namespace

extern "C" int balf_hardnet_forward_ex(const void *packed_dev, const float *patches_dev, int n_patches, int group,
                                       const int32_t *count_dev, int precision, float *desc_dev, void *workspace_dev,
                                       size_t workspace_bytes, void *stream) {
    if (!packed_dev || !patches_dev || !desc_dev || !workspace_dev || n_patches <= 0) return BALF_ERR_ARG;
    if (count_dev && (group <= 0 || n_patches % group != 0)) return BALF_ERR_ARG;
    if (precision != BALF_HARDNET_SPLIT_F16 && precision != BALF_HARDNET_PLAIN_F16) return BALF_ERR_ARG;
    if (workspace_bytes < balf_hardnet_workspace_bytes(n_patches)) return BALF_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return precision == BALF_HARDNET_SPLIT_F16
               ? hardnet_forward_impl<true>(packed_dev, patches_dev, n_patches, group, count_dev, desc_dev, workspace_dev, st)
               : hardnet_forward_impl<false>(packed_dev, patches_dev, n_patches, group, count_dev, desc_dev, workspace_dev, st);
}

extern "C" int balf_hardnet_forward_masked(const void *packed_dev, const float *patches_dev, int n_patches, int group,
                                           const int32_t *count_dev, float *desc_dev, void *workspace_dev,
                                           size_t workspace_bytes, void *stream) {
    return balf_hardnet_forward_ex(packed_dev, patches_dev, n_patches, group, count_dev, BALF_HARDNET_SPLIT_F16, desc_dev,
                                   workspace_dev, workspace_bytes, stream);
}

extern "C" int balf_hardnet_forward(const void *packed_dev, const float *patches_dev, int n_patches, float *desc_dev,
                                    void *workspace_dev, size_t workspace_bytes, void *stream) {
    return balf_hardnet_forward_masked(packed_dev, patches_dev, n_patches, 0, nullptr, desc_dev, workspace_dev,
                                       workspace_bytes, stream);
}
