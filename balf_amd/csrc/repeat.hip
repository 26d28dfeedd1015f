// Repeatability evaluation on gfx950.
//
// Reference: /root/reference/balf/benchmark_test/repeatability_tools.py:379-512 (compute_repeatability,
// intersection_area, union_area) and /root/reference/balf/benchmark_test/geometry_tools.py:43-86
// (apply_homography_to_points, getAff).  The reference walks all Ns x Nd pairs in a Python double loop, fills two
// dense overlap matrices, argsorts each and assigns greedily; callers: train_utils.py:189,257 (validation) and
// datasets/dataset_utils.py:332.
//
// Here nothing dense is stored.  All arithmetic is float64, like the reference's Python floats:
//   rep_count_kernel   one workgroup per source point: circle-overlap of every pair, counts the pairs whose
//                      single-scale / multi-scale overlap reaches 1 - overlap_err, and the "possible match" flag
//   rep_scan_kernel    exclusive scan of the per-row counts
//   rep_fill_kernel    one wave per source point: ordered compaction of the candidate pairs (key = overlap bits,
//                      value = flat index i*Nd + j, written in flat order)
//   rep_sort_kernel    stable LSD radix sort, descending by overlap => equal overlaps stay in flat-index order
//   rep_greedy_kernel  one wave walks the sorted candidates 64 at a time; visited bitmaps in LDS; the error sum is
//                      accumulated in the reference's order
// The candidate count is data dependent, so balf_repeatability synchronises the stream once to read it.
#include "common.h"

namespace balf {
namespace {

constexpr double kPi = 3.141592653589793;
constexpr double kEpsF64 = 2.220446049250313e-16;          // np.finfo(float).eps
constexpr double kEpsF32 = 1.1920928955078125e-07;         // np.finfo(np.float32).eps
constexpr int kMaxPoints = 65536;                          // visited bitmaps live in LDS

struct RepParams {
    double thr, eps, dist_match, radius, max_dist;
};

__device__ __forceinline__ double inter_area(double R, double r, double d) {
    if (d <= fabs(R - r)) { const double m = fmin(R, r); return kPi * (m * m); }
    if (d >= r + R) return 0.0;
    const double r2 = r * r, R2 = R * R, d2 = d * d;
    const double alpha = acos((d2 + r2 - R2) / (2 * d * r));
    const double beta = acos((d2 + R2 - r2) / (2 * d * R));
    return r2 * alpha + R2 * beta - 0.5 * (r2 * sin(2 * alpha) + R2 * sin(2 * beta));
}

__device__ __forceinline__ void pair_overlaps(double sx, double sy, double sr, double tx, double ty, double tr,
                                              const RepParams &p, double &single, double &multi, bool &possible) {
    const double dx = sx - tx, dy = sy - ty;
    const double dist = sqrt(dx * dx + dy * dy);
    possible = dist <= p.dist_match;
    single = 0.0; multi = 0.0;
    if (dist > p.max_dist) return;
    const double f = p.radius / (fmax(sr, tr) + kEpsF64);
    double I = inter_area(f * sr, f * tr, dist);
    double U = kPi * ((f * sr) * (f * sr)) + kPi * ((f * tr) * (f * tr)) - I + p.eps;
    multi = I / U;
    I = inter_area(p.radius, p.radius, dist);
    U = kPi * (p.radius * p.radius) + kPi * (p.radius * p.radius) - I + p.eps;
    single = I / U;
}

__global__ __launch_bounds__(256) void rep_count_kernel(const double *src, int ns, const double *dst, int nd, RepParams p,
                                                        int *cnt_s, int *cnt_m, int *poss) {
    __shared__ int red[3];
    const int i = blockIdx.x;
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();
    const double sx = src[3 * i], sy = src[3 * i + 1], sr = src[3 * i + 2];
    int cs = 0, cm = 0, ps = 0;
    for (int j = threadIdx.x; j < nd; j += 256) {
        double s, m; bool po;
        pair_overlaps(sx, sy, sr, dst[3 * j], dst[3 * j + 1], dst[3 * j + 2], p, s, m, po);
        cs += s >= p.thr; cm += m >= p.thr; ps |= po;
    }
    if (cs) atomicAdd(&red[0], cs);
    if (cm) atomicAdd(&red[1], cm);
    if (ps) atomicOr(&red[2], 1);
    __syncthreads();
    if (threadIdx.x == 0) { cnt_s[i] = red[0]; cnt_m[i] = red[1]; poss[i] = red[2]; }
}

// off[i] = sum_{k<i} cnt[k]; totals = {sum cnt_s, sum cnt_m, sum poss}
__global__ __launch_bounds__(1024) void rep_scan_kernel(const int *cnt_s, const int *cnt_m, const int *poss, int ns,
                                                        int *off_s, int *off_m, int *totals, int *poss_out) {
    __shared__ int wsum[3][16];
    __shared__ int base[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) base[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < ns; i0 += 1024) {
        const int i = i0 + tid;
        int v[3] = {i < ns ? cnt_s[i] : 0, i < ns ? cnt_m[i] : 0, i < ns ? poss[i] : 0};
        int incl[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int x = v[c];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            incl[c] = x;
            if (lane == 63) wsum[c][wave] = x;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int before = base[c];
            for (int w = 0; w < wave; ++w) before += wsum[c][w];
            incl[c] += before;
        }
        if (i < ns) { off_s[i] = incl[0] - v[0]; off_m[i] = incl[1] - v[1]; }
        __syncthreads();
        if (tid == 1023) { base[0] = incl[0]; base[1] = incl[1]; base[2] = incl[2]; }
        __syncthreads();
    }
    if (tid == 0) { totals[0] = base[0]; totals[1] = base[1]; totals[2] = base[2]; *poss_out = base[2]; }
}

__global__ __launch_bounds__(64) void rep_fill_kernel(const double *src, int ns, const double *dst, int nd, RepParams p,
                                                      const int *off_s, const int *off_m, int max_edges,
                                                      unsigned long long *key_s, unsigned *val_s, unsigned long long *key_m,
                                                      unsigned *val_m) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double sx = src[3 * i], sy = src[3 * i + 1], sr = src[3 * i + 2];
    int ws = off_s[i], wm = off_m[i];
    for (int j0 = 0; j0 < nd; j0 += 64) {
        const int j = j0 + lane;
        double s = 0.0, m = 0.0; bool po;
        if (j < nd) pair_overlaps(sx, sy, sr, dst[3 * j], dst[3 * j + 1], dst[3 * j + 2], p, s, m, po);
        const bool ks = j < nd && s >= p.thr, km = j < nd && m >= p.thr;
        const unsigned long long bs = __ballot(ks), bm = __ballot(km);
        const unsigned long long below = (1ull << lane) - 1ull;
        // a list longer than the caller's max_edges is cut here and reported by rep_greedy_kernel (count -1): the host never
        // learns the length, so nothing waits for it
        if (ks) { const int o = ws + __popcll(bs & below); if (o < max_edges) { key_s[o] = __double_as_longlong(s); val_s[o] = (unsigned)(i * nd + j); } }
        if (km) { const int o = wm + __popcll(bm & below); if (o < max_edges) { key_m[o] = __double_as_longlong(m); val_m[o] = (unsigned)(i * nd + j); } }
        ws += __popcll(bs); wm += __popcll(bm);
    }
}

// out: found[which], err[which], corr[which][k] = (x_pos = dst index, y_pos = src index) in assignment order
// *n_edges_dev > max_edges (the candidate list did not fit the workspace): found = -1, no correspondences
__global__ __launch_bounds__(64) void rep_greedy_kernel(const unsigned long long *keys, const unsigned *vals, const int *n_edges_dev,
                                                        int max_edges, int nd, int *found_out, double *err_out, int *corr, int cap) {
    __shared__ unsigned vis_x[kMaxPoints / 32], vis_y[kMaxPoints / 32];
    const int lane = threadIdx.x;
    const int n_edges = *n_edges_dev;
    if (n_edges > max_edges) {
        if (lane == 0) { *found_out = -1; *err_out = 0.0; }
        for (int k = lane; k < cap; k += 64) { corr[2 * k] = -1; corr[2 * k + 1] = -1; }
        return;
    }
    for (int k = lane; k < kMaxPoints / 32; k += 64) { vis_x[k] = 0u; vis_y[k] = 0u; }
    __syncthreads();
    int found = 0;
    double err = 0.0;
    for (int base = 0; base < n_edges; base += 64) {
        const int e = base + lane;
        unsigned idx = 0; double w = 0.0;
        if (e < n_edges) { idx = vals[e]; w = __longlong_as_double((long long)keys[e]); }
        const int yi = (int)(idx / (unsigned)nd), xj = (int)(idx % (unsigned)nd);
        const int lim = n_edges - base < 64 ? n_edges - base : 64;
        for (int l = 0; l < lim; ++l) {
            const int y = __shfl(yi, l), x = __shfl(xj, l);
            const double wl = __shfl(w, l);
            const bool taken = ((vis_x[x >> 5] >> (x & 31)) & 1u) || ((vis_y[y >> 5] >> (y & 31)) & 1u);
            if (!taken) {
                if (lane == 0) {
                    vis_x[x >> 5] |= 1u << (x & 31);
                    vis_y[y >> 5] |= 1u << (y & 31);
                    if (found < cap) { corr[2 * found] = x; corr[2 * found + 1] = y; }
                }
                found += 1;
                err += 1.0 - wl;
            }
            __syncthreads();        // single wave: orders lane 0's LDS update before the next read
        }
    }
    if (lane == 0) { *found_out = found; *err_out = err; }
    for (int k = found + lane; k < cap; k += 64) { corr[2 * k] = -1; corr[2 * k + 1] = -1; }
}

__global__ void homography_kernel(const double *pts, int n, const double *h, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pts[4 * i], y = pts[4 * i + 1], r = pts[4 * i + 2];
    const double den = h[6] * x + h[7] * y + h[8];
    const double nx = h[0] * x + h[1] * y + h[2], ny = h[3] * x + h[4] * y + h[5];
    const double fxdx = h[0] / den - nx * h[6] / (den * den), fxdy = h[1] / den - nx * h[7] / (den * den);
    const double fydx = h[3] / den - ny * h[6] / (den * den), fydy = h[4] / den - ny * h[7] / (den * den);
    const double tmp = r * r + kEpsF32;
    out[4 * i] = nx / den; out[4 * i + 1] = ny / den;
    out[4 * i + 2] = sqrt(tmp * fabs(fxdx * fydy - fxdy * fydx));
    out[4 * i + 3] = pts[4 * i + 3];
}

// Stable LSD radix sort of (key, value) pairs, DESCENDING by the 64-bit key, 4 bits per pass, one workgroup of 16
// waves.  Wave w owns the contiguous range [w * per, (w + 1) * per) of the input and walks it 64 elements at a time, so
// "input order" is (wave, row, lane) and a pass keeps it among equal digits: per row the lane's rank among the lanes with
// its digit comes from a ballot, per wave the digit counts go through a [digit][wave] table in LDS whose exclusive scan
// (digit-major) gives every wave its output cursor per digit.  Passes whose digit is the same for all keys are skipped
// (overlaps lie in [1 - overlap_err, 1]: the sign, exponent and leading mantissa digits agree), so the 16 possible passes
// are ~11 in practice.  The pairs ping-pong between (k0, v0) and (k1, v1); the result always ends in (k1, v1).
// (Round 2 called hipcub::DeviceRadixSort here; the library now has no third-party device code.)
constexpr int kSortWaves = 16;
__global__ __launch_bounds__(kSortWaves * 64) void rep_sort_kernel(unsigned long long *k0, unsigned *v0, unsigned long long *k1,
                                                                  unsigned *v1, const int *n_dev, int max_edges) {
    __shared__ int hist[16][kSortWaves];
    __shared__ int uniform_digit;
    const int n = *n_dev;
    if (n <= 0 || n > max_edges) return;                          // nothing to sort / overflow (reported by rep_greedy_kernel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = ((n + kSortWaves - 1) / kSortWaves + 63) / 64 * 64;
    const int lo = wave * per < n ? wave * per : n, hi = lo + per < n ? lo + per : n;
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long *kin = k0, *kout = k1;
    unsigned *vin = v0, *vout = v1;
    for (int shift = 0; shift < 64; shift += 4) {
        int cnt[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d] = 0;
        for (int e0 = lo; e0 < hi; e0 += 64) {
            const int e = e0 + lane;
            const int dig = e < hi ? 15 - (int)((kin[e] >> shift) & 15ull) : -1;       // descending: largest digit first
#pragma unroll
            for (int d = 0; d < 16; ++d) cnt[d] += __popcll(__ballot(dig == d));
        }
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < 16; ++d) hist[d][wave] = cnt[d];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0, uni = -1;
            for (int d = 0; d < 16; ++d) {
                int tot = 0;
                for (int w = 0; w < kSortWaves; ++w) { const int c = hist[d][w]; hist[d][w] = run; run += c; tot += c; }
                if (tot == n) uni = d;
            }
            uniform_digit = uni;
        }
        __syncthreads();
        const bool skip = uniform_digit >= 0;                    // every key has this digit: the pass would be the identity
        if (!skip) {
            int cur[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) cur[d] = hist[d][wave];
            for (int e0 = lo; e0 < hi; e0 += 64) {
                const int e = e0 + lane;
                unsigned long long k = 0; unsigned v = 0;
                if (e < hi) { k = kin[e]; v = vin[e]; }
                const int dig = e < hi ? 15 - (int)((k >> shift) & 15ull) : -1;
                int dst = 0;
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const unsigned long long b = __ballot(dig == d);
                    if (dig == d) dst = cur[d] + __popcll(b & below);
                    cur[d] += __popcll(b);
                }
                if (e < hi) { kout[dst] = k; vout[dst] = v; }
            }
        }
        __syncthreads();                                         // the pass's writes are visible to the whole workgroup; hist is free
        if (!skip) {
            unsigned long long *tk = kin; kin = kout; kout = tk;
            unsigned *tv = vin; vin = vout; vout = tv;
        }
    }
    if (kin != k1)                                               // (uniform) the sorted pairs sit in (k0, v0): copy
        for (int e = threadIdx.x; e < n; e += kSortWaves * 64) { k1[e] = kin[e]; v1[e] = vin[e]; }
}

struct RepWs {
    int *cnt_s, *cnt_m, *poss, *off_s, *off_m, *totals;
    unsigned long long *key_s, *key_m, *key_out;      // candidate lists (both filled in one pass), sorted keys
    unsigned *val_s, *val_m, *val_out;
    size_t total;
};

RepWs rep_layout(char *base, int ns, int max_edges) {
    RepWs w{};
    size_t o = 0;
    auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += balf_align_up(bytes, 256); return p; };
    w.cnt_s = (int *)take((size_t)ns * 4); w.cnt_m = (int *)take((size_t)ns * 4); w.poss = (int *)take((size_t)ns * 4);
    w.off_s = (int *)take((size_t)ns * 4); w.off_m = (int *)take((size_t)ns * 4); w.totals = (int *)take(16);
    w.key_s = (unsigned long long *)take((size_t)max_edges * 8); w.key_m = (unsigned long long *)take((size_t)max_edges * 8);
    w.key_out = (unsigned long long *)take((size_t)max_edges * 8);
    w.val_s = (unsigned *)take((size_t)max_edges * 4); w.val_m = (unsigned *)take((size_t)max_edges * 4);
    w.val_out = (unsigned *)take((size_t)max_edges * 4);
    w.total = o;
    return w;
}

}  // namespace
}  // namespace balf

using namespace balf;

extern "C" size_t balf_repeatability_workspace_bytes(int ns, int nd, int max_edges) {
    if (ns <= 0 || nd <= 0 || max_edges <= 0) return 0;
    return rep_layout(nullptr, ns, max_edges).total;
}

extern "C" int balf_repeatability(const double *src_dev, int ns, const double *dst_dev, int nd, double overlap_err,
                                  double eps, double dist_match_thresh, double radius_size, int max_edges,
                                  int32_t *counts_dev, double *errors_dev, int32_t *corr_s_dev, int32_t *corr_m_dev,
                                  void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!src_dev || !dst_dev || !counts_dev || !errors_dev || !corr_s_dev || !corr_m_dev || !workspace_dev) return BALF_ERR_ARG;
    if (ns <= 0 || nd <= 0 || max_edges <= 0 || ns > kMaxPoints || nd > kMaxPoints) return BALF_ERR_ARG;
    if ((long long)ns * nd > 0x7fffffffLL) return BALF_ERR_SHAPE;
    if (workspace_bytes < balf_repeatability_workspace_bytes(ns, nd, max_edges)) return BALF_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RepWs w = rep_layout(static_cast<char *>(workspace_dev), ns, max_edges);
    const RepParams p{1.0 - overlap_err, eps, dist_match_thresh, radius_size, 4.0 * radius_size};
    rep_count_kernel<<<ns, 256, 0, st>>>(src_dev, ns, dst_dev, nd, p, w.cnt_s, w.cnt_m, w.poss);
    BALF_LAUNCH_CHECK();
    rep_scan_kernel<<<1, 1024, 0, st>>>(w.cnt_s, w.cnt_m, w.poss, ns, w.off_s, w.off_m, w.totals, counts_dev + 2);
    BALF_LAUNCH_CHECK();
    // the list lengths stay on the device (round 5 read them back behind a hipStreamSynchronize): the fill pass cuts a list
    // at max_edges, the sort and assignment kernels read the length themselves
    rep_fill_kernel<<<ns, 64, 0, st>>>(src_dev, ns, dst_dev, nd, p, w.off_s, w.off_m, max_edges, w.key_s, w.val_s, w.key_m, w.val_m);
    BALF_LAUNCH_CHECK();
    const int cap = ns < nd ? ns : nd;
    for (int which = 0; which < 2; ++which) {
        // (sorts between the candidate list and the output buffers; the list is not needed again)
        rep_sort_kernel<<<1, kSortWaves * 64, 0, st>>>(which ? w.key_m : w.key_s, which ? w.val_m : w.val_s, w.key_out, w.val_out,
                                                      w.totals + which, max_edges);
        BALF_LAUNCH_CHECK();
        rep_greedy_kernel<<<1, 64, 0, st>>>(w.key_out, w.val_out, w.totals + which, max_edges, nd, counts_dev + which,
                                            errors_dev + which, which ? corr_m_dev : corr_s_dev, cap);
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}

extern "C" int balf_apply_homography(const double *points_dev, int n, const double *h_dev, double *out_dev, void *stream) {
    if (!points_dev || !h_dev || !out_dev || n <= 0) return BALF_ERR_ARG;
    homography_kernel<<<balf_ceil_div(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(points_dev, n, h_dev, out_dev);
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}


// ------------------------------------------------------------------------------------------------
// create_common_region_masks (/root/reference/balf/benchmark_test/geometry_tools.py:7-26): the part of each image that
// the other image covers.  The reference warps an all-ones image whose 15-pixel frame is zeroed with
// cv2.warpPerspective (default flags: bilinear, constant-zero border), thresholds at 0.75 and zeroes the frame of the
// result.  Restated here from OpenCV's algorithm: the output pixel (x, y) samples the input at M^-1 (x, y, 1), the
// source coordinates are rounded to 1/32 pixel (INTER_TAB_SIZE = 32, round half to even), the four bilinear weights
// are the exact products of those 5-bit fractions.  cv2 is not installed in the build container: parity with it is
// UNPINNED (checked against the oracle's restatement of the same algorithm only).
// ------------------------------------------------------------------------------------------------
namespace {

struct MaskArgs {
    double m[9];          // inverse map, row-major: output pixel -> input coordinates (homogeneous)
    int h_out, w_out;     // mask being produced
    int h_in, w_in;       // the all-ones image being warped
    int border;
    double *out;
};

__device__ __forceinline__ double ones_inner(int y, int x, int h, int w, int b) {
    return (y >= b && y < h - b && x >= b && x < w - b) ? 1.0 : 0.0;      // zero outside the image too
}

// fp contraction is OFF in this kernel and in invert3: every product and sum below is an individually rounded fp64
// operation in source order, so that the CPU oracle (NumPy, no FMA) reproduces the 1/32-pixel rounding bit for bit and the
// two {0,1} masks can be compared for equality.
__global__ __launch_bounds__(256) void common_mask_kernel(MaskArgs a) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)a.h_out * a.w_out) return;
    const int y = (int)(i / a.w_out), x = (int)(i - (long)y * a.w_out);
    double v = 0.0;
    if (y >= a.border && y < a.h_out - a.border && x >= a.border && x < a.w_out - a.border) {
        const double X0 = a.m[0] * x + a.m[1] * y + a.m[2];
        const double Y0 = a.m[3] * x + a.m[4] * y + a.m[5];
        double W = a.m[6] * x + a.m[7] * y + a.m[8];
        W = W != 0.0 ? 32.0 / W : 0.0;
        const double fx = fmax(-2147483648.0, fmin(2147483647.0, X0 * W));
        const double fy = fmax(-2147483648.0, fmin(2147483647.0, Y0 * W));
        const long long X = llrint(fx), Y = llrint(fy);                     // round half to even, like cvRound
        const int sx = (int)(X >> 5), sy = (int)(Y >> 5);
        const double ax = (double)(X & 31) * (1.0 / 32.0), ay = (double)(Y & 31) * (1.0 / 32.0);
        const double s = ones_inner(sy, sx, a.h_in, a.w_in, a.border) * ((1.0 - ax) * (1.0 - ay)) +
                         ones_inner(sy, sx + 1, a.h_in, a.w_in, a.border) * (ax * (1.0 - ay)) +
                         ones_inner(sy + 1, sx, a.h_in, a.w_in, a.border) * ((1.0 - ax) * ay) +
                         ones_inner(sy + 1, sx + 1, a.h_in, a.w_in, a.border) * (ax * ay);
        v = s >= 0.75 ? 1.0 : 0.0;
    }
    a.out[i] = v;
}

// closed-form 3x3 inverse (adjugate / determinant), the form OpenCV's cv::invert takes for n <= 3
bool invert3(const double *m, double *o) {
#pragma clang fp contract(off)
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    if (det == 0.0) return false;
    const double r = 1.0 / det;
    o[0] = A * r; o[1] = -(b * i - c * h) * r; o[2] = (b * f - c * e) * r;
    o[3] = B * r; o[4] = (a * i - c * g) * r;  o[5] = -(a * f - c * d) * r;
    o[6] = C * r; o[7] = -(a * h - b * g) * r; o[8] = (a * e - b * d) * r;
    return true;
}

}  // namespace

extern "C" int balf_common_region_masks(const double *h_dst_2_src_host, int h_src, int w_src, int h_dst, int w_dst,
                                        int border, double *mask_src_dev, double *mask_dst_dev, void *stream) {
    if (!h_dst_2_src_host || !mask_src_dev || !mask_dst_dev) return BALF_ERR_ARG;
    if (h_src <= 0 || w_src <= 0 || h_dst <= 0 || w_dst <= 0 || border < 0) return BALF_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // mask_src = warp(ones_dst, M = h_dst_2_src): samples ones_dst at M^-1 (x, y, 1)
    // mask_dst = warp(ones_src, M = inv(h_dst_2_src) / its [2,2]): samples ones_src at M^-1 = a multiple of h_dst_2_src
    MaskArgs ms{}, md{};
    double inv_h[9];
    if (!invert3(h_dst_2_src_host, ms.m)) return BALF_ERR_ARG;
    for (int k = 0; k < 9; ++k) inv_h[k] = ms.m[k] / ms.m[8];               // the matrix the reference hands to cv2 ...
    if (!invert3(inv_h, md.m)) return BALF_ERR_ARG;                          // ... and cv2 inverts again
    ms.h_out = h_src; ms.w_out = w_src; ms.h_in = h_dst; ms.w_in = w_dst; ms.border = border; ms.out = mask_src_dev;
    md.h_out = h_dst; md.w_out = w_dst; md.h_in = h_src; md.w_in = w_src; md.border = border; md.out = mask_dst_dev;
    common_mask_kernel<<<balf_ceil_div((long)h_src * w_src, 256), 256, 0, st>>>(ms);
    BALF_LAUNCH_CHECK();
    common_mask_kernel<<<balf_ceil_div((long)h_dst * w_dst, 256), 256, 0, st>>>(md);
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}
