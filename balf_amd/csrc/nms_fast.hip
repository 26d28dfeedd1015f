// Greedy "SuperPoint" NMS of the demo path on gfx950.
//
// Replaces get_points_direct_from_score_map + nms_fast (+ soft_argmax_points)
// (/root/reference/balf/utils/test_utils.py:97-215, called from /root/reference/demo/demo_match.py:45-57):
// every pixel with score >= conf_thresh is a candidate; candidates are visited in descending score order and
// one is kept iff no already-kept candidate lies within Chebyshev distance dist_thresh (a (2d+1)^2 window).
//
// The sequential sweep has an exact parallel form (greedy maximal independent set with fixed priorities):
// repeat { every ALIVE candidate that is the maximum of the alive candidates in its window is KEPT;
//          every alive candidate with a newly kept one in its window dies } until nobody is alive.
// Priority = (score, then raster-first) packed into one 64-bit key, so the result is deterministic; the
// reference's order among exactly equal scores is whatever NumPy's unstable argsort produces.
//
// Kernels: greedy_init (crop + border + threshold -> key map), greedy_keep (LDS-tiled separable window max of
// the 64-bit keys), greedy_kill (window OR of the newly-kept flags, alive count per tile; newly kept points are
// appended to the survivor list here), then the top-K / sort kernel of nms_topk.hip and the optional sub-pixel
// soft-argmax.  Tiles without alive candidates are skipped by both passes, so the long tail of rounds is cheap.
#include "common.h"
#include "prof.h"

int balf_topk_select_launch(const int2 *surv, const int *counts, long cap, int B, int K, int zero_fallback,
                            int32_t *idx_dev, float *score_dev, int32_t *count_dev, hipStream_t st,
                            unsigned thr_explicit);

namespace {

typedef unsigned long long u64;
constexpr int GT = 32;            // tile side
constexpr int GD_MAX = 16;        // max dist_thresh
constexpr int GTHREADS = 256;

struct GreedyArgs {
    const float *src;             // [B, Hs, Ws]
    int Hs, Ws, crop_y, crop_x, H, W, border;
    float conf;
    int d;                        // dist_thresh
    u64 *key;                     // [B, H, W]  (score bits << 32 | ~idx) while alive, 0 otherwise
    unsigned *newk;               // [B, H, ceil(W/32)] bit map: kept in the current round
    int2 *surv;                   // [B, cap] kept points (flat index, score bits), appended as they are kept
    int *counts;                  // [B]
    int *alive;                   // [1] alive candidates left (written by the last round of a group)
    int count_alive;
    const int *tile_in;           // [B, tiles]  alive candidates per 32x32 tile before this round
    int *tile_out;                // [B, tiles]  ... after it (written by the kill pass)
};

__device__ __forceinline__ float g_score(const GreedyArgs &a, const float *img, int y, int x) {
    if (y < a.border || y >= a.H - a.border || x < a.border || x >= a.W - a.border) return 0.0f;
    return img[(long)(a.crop_y + y) * a.Ws + (a.crop_x + x)];
}

__global__ __launch_bounds__(GTHREADS) void greedy_init_kernel(GreedyArgs a) {
    const long hw = (long)a.H * a.W;
    const int b = blockIdx.y;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    for (long i = (long)blockIdx.x * GTHREADS + threadIdx.x; i < hw; i += (long)gridDim.x * GTHREADS) {
        const int y = (int)(i / a.W), x = (int)(i - (long)y * a.W);
        const float v = g_score(a, img, y, x);
        a.key[b * hw + i] = (v >= a.conf) ? (((u64)__float_as_uint(v) << 32) | (u64)(0xffffffffu - (unsigned)i)) : 0ull;
    }
}

__device__ __forceinline__ u64 umax64(u64 x, u64 y) { return x > y ? x : y; }

// Window maxima of R consecutive outputs that share most of their inputs: out[j] = max(in[j .. j+w-1]), j = 0..R-1.
// The w-R+1 inputs common to all R windows are reduced once; each output adds a running maximum from the left
// remainder and one from the right remainder: ~(w + 3R) max operations for R outputs instead of R (w - 1).
// Keys are unsigned and 0 means "dead", so 0 is the identity.
template <int R, typename In, typename Out>
__device__ __forceinline__ void window_max_run(int w, In in, Out out) {
    if (w >= R) {
        u64 core = in(R - 1);
        for (int k = R; k < w; ++k) core = umax64(core, in(k));
        u64 left[R], right[R];
        u64 acc = 0ull;
        left[R - 1] = 0ull;
#pragma unroll
        for (int j = R - 2; j >= 0; --j) { acc = umax64(acc, in(j)); left[j] = acc; }
        acc = 0ull;
        right[0] = 0ull;
#pragma unroll
        for (int j = 1; j < R; ++j) { acc = umax64(acc, in(w - 1 + j)); right[j] = acc; }
#pragma unroll
        for (int j = 0; j < R; ++j) out(j, umax64(core, umax64(left[j], right[j])));
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            u64 m = in(j);
            for (int k = 1; k < w; ++k) m = umax64(m, in(j + k));
            out(j, m);
        }
    }
}

// newk is a bit map: word [b][y][tx] holds the "kept in this round" flags of pixels x = 32 tx .. 32 tx + 31 of row y
__global__ __launch_bounds__(GTHREADS) void greedy_keep_kernel(GreedyArgs a) {
    constexpr int S = GT + 2 * GD_MAX;
    __shared__ u64 s_in[S * (S + 1)];
    __shared__ u64 s_row[S * (GT + 1)];
    const int d = a.d, side = GT + 2 * d, w = 2 * d + 1;
    const int b = blockIdx.z, ty0 = blockIdx.y * GT, tx0 = blockIdx.x * GT;
    const long hw = (long)a.H * a.W;
    const u64 *key = a.key + b * hw;
    unsigned *bits = a.newk + ((long)b * a.H) * gridDim.x + blockIdx.x;       // + y * gridDim.x
    // a tile without alive candidates keeps nobody: after the first rounds that is almost every tile, and the long
    // tail of rounds (a dozen on real score maps) costs a few bytes per tile instead of a window max
    if (a.tile_in[((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] == 0) {
        if (threadIdx.x < GT && ty0 + threadIdx.x < a.H) bits[(long)(ty0 + threadIdx.x) * gridDim.x] = 0u;
        return;
    }
    for (int i = threadIdx.x; i < side * side; i += GTHREADS) {
        const int r = i / side, c = i - r * side;
        const int y = ty0 - d + r, x = tx0 - d + c;
        s_in[r * (S + 1) + c] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? key[(long)y * a.W + x] : 0ull;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < side * (GT / 8); i += GTHREADS) {        // row pass: 8 outputs per item
        const int r = i / (GT / 8), c0 = (i - r * (GT / 8)) * 8;
        const u64 *row = s_in + r * (S + 1) + c0;
        window_max_run<8>(w, [&](int k) { return row[k]; }, [&](int j, u64 v) { s_row[r * (GT + 1) + c0 + j] = v; });
    }
    __syncthreads();
    {                                                                       // column pass: 4 outputs per item
        const int c = threadIdx.x & (GT - 1), r0 = (threadIdx.x >> 5) * 4;  // 32 columns x 8 row groups of 4
        const u64 *col = s_row + r0 * (GT + 1) + c;
        unsigned keep4 = 0;
        window_max_run<4>(w, [&](int k) { return col[k * (GT + 1)]; }, [&](int j, u64 m) {
            const u64 own = s_in[(r0 + j + d) * (S + 1) + c + d];
            if (own != 0ull && own == m) keep4 |= 1u << j;
        });
        // one word per row: bit c of row r0 + j = lane's flag j; the 32 lanes of a row group sit in one half wave
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long bal = __ballot((keep4 >> j) & 1u);
            const unsigned word = (unsigned)((threadIdx.x & 32) ? (bal >> 32) : bal);
            const int y = ty0 + r0 + j;
            if (c == 0 && y < a.H) bits[(long)y * gridDim.x] = word;
        }
    }
}

__global__ __launch_bounds__(GTHREADS) void greedy_kill_kernel(GreedyArgs a) {
    constexpr int S = GT + 2 * GD_MAX;
    __shared__ u64 s_h[S];                          // per halo row: horizontal window-OR of the kept flags
    __shared__ u64 s_own[S];
    __shared__ int s_cnt;
    const int d = a.d, side = GT + 2 * d, w = 2 * d + 1;
    const int b = blockIdx.z, ty0 = blockIdx.y * GT, tx0 = blockIdx.x * GT;
    const long hw = (long)a.H * a.W;
    const long tile = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (a.tile_in[tile] == 0) {                       // nobody alive here: nothing to kill, nothing newly kept
        if (threadIdx.x == 0) a.tile_out[tile] = 0;
        return;
    }
    if (threadIdx.x == 0) s_cnt = 0;
    if (threadIdx.x < side) {
        // bits x = tx0 - d .. tx0 + 31 + d of halo row r as one 64-bit word (bit k = column tx0 - d + k)
        const int r = threadIdx.x, y = ty0 - d + r;
        u64 m = 0ull;
        if (y >= 0 && y < a.H) {
            const unsigned *row = a.newk + ((long)b * a.H + y) * gridDim.x;
            const int tx = blockIdx.x;
            const u64 mid = row[tx];
            const u64 lo = tx > 0 ? row[tx - 1] : 0u, hi = tx + 1 < (int)gridDim.x ? row[tx + 1] : 0u;
            m = (d > 0 ? (lo >> (32 - d)) : 0ull) | (mid << d) | (d > 0 ? (hi << (32 + d)) : 0ull);
        }
        s_own[r] = m;
        u64 t = m;                                    // horizontal OR over a window of w bits by doubling
        int have = 1;
        while (2 * have <= w) { t |= t >> have; have *= 2; }
        s_h[r] = t | (t >> (w - have));               // bit k = OR of columns k .. k + w - 1
    }
    __syncthreads();
    int alive = 0;
    for (int i = threadIdx.x; i < GT * GT; i += GTHREADS) {
        const int r = i / GT, c = i - r * GT;
        const int y = ty0 + r, x = tx0 + c;
        if (y >= a.H || x >= a.W) continue;
        const long p = b * hw + (long)y * a.W + x;
        const u64 k = a.key[p];
        if (k == 0ull) continue;
        u64 v = 0ull;                                 // vertical OR of the rows r .. r + 2d (wave-uniform per row)
        for (int q = 0; q < w; ++q) v |= s_h[r + q];
        if ((s_own[r + d] >> (c + d)) & 1ull) {       // newly kept: straight onto the survivor list
            const int pos = atomicAdd(&a.counts[b], 1);
            a.surv[b * hw + pos] = make_int2(y * a.W + x, (int)(k >> 32));
        }
        if ((v >> c) & 1ull) a.key[p] = 0ull;         // newly kept itself, or suppressed by a newly kept neighbour
        else ++alive;
    }
    if (alive) atomicAdd(&s_cnt, alive);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_out[tile] = s_cnt;
        if (a.count_alive && s_cnt) atomicAdd(a.alive, s_cnt);
    }
}

// soft_argmax_points (test_utils.py:170-215) on the selected points: the patch is normalised by its sum + 1e-6,
// log-ed and soft-max-ed, which is the patch itself re-normalised; the expected (x, y) inside the patch replaces
// the integer position: p += E[pos] - patch//2.  (torchgeometry's SpatialSoftArgmax2d is not installed in the
// build container, so this follows its documented definition: "parity unpinned".)
__global__ void subpixel_kernel(GreedyArgs a, const int32_t *idx, const int32_t *count, int K, int patch, float *xy) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count[b] < K ? count[b] : K;
    if (i >= K) return;
    float *o = xy + ((long)b * K + i) * 2;
    if (i >= n) { o[0] = 0.0f; o[1] = 0.0f; return; }
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    const int p = idx[(long)b * K + i];
    const int y = p / a.W, x = p - y * a.W;
    const int pad = patch / 2;
    float s = 0.0f, sx = 0.0f, sy = 0.0f;
    for (int dy = 0; dy < patch; ++dy)
        for (int dx = 0; dx < patch; ++dx) {
            const int yy = y - pad + dy, xx = x - pad + dx;
            const float v = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? g_score(a, img, yy, xx) : 0.0f;
            s += v; sx += v * dx; sy += v * dy;
        }
    o[0] = (float)x + sx / s - (float)pad;
    o[1] = (float)y + sy / s - (float)pad;
}

}  // namespace

extern "C" size_t balf_greedy_nms_workspace_bytes(int B, int H, int W, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    const size_t px = (size_t)B * H * W;
    const size_t tiles = (size_t)B * balf_ceil_div(W, GT) * balf_ceil_div(H, GT);
    return balf_align_up(px * 8, 256) + balf_align_up((size_t)B * H * balf_ceil_div(W, GT) * 4, 256) + 256 /*alive*/ +
           balf_align_up((size_t)B * sizeof(int), 256) + 2 * balf_align_up(tiles * sizeof(int), 256) + px * sizeof(int2);
}

extern "C" int balf_greedy_nms(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                               int border, float conf_thresh, int dist_thresh, int K, int subpixel_patch,
                               int32_t *idx_dev, float *score_dev, float *xy_dev, int32_t *count_dev,
                               int32_t *total_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!prob_dev || !idx_dev || !score_dev || !count_dev || !workspace_dev) return BALF_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || K > BALF_MAX_TOPK || border < 0) return BALF_ERR_ARG;
    if (!(conf_thresh > 0.0f) || dist_thresh < 0 || dist_thresh > GD_MAX) return BALF_ERR_ARG;
    if (subpixel_patch < 0 || subpixel_patch > 16 || (subpixel_patch > 0 && !xy_dev)) return BALF_ERR_ARG;
    if (crop_y < 0 || crop_x < 0 || crop_y + H > Hp || crop_x + W > Wp) return BALF_ERR_SHAPE;
    if ((long)H * W > 0x7fffffffL) return BALF_ERR_SHAPE;
    if (workspace_bytes < balf_greedy_nms_workspace_bytes(B, H, W, K)) return BALF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t px = (size_t)B * H * W;
    char *w = static_cast<char *>(workspace_dev);
    u64 *key = reinterpret_cast<u64 *>(w); w += balf_align_up(px * 8, 256);
    unsigned *newk = reinterpret_cast<unsigned *>(w); w += balf_align_up((size_t)B * H * balf_ceil_div(W, GT) * 4, 256);
    int *alive = reinterpret_cast<int *>(w); w += 256;
    int *counts = reinterpret_cast<int *>(w); w += balf_align_up((size_t)B * sizeof(int), 256);
    const size_t n_tiles = (size_t)B * balf_ceil_div(W, GT) * balf_ceil_div(H, GT);
    int *tile_a = reinterpret_cast<int *>(w); w += balf_align_up(n_tiles * sizeof(int), 256);
    int *tile_b = reinterpret_cast<int *>(w); w += balf_align_up(n_tiles * sizeof(int), 256);
    int2 *surv = reinterpret_cast<int2 *>(w);

    GreedyArgs a{prob_dev, Hp, Wp, crop_y, crop_x, H, W, border, conf_thresh, dist_thresh, key, newk, surv, counts,
                 alive, 0, tile_a, tile_b};
    if (hipMemsetAsync(counts, 0, balf_align_up((size_t)B * sizeof(int), 256), st) != hipSuccess) return BALF_ERR_LAUNCH;
    if (hipMemsetAsync(tile_a, 1, n_tiles * sizeof(int), st) != hipSuccess) return BALF_ERR_LAUNCH;   // every tile may be alive
    const dim3 lin((unsigned)balf_ceil_div((long)H * W, GTHREADS * 4), B), tiles(balf_ceil_div(W, GT), balf_ceil_div(H, GT), B);
    hipLaunchKernelGGL(greedy_init_kernel, lin, dim3(GTHREADS), 0, st, a);
    BALF_LAUNCH_CHECK();
    // rounds in groups of 4; the last kill of a group counts the candidates still alive, read back with ONE
    // stream synchronisation per group (the only entry point of the library that synchronises)
    for (int round = 0;; round += 4) {
        if (round > 4096) return BALF_ERR_LAUNCH;      // cannot happen: every round keeps >= 1 per alive region
        if (hipMemsetAsync(alive, 0, sizeof(int), st) != hipSuccess) return BALF_ERR_LAUNCH;
        for (int r = 0; r < 4; ++r) {
            a.count_alive = (r == 3);
            BALF_PROF(balf_prof::kGreedyKeep, st, hipLaunchKernelGGL(greedy_keep_kernel, tiles, dim3(GTHREADS), 0, st, a));
            BALF_PROF(balf_prof::kGreedyKill, st, hipLaunchKernelGGL(greedy_kill_kernel, tiles, dim3(GTHREADS), 0, st, a));
            int *t = const_cast<int *>(a.tile_in); a.tile_in = a.tile_out; a.tile_out = t;   // ping-pong the tile counts
        }
        BALF_LAUNCH_CHECK();
        int h_alive = 0;
        if (hipMemcpyAsync(&h_alive, alive, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return BALF_ERR_LAUNCH;
        if (h_alive == 0) break;
    }
    if (total_dev && hipMemcpyAsync(total_dev, counts, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return BALF_ERR_LAUNCH;
    // the K best kept points by score, sorted (score desc, index asc); count_dev = rows returned (<= K)
    int rc = balf_topk_select_launch(surv, counts, (long)H * W, B, K, /*zero_fallback=*/0, idx_dev, score_dev,
                                     count_dev, st, /*thr_explicit=*/0u);
    if (rc != BALF_OK) return rc;
    if (subpixel_patch > 0) {
        hipLaunchKernelGGL(subpixel_kernel, dim3(balf_ceil_div(K, 256), B), dim3(256), 0, st, a, idx_dev, count_dev, K,
                           subpixel_patch, xy_dev);
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}
