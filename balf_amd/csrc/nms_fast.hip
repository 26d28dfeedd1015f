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
// Round-6 form: STREAM-ORDERED (no hipStreamSynchronize, no read-back) with one bit per pixel of state.
//   * state in HBM: an `alive` bit map and a `kept in this round` bit map ([B][H][ceil(W/64)] 64-bit words), per 64 x 32
//     tile the round after which it holds no candidate (`dead_round`), per image two ping-pong lists of the tiles that
//     still hold candidates.  The score map is read-only; the 64-bit keys exist in LDS only.
//   * a round = two launches over each image's list of live tiles:
//       keep -- tile + d halo of keys in LDS, separable (2d+1)^2 window maximum with one v_max_f64 per comparison (a key is
//               a positive normal double: (score bits + 2^20) << 32 | ~index orders exactly like the unsigned integer);
//               where the halo region holds <= 512 candidates (every round but the first on a real score map) a pairwise
//               test of the enumerated candidates instead.  Round 1 takes its candidates straight from the score map
//               (crop, border, threshold in the load) and writes the alive map.
//       kill -- window OR of the kept bits as shifts of one 128-bit row word, alive &= ~that, kept points appended to the
//               survivor list, the tile to the next round's list (or dead_round = round).
//     A neighbour tile that was already dead when the round began is never read (dead_round < round): its words are stale.
//     A pass is SKIPPED for a tile with nothing to do: keep when no alive bit of its 3 x 3 tile neighbourhood changed in the
//     previous round (chg_round), kill when no point of that neighbourhood was kept in this one (kept_round) -- see nbr_round_is.
//   * the number of rounds is data dependent (4-8 on score maps, ~W/d on a monotone ramp).  `rounds_launched` rounds are
//     always enqueued (12); a launch whose image list is empty returns at once.  What is still alive after them is finished by
//     greedy_tail_kernel: ONE workgroup per image looping rounds over its own list until the list is empty -- images are
//     independent, so no workgroup ever waits for another one and every wave reaches its exit.  The adversarial inputs (a
//     1080p monotone ramp or constant plateau: ~120-190 rounds with a wavefront of kept points) take 0.2 s there (1.6 s
//     without the skip, BALF_GREEDY_NOSKIP=1) instead of 1.3 ms, exact like everything else (tools/greedy_ramp_probe.py).
// Then the top-K / sort kernel of nms_topk.hip and the optional sub-pixel soft-argmax.
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "prof.h"

int balf_topk_select_launch(const int2 *surv, const int *counts, long cap, int B, int K, int zero_fallback,
                            int32_t *idx_dev, float *score_dev, int32_t *count_dev, hipStream_t st,
                            unsigned thr_explicit);

namespace {

typedef unsigned long long u64;
typedef unsigned __int128 u128;
constexpr int TW = 64, TH = 32;              // tile: one 64-bit word wide
constexpr int GD_MAX = 16;                   // max dist_thresh (<= TH, <= TW: a halo reaches the adjacent tiles only)
constexpr int KTHREADS = 256;                // keep kernel, tail kernel
constexpr int LTHREADS = 64;                 // kill kernel: one wave per tile
constexpr int RS = TW + 2 * GD_MAX + 1;      // LDS row stride in keys: 97 = 194 dwords = 2 mod 64 -> a column of keys is conflict-free
constexpr int RH_MAX = TH + 2 * GD_MAX;      // 64 region rows
constexpr int NS = 512;                      // pairwise mode up to this many candidates in the halo region
constexpr int ROUNDS_DEFAULT = 12;           // rounds enqueued as launches of their own before the per-image tail (score maps need 7-9)
constexpr unsigned KEY_BIAS = 0x00100000u;   // keeps a key's high word out of the double's denormal range
constexpr int ALIVE_FOREVER = 0x7f7f7f7f;    // dead_round of a tile with candidates

struct GreedyArgs {
    const float *src;             // [B, Hs, Ws]
    int Hs, Ws, crop_y, crop_x, H, W, border;
    float conf;
    int d;                        // dist_thresh
    int B, ntx, nty;              // tiles per row / column; 64-bit words per bit-map row = ntx
    int noskip;                   // BALF_GREEDY_NOSKIP=1 (A/B switch, development aid): every pass visits every listed tile
    u64 *alive;                   // [B, H, ntx]
    u64 *kept;                    // [B, H, ntx] kept in the current round
    int *dead_round;              // [B, nty * ntx]
    int *chg_round;               // [B, nty * ntx] last round in which the kill pass changed the tile's alive bits (0: never)
    int *kept_round;              // [B, nty * ntx] last round in which the keep pass kept a point of the tile (0: never)
    int *list;                    // [2, B, nty * ntx] live tiles: list[r & 1] is read by round r, written by round r - 1
    int *dlist;                   // [B, nty * ntx] live tiles of the current round with > NS candidates around them (window mode)
    int *ctr;                     // counters, one per 256-byte line (CTR_STRIDE ints): survivors [B], list lengths [2, B], dlist length [B]
    int2 *surv;                   // [B, H * W] kept points (flat index, score bits), appended as they are kept
    int *counts;                  // [B] survivors per image, contiguous (written by the tail kernel for the top-K kernel)
    int32_t *total;               // [B] the caller's copy of it (may be null)
};
// every counter is hit by one atomic per tile and round: each gets a cache line (and so an L2 channel) of its own -- with
// the 32 images' counters in one 128-byte line the first kill pass spent 0.4 ms queueing on that line
constexpr int CTR_STRIDE = 64;
__device__ __forceinline__ int *ctr_surv(const GreedyArgs &a, int b) { return a.ctr + (long)b * CTR_STRIDE; }
__device__ __forceinline__ int *ctr_list(const GreedyArgs &a, int parity, int b) { return a.ctr + ((long)(1 + parity) * a.B + b) * CTR_STRIDE; }
__device__ __forceinline__ int *ctr_dlist(const GreedyArgs &a, int b) { return a.ctr + ((long)3 * a.B + b) * CTR_STRIDE; }

__device__ __forceinline__ float g_score(const GreedyArgs &a, const float *img, int y, int x) {
    if (y < a.border || y >= a.H - a.border || x < a.border || x >= a.W - a.border) return 0.0f;
    return img[(long)(a.crop_y + y) * a.Ws + (a.crop_x + x)];
}

__device__ __forceinline__ double make_key(float v, int idx) {
    const u64 k = ((u64)(__float_as_uint(v) + KEY_BIAS) << 32) | (u64)(0xffffffffu - (unsigned)idx);
    return __longlong_as_double((long long)k);
}
// maximum of two keys (positive normal doubles or +0) = the unsigned maximum of their bit patterns, in one instruction
__device__ __forceinline__ double kmax(double x, double y) {
    asm("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));      // ("+v": the repo-wide rule for inline-asm VALU, test_build_invariants.py)
    return x;
}

// workgroup barrier that waits for the LDS traffic only: the score values requested for the next tile stay in flight
// (__syncthreads() drains vmcnt too).  One asm statement with a memory clobber, so nothing moves across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Window maxima of R consecutive outputs that share most of their inputs: out[j] = max(in[j .. j+w-1]), j = 0..R-1.
// The w-R+1 inputs common to all R windows are reduced once; each output adds a running maximum from the left
// remainder and one from the right remainder: ~(w + 3R) max operations for R outputs instead of R (w - 1).
// +0 means "dead" and is the identity.
template <int R, int WC = 0, typename In, typename Out>
__device__ __forceinline__ void window_max_run(int w_runtime, In in, Out out) {
    const int w = WC > 0 ? WC : w_runtime;           // WC > 0: the window is a compile-time constant (every loop below unrolls)
    if (w >= R) {
        // the reads of a batch are independent (issued together, one wait): with two waves per SIMD a read-max-read chain
        // of 30 LDS round trips per item was the whole kernel
        double left[R], right[R], t[8];
#pragma unroll
        for (int j = 0; j < R - 1; ++j) left[j] = in(j);
#pragma unroll
        for (int j = 1; j < R; ++j) right[j] = in(w - 1 + j);
        double core = in(R - 1);
        if constexpr (WC > 0) {                       // compile-time window: straight-line code, immediate LDS offsets
#pragma unroll
            for (int k0 = R; k0 < WC; k0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = k0 + u < WC ? in(k0 + u) : 0.0;
                core = kmax(core, kmax(kmax(kmax(t[0], t[1]), kmax(t[2], t[3])), kmax(kmax(t[4], t[5]), kmax(t[6], t[7]))));
            }
        } else {
            int k = R;
            for (; k + 8 <= w; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = in(k + u);
                core = kmax(core, kmax(kmax(kmax(t[0], t[1]), kmax(t[2], t[3])), kmax(kmax(t[4], t[5]), kmax(t[6], t[7]))));
            }
            for (; k < w; ++k) core = kmax(core, in(k));
        }
        double acc = 0.0;
#pragma unroll
        for (int j = R - 2; j >= 0; --j) { acc = kmax(acc, left[j]); left[j] = acc; }
        left[R - 1] = 0.0;
        acc = 0.0;
#pragma unroll
        for (int j = 1; j < R; ++j) { acc = kmax(acc, right[j]); right[j] = acc; }
        right[0] = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) out(j, kmax(core, kmax(left[j], right[j])));
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double m = in(j);
            for (int k = 1; k < w; ++k) m = kmax(m, in(j + k));
            out(j, m);
        }
    }
}

struct KeepLds {
    double in[RH_MAX * RS];        // region keys (window mode); the candidate keys of the pairwise mode alias it
    double col[TH * RS];           // vertical window maxima; the candidate positions of the pairwise mode alias it
    u64 words[RH_MAX * 3];         // alive words of the region rows: left neighbour / own / right neighbour tile column, masked to the region
    unsigned kbits[TH * 2];        // kept bits of the tile's rows
    unsigned abits[TH * 2];        // alive bits of the tile's rows (round 1)
    int n, m, pad_[2];
};
struct KillLds {
    u64 h[RH_MAX];                 // per region row: horizontal window OR of the kept bits, bit c = tile column c
    int dead[12];                  // the 3 x 3 tile neighbourhood: dead when the round began
};
struct SparseLds {                 // pairwise keep pass, one wave per tile
    double key[NS];
    int2 pos[NS];
    u64 words[RH_MAX * 3];
    unsigned kbits[TH * 2];
    int dead[12];
    int n, m;
};

// did the tile (tyi, txi) of image b hold no candidate when `round` began?  (tiles outside the image: yes)
__device__ __forceinline__ bool tile_dead(const GreedyArgs &a, int b, int tyi, int txi, int round) {
    if (tyi < 0 || tyi >= a.nty || txi < 0 || txi >= a.ntx) return true;
    return a.dead_round[((long)b * a.nty + tyi) * a.ntx + txi] < round;
}

// Activity of a tile's 3 x 3 tile neighbourhood (a halo reaches the adjacent tiles only): did any of them record `value` in `arr`?
// keep pass of round r: chg_round == r - 1 -- if no alive bit of the neighbourhood changed in the previous round, the tile's
// candidates and everything around them are what they were when the tile last kept nothing (had it kept a point, its own bits
// would have changed), so it keeps nothing now either and its kept words are still zero: the pass is skipped.  kill pass of
// round r: kept_round == r -- no newly kept point in reach, nothing dies.  On a monotone ramp or a plateau (one wavefront of
// kept points moving one window per round, ~W/d rounds) all but the tiles on the front skip.  Every thread evaluates it alike.
__device__ __forceinline__ bool nbr_round_is(const GreedyArgs &a, const int *arr, int b, int tyi, int txi, int value) {
    bool hit = a.noskip != 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = tyi - 1 + k / 3, x = txi - 1 + k % 3;
        if (y >= 0 && y < a.nty && x >= 0 && x < a.ntx) hit |= arr[((long)b * a.nty + y) * a.ntx + x] == value;
    }
    return hit;
}

// the same for a workgroup that is ONE wave (pairwise keep, kill kernels): lanes 0..8 look at one neighbour each
__device__ __forceinline__ bool nbr_round_is_wave(const GreedyArgs &a, const int *arr, int b, int tyi, int txi, int value) {
    const int lane = threadIdx.x;
    bool hit = a.noskip != 0;
    if (lane < 9) {
        const int y = tyi - 1 + lane / 3, x = txi - 1 + lane % 3;
        if (y >= 0 && y < a.nty && x >= 0 && x < a.ntx) hit = arr[((long)b * a.nty + y) * a.ntx + x] == value;
    }
    return __ballot(hit) != 0ull;
}

// The two window passes over the keys in s.in (tile + d halo) and the kept / alive bits of the tile's rows into s.kbits /
// s.abits (zeroed by the caller before the barrier this begins with).
template <bool FIRST, int DC = -1>
__device__ __forceinline__ void window_passes(const GreedyArgs &a, KeepLds &s, int tyi, int txi) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int d = DC >= 0 ? DC : a.d, w = 2 * d + 1, RW = TW + 2 * d;      // DC >= 0: dist_thresh known at compile time
    constexpr int WC = DC >= 0 ? 2 * DC + 1 : 0;
    lds_barrier();
    for (int i = tid; i < RW * (TH / 8); i += nthr) {                      // column pass: 8 outputs per item, lanes along x
        const int blk = i / RW, c = i - blk * RW, r0 = blk * 8;
        const double *col = s.in + r0 * RS + c;
        window_max_run<8, WC>(w, [&](int k) { return col[k * RS]; }, [&](int j, double v) { s.col[(r0 + j) * RS + c] = v; });
    }
    lds_barrier();
    for (int i = tid; i < TH * (TW / 8); i += nthr) {                      // row pass: 8 outputs per item, lanes along y
        const int r = i & (TH - 1), blk = i / TH, c0 = blk * 8;
        const double *row = s.col + r * RS + c0;
        unsigned keep8 = 0, alive8 = 0;
        window_max_run<8, WC>(w, [&](int k) { return row[k]; }, [&](int j, double m) {
            const double own = s.in[(r + d) * RS + c0 + j + d];
            if (own != 0.0) {
                alive8 |= 1u << j;
                if (own == m) keep8 |= 1u << j;
            }
        });
        if (keep8) atomicOr(&s.kbits[r * 2 + (blk >> 2)], keep8 << ((blk & 3) * 8));
        if (FIRST && alive8) atomicOr(&s.abits[r * 2 + (blk >> 2)], alive8 << ((blk & 3) * 8));
    }
}

// Keep pass of one tile in rounds >= 2 (window-mode kernel and tail): writes the tile's words of the `kept` map.  All
// threads of the workgroup take part (blockDim.x a multiple of 64).
__device__ void keep_tile(const GreedyArgs &a, KeepLds &s, int b, int tile, int round, bool check_activity) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
    if (check_activity && !nbr_round_is(a, a.chg_round, b, tyi, txi, round - 1)) return;      // uniform
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int d = a.d, RH = TH + 2 * d, RW = TW + 2 * d;
    const int ry0 = ty0 - d, rx0 = tx0 - d;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    __syncthreads();                                  // the workgroup's previous tile is done with the LDS
    if (tid < TH * 2) s.kbits[tid] = 0u;
    if (tid == 0) { s.n = 0; s.m = 0; }
    bool window_mode;
    {
        for (int i = tid; i < RH * 3; i += nthr) {
            const int r = i / 3, j = i - r * 3, y = ry0 + r, wx = txi - 1 + j;
            u64 v = 0ull;
            if (y >= 0 && y < a.H && !tile_dead(a, b, y / TH, wx, round)) {
                v = a.alive[((long)b * a.H + y) * a.ntx + wx];
                if (j == 0) v = d ? (v >> (64 - d)) << (64 - d) : 0ull;         // columns tx0-d .. tx0-1
                if (j == 2) v = d ? (v & ((1ull << d) - 1ull)) : 0ull;          // columns tx0+64 .. tx0+63+d
            }
            s.words[i] = v;
        }
        __syncthreads();
        int c = 0;
        for (int i = tid; i < RH * 3; i += nthr) c += __popcll(s.words[i]);
        if (c) atomicAdd(&s.n, c);
        __syncthreads();
        window_mode = s.n > NS;
    }
    if (window_mode) {
        for (int i = tid; i < RH * RW; i += nthr) {
            const int r = i / RW, c = i - r * RW;
            const int bit = c + 64 - d;                                         // bit 0 of word 0 = column tx0 - 64
            double k = 0.0;
            if ((s.words[r * 3 + (bit >> 6)] >> (bit & 63)) & 1ull) k = make_key(g_score(a, img, ry0 + r, rx0 + c), (ry0 + r) * a.W + rx0 + c);
            s.in[r * RS + c] = k;
        }
        window_passes<false>(a, s, tyi, txi);
    } else {
        // pairwise mode: enumerate the region's candidates (position, key); a candidate of the tile is kept iff no
        // candidate with a higher key lies within Chebyshev distance d -- all of those are inside the region
        int2 *ex = reinterpret_cast<int2 *>(s.col);
        double *ek = s.in;
        for (int i = tid; i < RH * 3; i += nthr) {
            u64 v = s.words[i];
            if (!v) continue;
            const int r = i / 3, j = i - r * 3, y = ry0 + r;
            int e = atomicAdd(&s.m, __popcll(v));
            while (v) {
                const int bit = __builtin_ctzll(v);
                v &= v - 1ull;
                const int x = tx0 + (j - 1) * 64 + bit;
                ex[e] = make_int2(x, y);
                ek[e] = make_key(g_score(a, img, y, x), y * a.W + x);
                ++e;
            }
        }
        __syncthreads();
        const int n = s.n;
        for (int e = tid; e < n; e += nthr) {
            const int2 p = ex[e];
            if (p.x < tx0 || p.x >= tx0 + TW || p.y < ty0 || p.y >= ty0 + TH) continue;
            const double k = ek[e];
            bool keep = true;
            for (int q = 0; q < n; ++q) {
                const int2 o = ex[q];
                if (ek[q] > k && abs(o.x - p.x) <= d && abs(o.y - p.y) <= d) { keep = false; break; }
            }
            if (keep) atomicOr(&s.kbits[(p.y - ty0) * 2 + ((p.x - tx0) >> 5)], 1u << ((p.x - tx0) & 31));
        }
    }
    __syncthreads();
    if (tid < 64) {                                    // wave 0: the tile's 32 row words
        u64 kw = 0ull;
        if (tid < TH && ty0 + tid < a.H) {
            kw = (u64)s.kbits[tid * 2] | ((u64)s.kbits[tid * 2 + 1] << 32);
            a.kept[((long)b * a.H + ty0 + tid) * a.ntx + txi] = kw;
        }
        if (__ballot(kw != 0ull) != 0ull && tid == 0) a.kept_round[(long)b * a.nty * a.ntx + tile] = round;
    }
}

// wave-wide helpers (64 lanes, all active)
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// Kill pass of one tile.  Wave 0 of the workgroup does the work; every thread must call it (two barriers).
__device__ void kill_tile(const GreedyArgs &a, KillLds &s, int b, int tile, int round, bool check_activity) {
    const int tid = threadIdx.x;
    const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
    if (check_activity && !nbr_round_is(a, a.kept_round, b, tyi, txi, round)) {       // uniform: no newly kept point in reach
        if (tid == 0) {                                                                // -> the tile lives on unchanged
            const int nxt = (round + 1) & 1;
            a.list[((long)nxt * a.B + b) * a.nty * a.ntx + atomicAdd(ctr_list(a, nxt, b), 1)] = tile;
        }
        return;
    }
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int d = a.d, w = 2 * d + 1, RH = TH + 2 * d;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    // every global load of the pass is issued before anything waits: dead flags, the kept words of the region rows (stale
    // words of dead tiles are read and masked afterwards), the tile's own alive and kept words
    u64 L = 0ull, M = 0ull, R = 0ull, al = 0ull, kp = 0ull;
    int tr = 0;
    __syncthreads();                                  // the previous tile of this workgroup is done with the LDS
    if (tid < 64) {
        if (tid < 9) s.dead[tid] = tile_dead(a, b, tyi - 1 + tid / 3, txi - 1 + tid % 3, round);
        if (tid < RH) {
            const int y = ty0 - d + tid;
            if (y >= 0 && y < a.H) {
                tr = y / TH - (tyi - 1);
                const u64 *row = a.kept + ((long)b * a.H + y) * a.ntx;
                if (txi > 0) L = row[txi - 1];
                M = row[txi];
                if (txi + 1 < a.ntx) R = row[txi + 1];
            }
        }
        if (tid < TH && ty0 + tid < a.H) {
            const long wi = ((long)b * a.H + ty0 + tid) * a.ntx + txi;
            al = a.alive[wi];
            kp = a.kept[wi];
        }
    }
    __syncthreads();
    if (tid < RH) {
        if (s.dead[tr * 3 + 0]) L = 0ull;
        if (s.dead[tr * 3 + 1]) M = 0ull;
        if (s.dead[tr * 3 + 2]) R = 0ull;
        // 128 bits of the row, bit k = column tx0 - 16 + k
        u128 t = ((u128)((M >> 48) | (R << 16)) << 64) | (u128)((L >> 48) | (M << 16));
        int have = 1;                                 // OR over a window of w bits by doubling
        while (2 * have <= w) { t |= t >> have; have *= 2; }
        t |= t >> (w - have);                         // bit k = OR of bits k .. k + w - 1
        s.h[tid] = (u64)(t >> (16 - d));              // bit c = OR of the columns tx0 + c - d .. tx0 + c + d
    }
    __syncthreads();
    if (tid >= 64) return;
    u64 v = 0ull;
    if (tid < TH)
        for (int q = 0; q < w; ++q) v |= s.h[tid + q];
    // newly kept points onto the survivor list: one atomic per tile
    const int nk = __popcll(kp);
    const int incl = wave_incl_scan(nk);
    const int total = __shfl(incl, 63);
    if (total) {
        int base = 0;
        if (tid == 0) base = atomicAdd(ctr_surv(a, b), total);
        base = __shfl(base, 0);
        int2 *out = a.surv + (long)b * a.H * a.W + base + incl - nk;
        const int y = ty0 + tid;
        while (kp) {
            const int x = tx0 + __builtin_ctzll(kp);
            kp &= kp - 1ull;
            *out++ = make_int2(y * a.W + x, (int)__float_as_uint(g_score(a, img, y, x)));
        }
    }
    const u64 na = al & ~v;                            // the kept ones themselves and whoever they suppress
    if (na != al) a.alive[((long)b * a.H + ty0 + tid) * a.ntx + txi] = na;
    const int alive = wave_sum(__popcll(na));
    const bool changed = __ballot(na != al) != 0ull;
    if (tid == 0) {
        if (changed) a.chg_round[(long)b * a.nty * a.ntx + tile] = round;
        if (alive == 0) {
            a.dead_round[(long)b * a.nty * a.ntx + tile] = round;
        } else {
            const int nxt = (round + 1) & 1;
            const int pos = atomicAdd(ctr_list(a, nxt, b), 1);
            a.list[((long)nxt * a.B + b) * a.nty * a.ntx + pos] = tile;
        }
    }
}

// Keep pass of rounds >= 2 where the tile's halo region holds <= NS candidates (every tile of a real score map): ONE WAVE
// per tile and a few KB of LDS, so that a CU keeps dozens of tiles in flight.  Tiles with more candidates go to the
// window-mode list.  The tile's own candidates are enumerated first (entries 0 .. n_tile - 1): the pairwise loop runs on
// full lanes.
__device__ void keep_tile_sparse(const GreedyArgs &a, SparseLds &s, int b, int tile, int round) {
    const int lane = threadIdx.x;
    const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const int d = a.d, RH = TH + 2 * d, ry0 = ty0 - d;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    if (!nbr_round_is_wave(a, a.chg_round, b, tyi, txi, round - 1)) return;      // nothing around the tile changed: see nbr_round_is
    __syncthreads();
    u64 wv[3] = {0ull, 0ull, 0ull};                   // words lane, lane + 64, lane + 128 of the RH x 3 region words
    if (lane < 9) s.dead[lane] = tile_dead(a, b, tyi - 1 + lane / 3, txi - 1 + lane % 3, round);
    if (lane < TH * 2) s.kbits[lane] = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = lane + 64 * k;
        if (i < RH * 3) {
            const int r = i / 3, j = i - r * 3, y = ry0 + r, wx = txi - 1 + j;
            if (y >= 0 && y < a.H && wx >= 0 && wx < a.ntx) wv[k] = a.alive[((long)b * a.H + y) * a.ntx + wx];
        }
    }
    __syncthreads();
    int c = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = lane + 64 * k;
        if (i < RH * 3) {
            const int r = i / 3, j = i - r * 3, y = ry0 + r;
            u64 v = wv[k];
            if (y < 0 || y >= a.H || s.dead[(y / TH - (tyi - 1)) * 3 + j]) v = 0ull;
            if (j == 0) v = d ? (v >> (64 - d)) << (64 - d) : 0ull;             // columns tx0-d .. tx0-1
            if (j == 2) v = d ? (v & ((1ull << d) - 1ull)) : 0ull;              // columns tx0+64 .. tx0+63+d
            s.words[i] = v;
            c += __popcll(v);
        }
    }
    __syncthreads();
    u64 own = lane < TH ? s.words[(lane + d) * 3 + 1] : 0ull;    // the tile's own word of row ty0 + lane
    const int c_tile = __popcll(own);
    const int n = wave_sum(c), incl = wave_incl_scan(c_tile), n_tile = __shfl(incl, 63);
    if (n > NS) {                                      // uniform: too many for the pairwise test
        if (lane == 0) a.dlist[(long)b * a.nty * a.ntx + atomicAdd(ctr_dlist(a, b), 1)] = tile;
        return;
    }
    if (lane == 0) s.m = n_tile;
    {                                                  // the tile's candidates: entries 0 .. n_tile - 1
        int e = incl - c_tile;
        const int y = ty0 + lane;
        while (own) {
            s.pos[e++] = make_int2(tx0 + __builtin_ctzll(own), y);
            own &= own - 1ull;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {                      // the halo's candidates behind them
        const int i = lane + 64 * k;
        if (i < RH * 3) {
            const int r = i / 3, j = i - r * 3, y = ry0 + r;
            u64 v = s.words[i];
            if (j == 1 && r >= d && r < d + TH) v = 0ull;
            if (v) {
                int e = atomicAdd(&s.m, __popcll(v));
                while (v) {
                    s.pos[e++] = make_int2(tx0 + (j - 1) * 64 + __builtin_ctzll(v), y);
                    v &= v - 1ull;
                }
            }
        }
    }
    __syncthreads();
    for (int e = lane; e < n; e += 64) {               // the scores: independent loads, one per lane
        const int2 p = s.pos[e];
        s.key[e] = make_key(g_score(a, img, p.y, p.x), p.y * a.W + p.x);
    }
    __syncthreads();
    for (int e = lane; e < n_tile; e += 64) {
        const int2 p = s.pos[e];
        const double k = s.key[e];
        bool keep = true;
        for (int q = 0; q < n; ++q) {
            const int2 o = s.pos[q];
            if (s.key[q] > k && abs(o.x - p.x) <= d && abs(o.y - p.y) <= d) { keep = false; break; }
        }
        if (keep) atomicOr(&s.kbits[(p.y - ty0) * 2 + ((p.x - tx0) >> 5)], 1u << ((p.x - tx0) & 31));
    }
    __syncthreads();
    u64 kw = 0ull;
    if (lane < TH && ty0 + lane < a.H) {
        kw = (u64)s.kbits[lane * 2] | ((u64)s.kbits[lane * 2 + 1] << 32);
        a.kept[((long)b * a.H + ty0 + lane) * a.ntx + txi] = kw;
    }
    if (__ballot(kw != 0ull) != 0ull && lane == 0) a.kept_round[(long)b * a.nty * a.ntx + tile] = round;
}

// Round 1: every tile, candidates straight from the score map (crop, border, threshold in the load), window mode.
// PERSISTENT: two workgroups per CU walk contiguous runs of tiles (32 640 short-lived workgroups with 75 KB of LDS each
// spent a third of the pass being launched), and the next tile's score values are requested into registers before the
// two window passes of the current one, so that the HBM round trip runs under them (it was half of the pass).
constexpr int FIRST_WG = 512;
#ifndef BALF_GREEDY_FT
#define BALF_GREEDY_FT 512      // (256: 0.88 ms of keep passes per 32 x 1080p, 512: 0.79 -- four waves per SIMD hide the LDS round trips of the window passes)
#endif
constexpr int FTHREADS = BALF_GREEDY_FT;                 // threads of the round-1 workgroup
constexpr int FROWS = RH_MAX / (FTHREADS / 64);          // region rows per wave

// wave wv holds region rows wv, wv + 4, ..., lanes along x (columns lane and lane + 64): coalesced, no index division
__device__ __forceinline__ void first_fetch(const GreedyArgs &a, int g, float (&v)[FROWS][2]) {
    const int tiles = a.nty * a.ntx, b = g / tiles, tile = g - b * tiles;
    const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
    const int d = a.d, RH = TH + 2 * d, RW = TW + 2 * d, ry0 = tyi * TH - d, rx0 = txi * TW - d;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lo_y = a.border, hi_y = a.H - a.border, lo_x = a.border, hi_x = a.W - a.border;
    const float *img = a.src + (long)b * a.Hs * a.Ws + (long)a.crop_y * a.Ws + a.crop_x;
#pragma unroll
    for (int k = 0; k < FROWS; ++k) {
        const int r = wv + (FTHREADS / 64) * k, y = ry0 + r;
        const bool yok = r < RH && y >= lo_y && y < hi_y;
        const float *rowp = img + (long)y * a.Ws;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h, x = rx0 + c;
            v[k][h] = (yok && c < RW && x >= lo_x && x < hi_x) ? rowp[x] : 0.0f;
        }
    }
}

template <int DC>     // DC = 15: the demo's dist_thresh with every window loop unrolled and its index arithmetic folded; -1: any distance
__global__ __launch_bounds__(FTHREADS) void greedy_keep_first_kernel(GreedyArgs a, int total, int per) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KeepLds &s = *reinterpret_cast<KeepLds *>(smem);
    // consecutive workgroups go to different XCDs (8 L2s): workgroup i takes run (i % 8) * (n / 8) + i / 8, so that each XCD
    // walks one contiguous eighth of the batch and the halo rows neighbouring tiles share are fetched into one L2 once
    const int run = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int g0 = run * per, g1 = g0 + per < total ? g0 + per : total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles = a.nty * a.ntx, d = DC >= 0 ? DC : a.d, RH = TH + 2 * d, RW = TW + 2 * d;
    float v[FROWS][2];
    if (g0 < g1) first_fetch(a, g0, v);
    for (int g = g0; g < g1; ++g) {
        const int b = g / tiles, tile = g - b * tiles, tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
        const int ry0 = tyi * TH - d, rx0 = txi * TW - d;
        lds_barrier();                              // the previous tile is done with the LDS
        if (tid < TH * 2) { s.kbits[tid] = 0u; s.abits[tid] = 0u; }
#pragma unroll
        for (int k = 0; k < FROWS; ++k) {
            const int r = wv + (FTHREADS / 64) * k;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (r < RH && c < RW) s.in[r * RS + c] = v[k][h] >= a.conf ? make_key(v[k][h], (ry0 + r) * a.W + rx0 + c) : 0.0;
            }
        }
        if (g + 1 < g1) first_fetch(a, g + 1, v);     // in flight during the passes
        window_passes<true, DC>(a, s, tyi, txi);
        lds_barrier();
        if (tid < 64) {
            u64 kw = 0ull;
            if (tid < TH && tyi * TH + tid < a.H) {
                const long wi = ((long)b * a.H + tyi * TH + tid) * a.ntx + txi;
                kw = (u64)s.kbits[tid * 2] | ((u64)s.kbits[tid * 2 + 1] << 32);
                a.kept[wi] = kw;
                a.alive[wi] = (u64)s.abits[tid * 2] | ((u64)s.abits[tid * 2 + 1] << 32);
            }
            if (__ballot(kw != 0ull) != 0ull && tid == 0) a.kept_round[(long)b * tiles + tile] = 1;
        }
    }
}

// rounds >= 2, first launch: the listed tiles in pairwise mode; crowded tiles are passed on to greedy_keep_window_kernel
__global__ __launch_bounds__(LTHREADS) void greedy_keep_sparse_kernel(GreedyArgs a, int round) {
    __shared__ SparseLds s;
    const int b = blockIdx.y, tiles = a.nty * a.ntx, cur = round & 1;
    const int n = *ctr_list(a, cur, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) *ctr_list(a, cur ^ 1, b) = 0;       // this round's kill pass fills it
    const int *list = a.list + ((long)cur * a.B + b) * tiles;
    for (int i = blockIdx.x; i < n; i += gridDim.x) keep_tile_sparse(a, s, b, list[i], round);
}

// rounds >= 2, second launch: the crowded tiles (ramps, plateaus) with the window maximum
__global__ __launch_bounds__(KTHREADS) void greedy_keep_window_kernel(GreedyArgs a, int round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KeepLds &s = *reinterpret_cast<KeepLds *>(smem);
    const int b = blockIdx.y, tiles = a.nty * a.ntx;
    const int n = *ctr_dlist(a, b);
    const int *list = a.dlist + (long)b * tiles;
    for (int i = blockIdx.x; i < n; i += gridDim.x) keep_tile(a, s, b, list[i], round, false);    // (listed by the pairwise kernel: active)
}

template <bool FIRST>
__global__ __launch_bounds__(LTHREADS) void greedy_kill_kernel(GreedyArgs a, int round) {
    __shared__ KillLds s;
    const int b = blockIdx.y, tiles = a.nty * a.ntx;
    if (FIRST) {
        for (int t = blockIdx.x; t < tiles; t += gridDim.x) kill_tile(a, s, b, t, round, false);
        return;
    }
    const int cur = round & 1;
    const int n = *ctr_list(a, cur, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) *ctr_dlist(a, b) = 0;                 // the window-mode list of the next round
    const int *list = a.list + ((long)cur * a.B + b) * tiles;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int tile = list[i], tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
        if (nbr_round_is_wave(a, a.kept_round, b, tyi, txi, round)) {
            kill_tile(a, s, b, tile, round, false);
        } else if (threadIdx.x == 0) {                 // no newly kept point in reach: the tile lives on unchanged
            const int nxt = (round + 1) & 1;
            a.list[((long)nxt * a.B + b) * tiles + atomicAdd(ctr_list(a, nxt, b), 1)] = tile;
        }
    }
}

// Whatever the enqueued rounds left alive: one workgroup per image runs further rounds over the image's own tile list.
// The workgroup reads what it wrote itself in the previous pass (bit maps, lists, dead_round): device-scope fences around
// the barriers, the counters through atomic loads.  Always launched: it also hands the survivor count to the top-K kernel.
struct TailLds {
    int act[KTHREADS];             // the listed tiles of the current chunk that have something to do
    int nact, pad_[3];
};

__global__ __launch_bounds__(KTHREADS) void greedy_tail_kernel(GreedyArgs a, int round0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KeepLds &sk = *reinterpret_cast<KeepLds *>(smem);
    KillLds &sl = *reinterpret_cast<KillLds *>(smem + sizeof(KeepLds));
    TailLds &st = *reinterpret_cast<TailLds *>(smem + sizeof(KeepLds) + sizeof(KillLds));
    const int b = blockIdx.x, tiles = a.nty * a.ntx, tid = threadIdx.x;
    for (int round = round0;; ++round) {
        const int cur = round & 1;
        const int n = __hip_atomic_load(ctr_list(a, cur, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == 0) break;                                    // uniform: every thread reads the same settled word
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctr_list(a, cur ^ 1, b), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int *list = a.list + ((long)cur * a.B + b) * tiles;
        int *next = a.list + ((long)(cur ^ 1) * a.B + b) * tiles;
        // The list is walked in chunks of one tile per thread: every thread decides for ITS tile whether the pass has anything to
        // do there (nbr_round_is), the tiles that do are collected in LDS and worked off one after the other by the whole
        // workgroup.  On a ramp or a plateau -- the inputs that reach this kernel with long lists -- that is the wavefront only.
        for (int base = 0; base < n; base += KTHREADS) {      // keep pass
            __syncthreads();
            if (tid == 0) st.nact = 0;
            __syncthreads();
            if (base + tid < n) {
                const int tile = list[base + tid], tyi = tile / a.ntx;
                if (nbr_round_is(a, a.chg_round, b, tyi, tile - tyi * a.ntx, round - 1)) st.act[atomicAdd(&st.nact, 1)] = tile;
            }
            __syncthreads();
            const int m = st.nact;
            for (int j = 0; j < m; ++j) keep_tile(a, sk, b, st.act[j], round, false);
        }
        __threadfence();
        __syncthreads();
        __threadfence();
        for (int base = 0; base < n; base += KTHREADS) {      // kill pass
            __syncthreads();
            if (tid == 0) st.nact = 0;
            __syncthreads();
            if (base + tid < n) {
                const int tile = list[base + tid], tyi = tile / a.ntx;
                if (nbr_round_is(a, a.kept_round, b, tyi, tile - tyi * a.ntx, round)) st.act[atomicAdd(&st.nact, 1)] = tile;
                else next[atomicAdd(ctr_list(a, cur ^ 1, b), 1)] = tile;          // no newly kept point in reach: lives on unchanged
            }
            __syncthreads();
            const int m = st.nact;
            for (int j = 0; j < m; ++j) kill_tile(a, sl, b, st.act[j], round, false);
        }
        __threadfence();
        __syncthreads();
        __threadfence();
    }
    if (tid == 0) {
        const int n = __hip_atomic_load(ctr_surv(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.counts[b] = n;
        if (a.total) a.total[b] = n;
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

struct Layout {
    size_t zero_bytes;            // the counters (one line each) + counts [B]: cleared per call
    size_t off_counts, off_chg, off_keptr, off_dead, dead_bytes, off_list, off_dlist, off_alive, off_kept, off_surv, total;
};
Layout layout(int B, int H, int W) {
    const size_t ntx = balf_ceil_div(W, TW), nty = balf_ceil_div(H, TH), tiles = ntx * nty;
    Layout l;
    l.off_counts = (size_t)4 * B * CTR_STRIDE * sizeof(int);
    l.dead_bytes = balf_align_up((size_t)B * tiles * sizeof(int), 256);
    l.off_chg = l.off_counts + balf_align_up((size_t)B * sizeof(int), 256);
    l.off_keptr = l.off_chg + l.dead_bytes;
    l.zero_bytes = l.off_keptr + l.dead_bytes;
    l.off_dead = l.zero_bytes;
    l.off_list = l.off_dead + l.dead_bytes;
    l.off_dlist = l.off_list + balf_align_up((size_t)2 * B * tiles * sizeof(int), 256);
    l.off_alive = l.off_dlist + balf_align_up((size_t)B * tiles * sizeof(int), 256);
    const size_t map = balf_align_up((size_t)B * H * ntx * sizeof(u64), 256);
    l.off_kept = l.off_alive + map;
    l.off_surv = l.off_kept + map;
    l.total = l.off_surv + (size_t)B * H * W * sizeof(int2);
    return l;
}

// hipFuncSetAttribute is a driver round trip: once per kernel and device of the process (as allow_lds32 of detector.hip)
template <auto Kernel>
bool allow_lds(int bytes) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done.load(std::memory_order_acquire) >> dev & 1) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return false;
    done.fetch_or(1ull << dev, std::memory_order_release);
    return true;
}

// rounds enqueued before the tail kernel; BALF_GREEDY_ROUNDS overrides it (read per call: the tests run the tail with 1)
int rounds_launched() {
    const char *e = getenv("BALF_GREEDY_ROUNDS");
    const int v = e ? atoi(e) : ROUNDS_DEFAULT;
    return v < 1 ? 1 : (v > 256 ? 256 : v);
}

}  // namespace

extern "C" size_t balf_greedy_nms_workspace_bytes(int B, int H, int W, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    return layout(B, H, W).total;
}

extern "C" int balf_greedy_nms(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                               int border, float conf_thresh, int dist_thresh, int K, int subpixel_patch,
                               int32_t *idx_dev, float *score_dev, float *xy_dev, int32_t *count_dev,
                               int32_t *total_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!prob_dev || !idx_dev || !score_dev || !count_dev || !workspace_dev) return BALF_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || K > BALF_MAX_TOPK || border < 0) return BALF_ERR_ARG;
    if (B > 65535) return BALF_ERR_ARG;                       // blockIdx.y
    if (!(conf_thresh > 0.0f) || dist_thresh < 0 || dist_thresh > GD_MAX) return BALF_ERR_ARG;
    if (subpixel_patch < 0 || subpixel_patch > 16 || (subpixel_patch > 0 && !xy_dev)) return BALF_ERR_ARG;
    if (crop_y < 0 || crop_x < 0 || crop_y + H > Hp || crop_x + W > Wp) return BALF_ERR_SHAPE;
    if ((long)H * W > 0x7fffffffL) return BALF_ERR_SHAPE;
    if ((long)balf_ceil_div(W, TW) * balf_ceil_div(H, TH) * B > 0x7fffffffL) return BALF_ERR_SHAPE;
    const Layout l = layout(B, H, W);
    if (workspace_bytes < l.total) return BALF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char *w = static_cast<char *>(workspace_dev);
    const int ntx = balf_ceil_div(W, TW), nty = balf_ceil_div(H, TH), tiles = ntx * nty;

    GreedyArgs a{prob_dev, Hp, Wp, crop_y, crop_x, H, W, border, conf_thresh, dist_thresh, B, ntx, nty,
                 getenv("BALF_GREEDY_NOSKIP") != nullptr,
                 reinterpret_cast<u64 *>(w + l.off_alive), reinterpret_cast<u64 *>(w + l.off_kept),
                 reinterpret_cast<int *>(w + l.off_dead), reinterpret_cast<int *>(w + l.off_chg),
                 reinterpret_cast<int *>(w + l.off_keptr), reinterpret_cast<int *>(w + l.off_list),
                 reinterpret_cast<int *>(w + l.off_dlist), reinterpret_cast<int *>(w),
                 reinterpret_cast<int2 *>(w + l.off_surv), reinterpret_cast<int *>(w + l.off_counts), total_dev};
    // (fill kernels, not hipMemsetAsync: see balf_fill_u32 in common.h)
    if (balf_fill_u32(w, 0u, l.zero_bytes / 4, st) != BALF_OK) return BALF_ERR_LAUNCH;
    if (balf_fill_u32(w + l.off_dead, (unsigned)ALIVE_FOREVER, l.dead_bytes / 4, st) != BALF_OK) return BALF_ERR_LAUNCH;

    constexpr int keep_lds = (int)sizeof(KeepLds), tail_lds = (int)(sizeof(KeepLds) + sizeof(KillLds) + sizeof(TailLds));
    static_assert(sizeof(KeepLds) % 16 == 0 && 2 * sizeof(KeepLds) <= 160 * 1024, "two window-mode workgroups per CU");
    if (!allow_lds<greedy_keep_first_kernel<15>>(keep_lds) || !allow_lds<greedy_keep_first_kernel<-1>>(keep_lds) ||
        !allow_lds<greedy_keep_window_kernel>(keep_lds) ||
        !allow_lds<greedy_tail_kernel>(tail_lds))
        return BALF_ERR_LAUNCH;
    // round 1 over every tile; rounds 2.. over the lists: launches sized for a full list that return at once on an empty one
    // (pairwise keep and kill: one wave per tile, the chip holds 8192 of them)
    const int g1 = tiles < 16384 / B + 64 ? tiles : 16384 / B + 64;
    const int gw = tiles < 512 / B + 16 ? tiles : 512 / B + 16;
    const int total = tiles * B;
    const int first_wg = total < FIRST_WG ? (total + 7) / 8 * 8 : FIRST_WG, per = (total + first_wg - 1) / first_wg;
    if (dist_thresh == 15)
        BALF_PROF(balf_prof::kGreedyKeep, st,
                  hipLaunchKernelGGL(greedy_keep_first_kernel<15>, dim3(first_wg), dim3(FTHREADS), keep_lds, st, a, total, per));
    else
        BALF_PROF(balf_prof::kGreedyKeep, st,
                  hipLaunchKernelGGL(greedy_keep_first_kernel<-1>, dim3(first_wg), dim3(FTHREADS), keep_lds, st, a, total, per));
    BALF_PROF(balf_prof::kGreedyKill, st,
              hipLaunchKernelGGL(greedy_kill_kernel<true>, dim3(tiles, B), dim3(LTHREADS), 0, st, a, 1));
    BALF_LAUNCH_CHECK();
    const int rounds = rounds_launched();
    for (int r = 2; r <= rounds; ++r) {
        // the lists shrink by a factor of 2-4 per round on a score map: later rounds get smaller grids (the workgroups stride over
        // the list, so a long list -- a ramp -- is still covered; an empty launch of the full grid costs 4.7 us, of 1/16 of it 2)
        const int gr = r <= 3 ? g1 : (r <= 6 ? (g1 + 3) / 4 : (g1 + 15) / 16);
        BALF_PROF(balf_prof::kGreedyKeep, st,
                  hipLaunchKernelGGL(greedy_keep_sparse_kernel, dim3(gr, B), dim3(LTHREADS), 0, st, a, r));
        BALF_PROF(balf_prof::kGreedyKeep, st,
                  hipLaunchKernelGGL(greedy_keep_window_kernel, dim3(gw, B), dim3(KTHREADS), keep_lds, st, a, r));
        BALF_PROF(balf_prof::kGreedyKill, st,
                  hipLaunchKernelGGL(greedy_kill_kernel<false>, dim3(gr, B), dim3(LTHREADS), 0, st, a, r));
    }
    BALF_LAUNCH_CHECK();
    BALF_PROF(balf_prof::kGreedyKill, st,
              hipLaunchKernelGGL(greedy_tail_kernel, dim3(B), dim3(KTHREADS), tail_lds, st, a, rounds + 1));
    BALF_LAUNCH_CHECK();
    // the K best kept points by score, sorted (score desc, index asc); count_dev = rows returned (<= K)
    int rc = balf_topk_select_launch(a.surv, a.counts, (long)H * W, B, K, /*zero_fallback=*/0, idx_dev, score_dev,
                                     count_dev, st, /*thr_explicit=*/0u);
    if (rc != BALF_OK) return rc;
    if (subpixel_patch > 0) {
        hipLaunchKernelGGL(subpixel_kernel, dim3(balf_ceil_div(K, 256), B), dim3(256), 0, st, a, idx_dev, count_dev, K,
                           subpixel_patch, xy_dev);
        BALF_LAUNCH_CHECK();
    }
    return BALF_OK;
}
