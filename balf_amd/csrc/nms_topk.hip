// Window-max NMS + exact K-th-threshold top-K on the score map, gfx950.
//
// Replaces (bit-exact on identical input) the reference's NumPy/SciPy post-processing:
//   crop                        /root/reference/balf/utils/train_utils.py:437-442
//   remove_borders              /root/reference/balf/utils/test_utils.py:34-47
//   apply_nms (maximum_filter)  /root/reference/balf/utils/test_utils.py:50-54
//   find_index_higher_scores    /root/reference/balf/utils/test_utils.py:74-95
//   final sort                  /root/reference/balf/utils/train_utils.py:451-452
//
// Kernels (all HBM-bound scans, no MFMA):
//   nms_tile_kernel<SIZE>  one 64x64 output tile per workgroup: tile + halo -> LDS with the crop
//                          offset and the border zeroing applied on the fly; separable window max
//                          (row pass, column pass; doubling trick for SIZE 15); survivors
//                          (v > 0 && v == window max) appended to a per-image list with one global
//                          atomic per workgroup, or, in dense mode, the apply_nms map is written.
//   topk_select_kernel     one workgroup per image: MSB-first radix select of the K-th largest
//                          survivor score (positive floats order like uint32); if more than K
//                          survivors reach it, a second radix select on the flat index keeps the
//                          raster-first K of them (as the reference does); LDS bitonic sort of the
//                          <= K selected (score desc, index asc) and the padded output rows.
#include <cstdint>
#include <cstdlib>
#include "common.h"
#include "prof.h"

namespace {

constexpr int TH = 64;   // tile height (outputs)
constexpr int TW = 64;   // tile width  (outputs)
constexpr int SEG = 16;  // outputs per thread along the sliding axis
constexpr int NTHREADS = 256;

template <int SIZE, int N>
__device__ __forceinline__ void window_max_1d(const float (&v)[N + SIZE - 1], float (&o)[N]) {
    if constexpr (SIZE == 15) {
        // max is idempotent, so overlapping windows combine: 2 -> 4 -> 8 -> 15 (= 8 + 8, overlap 1)
        constexpr int L = N + 14;
        float a[L - 1], b[L - 3], c[L - 7];
#pragma unroll
        for (int i = 0; i < L - 1; ++i) a[i] = fmaxf(v[i], v[i + 1]);
#pragma unroll
        for (int i = 0; i < L - 3; ++i) b[i] = fmaxf(a[i], a[i + 2]);
#pragma unroll
        for (int i = 0; i < L - 7; ++i) c[i] = fmaxf(b[i], b[i + 4]);
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = fmaxf(c[i], c[i + 7]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float m = v[i];
#pragma unroll
            for (int k = 1; k < SIZE; ++k) m = fmaxf(m, v[i + k]);
            o[i] = m;
        }
    }
}

struct NmsArgs {
    const float *src;      // [B, Hs, Ws]
    int Hs, Ws;            // pitch of the source maps
    int crop_y, crop_x;    // score(y,x) = src[b, crop_y + y, crop_x + x]
    int H, W;              // cropped size
    int border;
    int size;              // nms window side (runtime copy)
    int2 *surv;            // [B, cap] (flat idx, score bits)   (sparse mode)
    int *surv_count;       // [B]
    long cap;
    float *dense;          // [B, H, W] apply_nms output         (dense mode) or nullptr
};

__device__ __forceinline__ float load_score(const NmsArgs &a, const float *img, int y, int x) {
    if (y < 0 || y >= a.H || x < 0 || x >= a.W) return -INFINITY;          // clipped window
    if (y < a.border || y >= a.H - a.border || x < a.border || x >= a.W - a.border) return 0.0f;
    return img[(long)(a.crop_y + y) * a.Ws + (a.crop_x + x)];
}

// Block-wide exclusive scan of one int per thread (256 threads = 4 waves); returns the exclusive
// prefix and leaves the block total in *total.
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave /*[4]*/, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) base += (w < wave) ? s_wave[w] : 0;
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

template <int SIZE>
__global__ __launch_bounds__(NTHREADS) void nms_tile_kernel(NmsArgs a) {
    constexpr int R = SIZE - 1;
    constexpr int LO = SIZE / 2;              // window = [i - LO, i + (SIZE-1)/2]
    constexpr int IN_H = TH + R, IN_W = TW + R;
    constexpr int PIN = IN_W | 1;             // odd pitch: lanes walk rows in the row pass
    constexpr int PROW = TW + 1;
    __shared__ float s_in[IN_H * PIN];
    __shared__ float s_row[IN_H * PROW];
    __shared__ int s_scan[4];
    __shared__ int s_base;

    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    const int tid = threadIdx.x;

    {   // tile + halo -> LDS; every load of this thread is issued before the first store (latency paid once)
        constexpr int NLD = (IN_H * IN_W + NTHREADS - 1) / NTHREADS;
        float tmp[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * NTHREADS;
            const int r = i / IN_W, c = i - r * IN_W;
            tmp[j] = (i < IN_H * IN_W) ? load_score(a, img, ty0 - LO + r, tx0 - LO + c) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * NTHREADS;
            const int r = i / IN_W, c = i - r * IN_W;
            if (i < IN_H * IN_W) s_in[r * PIN + c] = tmp[j];
        }
    }
    __syncthreads();

    // row pass: item = (row, 16-column segment); consecutive lanes take consecutive rows
    for (int item = tid; item < IN_H * (TW / SEG); item += NTHREADS) {
        const int r = item % IN_H, seg = item / IN_H;
        float v[SEG + R], o[SEG];
#pragma unroll
        for (int k = 0; k < SEG + R; ++k) v[k] = s_in[r * PIN + seg * SEG + k];
        window_max_1d<SIZE, SEG>(v, o);
#pragma unroll
        for (int k = 0; k < SEG; ++k) s_row[r * PROW + seg * SEG + k] = o[k];
    }
    __syncthreads();

    // column pass: thread = (column, 16-row segment)
    const int x = tid & (TW - 1), seg = tid / TW;
    float v[SEG + R], m[SEG];
#pragma unroll
    for (int k = 0; k < SEG + R; ++k) v[k] = s_row[(seg * SEG + k) * PROW + x];
    window_max_1d<SIZE, SEG>(v, m);

    const int gx = tx0 + x;
    unsigned keepmask = 0;
    float cen[SEG];
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        const int gy = ty0 + seg * SEG + j;
        cen[j] = s_in[(seg * SEG + j + LO) * PIN + x + LO];
        const bool inside = (gy < a.H) && (gx < a.W);
        const bool keep = inside && (cen[j] == m[j]);
        if (a.dense != nullptr) {
            if (inside) a.dense[((long)b * a.H + gy) * a.W + gx] = cen[j] * (keep ? 1.0f : 0.0f);
        } else if (keep && cen[j] > 0.0f) {
            keepmask |= 1u << j;
        }
    }
    if (a.dense != nullptr) return;

    int total;
    const int excl = block_exclusive_scan(__popc(keepmask), s_scan, &total);
    if (total == 0) return;                                   // block-uniform
    if (tid == 0) s_base = atomicAdd(&a.surv_count[b], total);
    __syncthreads();
    long pos = (long)b * a.cap + s_base + excl;
#pragma unroll
    for (int j = 0; j < SEG; ++j) {
        if (keepmask & (1u << j)) {
            const int gy = ty0 + seg * SEG + j;
            a.surv[pos++] = make_int2(gy * a.W + gx, __float_as_int(cen[j]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Window 15, survivor-list mode, 16-byte aligned rows: the tuned form of nms_tile_kernel<15>.
//   * one 64 (wide) x 114 (tall) output tile per workgroup: 128 input rows, so the row pass is exactly two items per
//     thread (the 64 x 64 tile has 78 rows x 4 segments = 312 items for 256 threads: a second round at 22 % use) and
//     the vertical halo is 12 % instead of 22 %; the kernel's time goes with the number of workgroups (per-tile overheads);
//   * the tile + halo (columns tx0 - 8 .. tx0 + 71) comes in as 16-byte loads, five per thread, goes to LDS as
//     16-byte stores and is read back by the row pass as 16-byte reads (pitch 21 x 16 B: consecutive rows fall into
//     different bank groups); the per-element crop / border / clip rules are applied in registers;
//   * the maxima are integer maxima on the float bits, three inputs per instruction (v_max3_i32), windows
//     3 -> 9 -> 15: 66 instructions per 16 outputs instead of 95 two-input v_max_f32 (which cost 5.8 cycles each).
//     Every value here is a score (>= 0 counts; anything <= 0 never survives) or the clip value -inf, and signed
//     integer order equals float order on positive floats, so whenever the centre can survive (> 0) the integer
//     window maximum IS the float window maximum.  (The dense apply_nms map needs the maximum of negative windows
//     too: it stays with nms_tile_kernel.)
// Same survivors as nms_tile_kernel<15>; their order in the list is arbitrary in both (selection is by value).
// ---------------------------------------------------------------------------------------------
constexpr int V_INH = 128, V_TH = V_INH - 14, V_IN4 = 20, V_PIN4 = 21, V_PROW4 = 17, V_SEGC = (V_TH + 3) / 4, V_NLD = V_INH * V_IN4 / NTHREADS;
static_assert(V_INH * V_IN4 % NTHREADS == 0 && V_INH % 64 == 0, "whole chunks per thread, whole waves per row-pass round");

__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }   // v_max3_i32

// o[i] = max(v[i .. i + 14]), i < N, on the float bits
template <int N>
__device__ __forceinline__ void window15_i(const int (&v)[N + 14], int (&o)[N]) {
    int t3[N + 12], t9[N + 6];
#pragma unroll
    for (int i = 0; i < N + 12; ++i) t3[i] = imax3(v[i], v[i + 1], v[i + 2]);
#pragma unroll
    for (int i = 0; i < N + 6; ++i) t9[i] = imax3(t3[i], t3[i + 3], t3[i + 6]);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = max(t9[i], t9[i + 6]);
}

__global__ __launch_bounds__(NTHREADS) void nms_tile15_vec_kernel(NmsArgs a) {
    __shared__ int4 s_in[V_INH * V_PIN4];
    __shared__ int4 s_row[V_INH * V_PROW4];
    __shared__ int s_scan[4];
    __shared__ int s_base;
    // XCD-aware tile order (speed only; round 5).  A tile's 80-column input row starts 32 bytes before a 256-byte boundary and
    // touches FOUR 128-byte lines, two of them shared with the x-neighbours, and 14 of its 128 input rows with the y-neighbours:
    // 2.25x the output bytes per tile -- which is what the counters showed leaving HBM (291 MB per 16 x 1080p against 133 MB,
    // TCC hit rate 9 %), because consecutive workgroups go to different XCDs and each XCD's L2 fetched the shared lines again.
    // With every XCD walking its own contiguous run of tiles (row-major), the shared lines are L2 hits.
    const int ntx = gridDim.x, nty = gridDim.y;
    const int lin = blockIdx.x + ntx * (blockIdx.y + nty * blockIdx.z), ntile = ntx * nty * gridDim.z;
    const int xq = ntile >> 3, xr = ntile & 7, xl = lin & 7, xj = lin >> 3;
    const int item = (xl < xr ? xl * (xq + 1) : xr * (xq + 1) + (xl - xr) * xq) + xj;
    const int b = item / (ntx * nty), trem = item - b * (ntx * nty);
    const int tyi = trem / ntx, txi = trem - tyi * ntx;
    const int ty0 = tyi * V_TH, tx0 = txi * TW;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    const int tid = threadIdx.x;
    constexpr int NEG_INF = (int)0xFF800000u;

    {   // tile + halo -> LDS: 64 rows x 20 chunks of four columns = 5 chunks per thread, all requested before the first store
        // (block-uniform) the tile and its halo lie inside the border frame -- 85 % of the tiles of a 1080p map: plain copy
        const bool interior = tx0 - 8 >= a.border && tx0 + 72 <= a.W - a.border && ty0 - 7 >= a.border && ty0 - 7 + V_INH <= a.H - a.border;
        int4 tmp[V_NLD];
        if (interior) {
            const float *base = img + (long)(a.crop_y + ty0 - 7) * a.Ws + (a.crop_x + tx0 - 8);
#pragma unroll
            for (int j = 0; j < V_NLD; ++j) {
                const int i = tid + j * NTHREADS;
                const int r = i / V_IN4, c4 = i - r * V_IN4;
                tmp[j] = *reinterpret_cast<const int4 *>(base + (long)r * a.Ws + 4 * c4);
            }
        } else {
            // every chunk is loaded from a clamped (always valid) address -- no branch around the loads, they all go out
            // together --; an address is only ever clamped for chunks whose four elements are all clipped (x, crop_x and
            // the pitch are multiples of four: a chunk never straddles the end of a source row)
            int xs[V_NLD], ys[V_NLD];
#pragma unroll
            for (int j = 0; j < V_NLD; ++j) {
                const int i = tid + j * NTHREADS;
                const int r = i / V_IN4, c4 = i - r * V_IN4;
                ys[j] = ty0 - 7 + r;
                xs[j] = tx0 - 8 + 4 * c4;
                const int yc = min(max(ys[j], 0), a.H - 1), sx = min(max(a.crop_x + xs[j], 0), a.Ws - 4);
                tmp[j] = *reinterpret_cast<const int4 *>(img + (long)(a.crop_y + yc) * a.Ws + sx);
            }
#pragma unroll
            for (int j = 0; j < V_NLD; ++j) {
                const int y = ys[j], x = xs[j];
                const bool row_out = y < 0 || y >= a.H, row_border = y < a.border || y >= a.H - a.border;
                int e[4] = {tmp[j].x, tmp[j].y, tmp[j].z, tmp[j].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int xx = x + k;
                    const bool out = row_out || xx < 0 || xx >= a.W;                                  // clipped window
                    const bool bord = row_border || xx < a.border || xx >= a.W - a.border;            // remove_borders
                    e[k] = out ? NEG_INF : (bord ? 0 : e[k]);
                }
                tmp[j] = make_int4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < V_NLD; ++j) {
            const int i = tid + j * NTHREADS;
            const int r = i / V_IN4, c4 = i - r * V_IN4;
            s_in[r * V_PIN4 + c4] = tmp[j];
        }
    }
    __syncthreads();

    // row pass: lane = input row, wave = 16-column segment; outputs x = 16 seg .. + 15 need tile columns 16 seg + 1 .. + 30
#pragma unroll
    for (int rr = 0; rr < V_INH / 64; ++rr) {
        const int r = rr * 64 + (tid & 63), seg = tid >> 6;
        int w[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int4 q = s_in[r * V_PIN4 + 4 * seg + k];
            w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
        }
        int v[30], o[16];
#pragma unroll
        for (int k = 0; k < 30; ++k) v[k] = w[k + 1];
        window15_i<16>(v, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) s_row[r * V_PROW4 + 4 * seg + k] = make_int4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
    __syncthreads();

    // column pass: thread = (column, quarter); a quarter owns V_SEGC output rows (the last one starts early so that its
    // input rows stay inside the tile, and drops the rows the third quarter already owns)
    const int x = tid & 63, qd = tid >> 6;
    const int start = qd < 3 ? V_SEGC * qd : V_TH - V_SEGC, first = V_SEGC * qd - start;
    const int *rowp = reinterpret_cast<const int *>(s_row), *inp = reinterpret_cast<const int *>(s_in);
    int v[V_SEGC + 14], m[V_SEGC];
#pragma unroll
    for (int k = 0; k < V_SEGC + 14; ++k) v[k] = rowp[(start + k) * (4 * V_PROW4) + x];
    window15_i<V_SEGC>(v, m);
    const int gx = tx0 + x;
    unsigned keepmask = 0;
    int cen[V_SEGC];
#pragma unroll
    for (int j = 0; j < V_SEGC; ++j) {
        const int gy = ty0 + start + j;
        cen[j] = inp[(start + j + 7) * (4 * V_PIN4) + x + 8];
        if (j >= first && gy < a.H && gx < a.W && cen[j] == m[j] && cen[j] > 0) keepmask |= 1u << j;   // (> 0 as int == > 0 as float)
    }
    int total;
    const int excl = block_exclusive_scan(__popc(keepmask), s_scan, &total);
    if (total == 0) return;                                   // block-uniform
    if (tid == 0) s_base = atomicAdd(&a.surv_count[b], total);
    __syncthreads();
    long pos = (long)b * a.cap + s_base + excl;
    // (a thread keeps 0.1 points on average: a loop over the set bits, not V_SEGC predicated stores)
    while (keepmask) {
        const int j = __builtin_ctz(keepmask);
        keepmask &= keepmask - 1;
        int c = cen[0];
#pragma unroll
        for (int t = 1; t < V_SEGC; ++t) c = (t == j) ? cen[t] : c;
        a.surv[pos++] = make_int2((ty0 + start + j) * a.W + gx, c);
    }
}

// Any window size up to BALF_MAX_NMS_SIZE: plain per-output loops over the LDS tile.
__global__ __launch_bounds__(NTHREADS) void nms_tile_kernel_generic(NmsArgs a) {
    constexpr int RMAX = BALF_MAX_NMS_SIZE - 1;
    constexpr int PIN = (TW + RMAX) | 1;
    constexpr int PROW = TW + 1;
    __shared__ float s_in[(TH + RMAX) * PIN];
    __shared__ float s_row[(TH + RMAX) * PROW];
    __shared__ int s_scan[4];
    __shared__ int s_base;

    const int size = a.size, R = size - 1, LO = size / 2;
    const int IN_H = TH + R, IN_W = TW + R;
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const float *img = a.src + (long)b * a.Hs * a.Ws;
    const int tid = threadIdx.x;

    for (int i = tid; i < IN_H * IN_W; i += NTHREADS) {
        const int r = i / IN_W, c = i - r * IN_W;
        s_in[r * PIN + c] = load_score(a, img, ty0 - LO + r, tx0 - LO + c);
    }
    __syncthreads();
    for (int i = tid; i < IN_H * TW; i += NTHREADS) {
        const int r = i / TW, c = i - r * TW;
        float m = s_in[r * PIN + c];
        for (int k = 1; k < size; ++k) m = fmaxf(m, s_in[r * PIN + c + k]);
        s_row[r * PROW + c] = m;
    }
    __syncthreads();

    const int x = tid & (TW - 1), seg = tid / TW;
    const int gx = tx0 + x;
    unsigned keepmask = 0;
    float cen[SEG];
    for (int j = 0; j < SEG; ++j) {
        const int ry = seg * SEG + j;
        float m = s_row[ry * PROW + x];
        for (int k = 1; k < size; ++k) m = fmaxf(m, s_row[(ry + k) * PROW + x]);
        const int gy = ty0 + ry;
        cen[j] = s_in[(ry + LO) * PIN + x + LO];
        const bool inside = (gy < a.H) && (gx < a.W);
        const bool keep = inside && (cen[j] == m);
        if (a.dense != nullptr) {
            if (inside) a.dense[((long)b * a.H + gy) * a.W + gx] = cen[j] * (keep ? 1.0f : 0.0f);
        } else if (keep && cen[j] > 0.0f) {
            keepmask |= 1u << j;
        }
    }
    if (a.dense != nullptr) return;

    int total;
    const int excl = block_exclusive_scan(__popc(keepmask), s_scan, &total);
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&a.surv_count[b], total);
    __syncthreads();
    long pos = (long)b * a.cap + s_base + excl;
    for (int j = 0; j < SEG; ++j) {
        if (keepmask & (1u << j)) {
            const int gy = ty0 + seg * SEG + j;
            a.surv[pos++] = make_int2(gy * a.W + gx, __float_as_int(cen[j]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// top-K selection: one 1024-thread workgroup per image
// ---------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;

// Radix select over 32-bit keys of the elements that pass `pred`.  Finds the key of rank `rank`
// (1-based) counting from the top (FROM_TOP) or from the bottom; *n_same = how many elements carry
// exactly that key and *rank_in_same = how many of them are needed to reach `rank`.
template <bool FROM_TOP, typename KeyFn>
__device__ unsigned radix_select(int n, int rank, bool cached, const int2 (&ent)[16], const int2 *surv, KeyFn key_of,
                                 unsigned *s_hist /*[256]*/, int *s_tmp /*[4]*/, int *n_same, int *rank_in_same) {
    unsigned prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += SEL_THREADS) s_hist[i] = 0;
        __syncthreads();
        if (cached) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                unsigned k;
                if (threadIdx.x + j * SEL_THREADS < n && key_of(ent[j], &k) && (k & mask) == prefix)
                    atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
            }
        } else {
            for (int i = threadIdx.x; i < n; i += SEL_THREADS) {
                unsigned k;
                if (key_of(surv[i], &k) && (k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // first bin (in scan order) at which the running count reaches `rank`: wave 0, four bins per lane, a
            // shuffle prefix over the lanes (the scan used to be 256 serial LDS reads on thread 0, four times per select)
            const int l = threadIdx.x;
            int h[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 4 * l + i;
                h[i] = (int)s_hist[FROM_TOP ? 255 - t : t];
                sum += h[i];
            }
            int inc = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(inc, d, 64);
                if (l >= d) inc += v;
            }
            int cum = inc - sum;
            if (cum < rank && rank <= inc) {                 // exactly one lane
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (cum + h[i] >= rank) {
                        const int t = 4 * l + i;
                        s_tmp[0] = FROM_TOP ? 255 - t : t;
                        s_tmp[1] = rank - cum;
                        s_tmp[2] = h[i];
                        break;
                    }
                    cum += h[i];
                }
            }
        }
        __syncthreads();
        prefix |= (unsigned)s_tmp[0] << shift;
        mask |= 255u << shift;
        rank = s_tmp[1];
        __syncthreads();
    }
    *n_same = s_tmp[2];
    *rank_in_same = rank;
    return prefix;
}

// thr_explicit = 0: the threshold is the K-th largest survivor score (find_index_higher_scores with threshold = -1,
// test_utils.py:76-89).  thr_explicit != 0: the bits of a caller-given positive threshold (`threshold != -1`,
// test_utils.py:91-95): every survivor with score >= it, the raster-first K of them if there are more.
__global__ __launch_bounds__(SEL_THREADS) void topk_select_kernel(const int2 *surv_all, const int *surv_count,
                                                                  long cap, int K, int npow2, int zero_fallback,
                                                                  unsigned thr_explicit, int32_t *idx_out,
                                                                  float *score_out, int32_t *count_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);       // [npow2]
    unsigned *s_hist = reinterpret_cast<unsigned *>(keys + npow2);                 // [256]
    int *s_tmp = reinterpret_cast<int *>(s_hist + 256);                            // [4]
    int *s_cnt = s_tmp + 4;                                                        // [1]

    const int b = blockIdx.x;
    const int2 *surv = surv_all + (long)b * cap;
    const int n = surv_count[b];
    int32_t *idx_o = idx_out + (long)b * K;
    float *sc_o = score_out + (long)b * K;

    if (n == 0 && (!zero_fallback || thr_explicit)) {     // greedy-NMS caller / explicit threshold: no candidates, no points
        for (int i = threadIdx.x; i < K; i += SEL_THREADS) { idx_o[i] = -1; sc_o[i] = 0.0f; }
        if (threadIdx.x == 0) count_out[b] = 0;
        return;
    }
    if (n == 0) {
        // No positive NMS score: the reference's threshold falls back to 0.0 and `map >= 0` holds
        // everywhere, so it returns the first K pixels in raster order (test_utils.py:84-95).
        for (int i = threadIdx.x; i < K; i += SEL_THREADS) { idx_o[i] = i; sc_o[i] = 0.0f; }
        if (threadIdx.x == 0) count_out[b] = K;
        return;
    }

    // Selected set = the first K pixels in raster order among those with score >= thr, where thr is the
    // K-th largest score: with ties at thr this can drop a later, higher-scoring pixel -- that is what
    // `argwhere(map >= thr)[:K]` does (test_utils.py:93-95).
    // survivors of this image, cached in registers when they fit (<= 16 per thread = 16384: a 1080p image has
    // ~10-20 thousand): the radix passes then never go back to global memory
    constexpr int CACHE = 16;
    const bool cached = n <= CACHE * SEL_THREADS;
    int2 ent[CACHE];
    if (cached) {
#pragma unroll
        for (int j = 0; j < CACHE; ++j) {
            const int i = threadIdx.x + j * SEL_THREADS;
            ent[j] = (i < n) ? surv[i] : make_int2(0x7fffffff, 0);      // score bits 0 never reach a threshold > 0
        }
    }

    unsigned thr = 0;            // score bits; select score >= thr with idx <= idx_cut
    int idx_cut = 0x7fffffff;
    if (thr_explicit) {
        thr = thr_explicit;
        if (threadIdx.x == 0) *s_cnt = 0;
        __syncthreads();
        int mine = 0;
        if (cached) {
#pragma unroll
            for (int j = 0; j < CACHE; ++j)
                if (threadIdx.x + j * SEL_THREADS < n && (unsigned)ent[j].y >= thr) ++mine;
        } else {
            for (int i = threadIdx.x; i < n; i += SEL_THREADS)
                if ((unsigned)surv[i].y >= thr) ++mine;
        }
        if (mine) atomicAdd(s_cnt, mine);
        __syncthreads();
        const int reach = *s_cnt;
        __syncthreads();
        if (reach > K) {         // more than K pixels reach the threshold: `argwhere(...)[:K]` keeps the raster-first K
            int d0, d1;
            const unsigned t = thr;
            idx_cut = (int)radix_select<false>(
                n, K, cached, ent, surv, [t](int2 e, unsigned *k) { *k = (unsigned)e.x; return (unsigned)e.y >= t; },
                s_hist, s_tmp, &d0, &d1);
        }
    } else if (n > K) {
        int n_eq, need_eq;
        thr = radix_select<true>(n, K, cached, ent, surv, [](int2 e, unsigned *k) { *k = (unsigned)e.y; return true; },
                                 s_hist, s_tmp, &n_eq, &need_eq);
        if (n_eq > need_eq) {    // more than K candidates reach the threshold: keep the raster-first K
            int d0, d1;
            const unsigned t = thr;
            idx_cut = (int)radix_select<false>(
                n, K, cached, ent, surv, [t](int2 e, unsigned *k) { *k = (unsigned)e.x; return (unsigned)e.y >= t; },
                s_hist, s_tmp, &d0, &d1);
        }
    }

    if (threadIdx.x == 0) *s_cnt = 0;
    for (int i = threadIdx.x; i < npow2; i += SEL_THREADS) keys[i] = ~0ull;
    __syncthreads();
    auto take = [&](int2 e) {
        const unsigned sb = (unsigned)e.y;
        if (sb >= thr && e.x <= idx_cut) {
            const int p = atomicAdd(s_cnt, 1);
            keys[p] = ((unsigned long long)(~sb) << 32) | (unsigned)e.x;   // ascending = score desc, idx asc
        }
    };
    if (cached) {
#pragma unroll
        for (int j = 0; j < CACHE; ++j)
            if (threadIdx.x + j * SEL_THREADS < n) take(ent[j]);
    } else {
        for (int i = threadIdx.x; i < n; i += SEL_THREADS) take(surv[i]);
    }
    __syncthreads();
    const int cnt = *s_cnt;

    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += SEL_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = keys[i], y = keys[l];
                    const bool up = ((i & k) == 0);
                    if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < K; i += SEL_THREADS) {
        if (i < cnt) {
            const unsigned long long kv = keys[i];
            idx_o[i] = (int32_t)(unsigned)(kv & 0xffffffffull);
            sc_o[i] = __uint_as_float(~(unsigned)(kv >> 32));
        } else {
            idx_o[i] = -1;
            sc_o[i] = 0.0f;
        }
    }
    if (threadIdx.x == 0) count_out[b] = cnt;
}

int launch_nms_tiles(const NmsArgs &a, int B, hipStream_t stream) {
    dim3 grid(balf_ceil_div(a.W, TW), balf_ceil_div(a.H, TH), B), block(NTHREADS);
    if (balf_prof::g_on) balf_prof::before(balf_prof::kNmsTile, stream);
    // window 15 into the survivor list with 16-byte aligned rows (every caller of the detection path): the tuned kernel
    static const bool no_vec = getenv("BALF_NMS_NO_VEC") != nullptr;      // (A/B switch, development aid)
    if (a.size == 15 && a.dense == nullptr && !no_vec && a.Ws % 4 == 0 && a.crop_x % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 && ((long)a.Hs * a.Ws) % 4 == 0) {
        hipLaunchKernelGGL(nms_tile15_vec_kernel, dim3(balf_ceil_div(a.W, TW), balf_ceil_div(a.H, V_TH), B), block, 0, stream, a);
        if (balf_prof::g_on) balf_prof::after(stream);
        BALF_LAUNCH_CHECK();
        return BALF_OK;
    }
    switch (a.size) {
        case 15: hipLaunchKernelGGL(nms_tile_kernel<15>, grid, block, 0, stream, a); break;
        case 5: hipLaunchKernelGGL(nms_tile_kernel<5>, grid, block, 0, stream, a); break;
        case 3: hipLaunchKernelGGL(nms_tile_kernel<3>, grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL(nms_tile_kernel_generic, grid, block, 0, stream, a); break;
    }
    if (balf_prof::g_on) balf_prof::after(stream);
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned *dst, unsigned value, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = value;
}
}  // namespace

int balf_fill_u32(void *dst_dev, unsigned value, size_t n_words, hipStream_t stream) {
    if (n_words == 0) return BALF_OK;
    const size_t blocks = (n_words + 255) / 256;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream,
                       static_cast<unsigned *>(dst_dev), value, n_words);
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

// shared by balf_nms_topk and balf_greedy_nms (nms_fast.hip)
int balf_topk_select_launch(const int2 *surv, const int *counts, long cap, int B, int K, int zero_fallback,
                            int32_t *idx_dev, float *score_dev, int32_t *count_dev, hipStream_t st,
                            unsigned thr_explicit) {
    const int npow2 = next_pow2(K);
    const size_t smem = (size_t)npow2 * 8 + 256 * 4 + 8 * 4;
    if (smem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(topk_select_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return BALF_ERR_LAUNCH;
    BALF_PROF(balf_prof::kTopkSelect, st,
              hipLaunchKernelGGL(topk_select_kernel, dim3(B), dim3(SEL_THREADS), smem, st, surv, counts, cap, K, npow2,
                                 zero_fallback, thr_explicit, idx_dev, score_dev, count_dev));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

extern "C" int balf_window_nms(const float *score_dev, int B, int H, int W, int border, int nms_size,
                               float *out_dev, void *stream) {
    if (!score_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || border < 0) return BALF_ERR_ARG;
    if (nms_size < 1 || nms_size > BALF_MAX_NMS_SIZE) return BALF_ERR_ARG;
    if ((long)H * W > 0x7fffffffL) return BALF_ERR_SHAPE;
    NmsArgs a{score_dev, H, W, 0, 0, H, W, border, nms_size, nullptr, nullptr, 0, out_dev};
    return launch_nms_tiles(a, B, (hipStream_t)stream);
}

extern "C" size_t balf_nms_topk_workspace_bytes(int B, int H, int W, int K) {
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
    // [B] survivor counters (padded to 256 B) + [B, H*W] (idx, score) survivor slots.  Only slots that
    // receive a survivor are ever touched; H*W is the exact worst case (a constant plateau).
    return balf_align_up((size_t)B * sizeof(int), 256) + (size_t)B * (size_t)H * (size_t)W * sizeof(int2);
}

namespace {
int nms_select(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W, int border,
               int nms_size, int K, unsigned thr_explicit, int32_t *idx_dev, float *score_dev, int32_t *count_dev,
               void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!prob_dev || !idx_dev || !score_dev || !count_dev || !workspace_dev) return BALF_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || border < 0) return BALF_ERR_ARG;
    if (nms_size < 1 || nms_size > BALF_MAX_NMS_SIZE || K > BALF_MAX_TOPK) return BALF_ERR_ARG;
    if (crop_y < 0 || crop_x < 0 || crop_y + H > Hp || crop_x + W > Wp) return BALF_ERR_SHAPE;
    if ((long)H * W > 0x7fffffffL || (long)K > (long)H * W) return BALF_ERR_SHAPE;
    if (workspace_bytes < balf_nms_topk_workspace_bytes(B, H, W, K)) return BALF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;

    int *counts = reinterpret_cast<int *>(workspace_dev);
    int2 *surv = reinterpret_cast<int2 *>(reinterpret_cast<char *>(workspace_dev) +
                                          balf_align_up((size_t)B * sizeof(int), 256));
    if (balf_fill_u32(counts, 0u, balf_align_up((size_t)B * sizeof(int), 256) / 4, st) != BALF_OK) return BALF_ERR_LAUNCH;
    NmsArgs a{prob_dev, Hp, Wp, crop_y, crop_x, H, W, border, nms_size, surv, counts, (long)H * W, nullptr};
    int rc = launch_nms_tiles(a, B, st);
    if (rc != BALF_OK) return rc;

    return balf_topk_select_launch(surv, counts, (long)H * W, B, K, /*zero_fallback=*/1, idx_dev, score_dev, count_dev, st,
                                   thr_explicit);
}
}  // namespace

extern "C" int balf_nms_topk(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                             int border, int nms_size, int K, int32_t *idx_dev, float *score_dev,
                             int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    return nms_select(prob_dev, B, Hp, Wp, crop_y, crop_x, H, W, border, nms_size, K, 0u, idx_dev, score_dev, count_dev,
                      workspace_dev, workspace_bytes, stream);
}

extern "C" int balf_nms_threshold(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                                  int border, int nms_size, float threshold, int K, int32_t *idx_dev, float *score_dev,
                                  int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!(threshold > 0.0f) || !(threshold < INFINITY)) return BALF_ERR_ARG;   // <= 0 selects every pixel: host-side
    const unsigned bits = __builtin_bit_cast(unsigned, threshold);
    return nms_select(prob_dev, B, Hp, Wp, crop_y, crop_x, H, W, border, nms_size, K, bits, idx_dev, score_dev, count_dev,
                      workspace_dev, workspace_bytes, stream);
}
