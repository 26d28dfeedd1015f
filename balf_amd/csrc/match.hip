// Descriptor matching of the demo path on gfx950.
//
// Reference call site: /root/reference/demo/demo_match.py:104-110 -- kornia.feature.match_smnn(desc1, desc2, 0.99).
// kornia is not installed offline; this follows its published algorithm (kornia/feature/matching.py; PARITY
// UNPINNED, see oracle.py match_smnn): Euclidean distance matrix, per row the two smallest distances d1 <= d2,
// ratio test d1/d2 <= th in both directions, keep mutual pairs, distance = max of the two ratios, sorted by the
// index in desc1.
//
//   match_nn_kernel      one wave per 16 query rows; the 16x16 dot-product tile against 16 candidate rows is
//                        32 x v_mfma_f32_16x16x4_f32 (fp32 operands -- descriptors are compared at full precision);
//                        d^2 = |a|^2 + |b|^2 - 2 a.b clamped at 0; every lane keeps (best, second best, arg best)
//                        for its 4 rows over the candidate columns it sees, merged across lanes at the end
//   match_mutual_kernel  ratio + mutual check and an ordered compaction (single workgroup scan)
#include "common.h"
#include "prof.h"

namespace balf {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kD = 128;

struct Best {
    float d1, d2;
    int i1;
};

__device__ __forceinline__ void best_insert(Best &b, float d, int idx) {
    if (d < b.d1 || (d == b.d1 && idx < b.i1)) {
        b.d2 = b.d1; b.d1 = d; b.i1 = idx;
    } else if (d < b.d2) {
        b.d2 = d;
    }
}

__device__ __forceinline__ void best_merge(Best &b, float od1, float od2, int oi1) {
    if (od1 < b.d1 || (od1 == b.d1 && oi1 < b.i1)) {
        b.d2 = fminf(b.d1, od2); b.d1 = od1; b.i1 = oi1;
    } else {
        b.d2 = fminf(b.d2, od1);
    }
}

// lane (r = lane & 15, kq = lane >> 4) of a 16-row operand tile holds floats [16c + 4kq .. +3], c = 0..7, of row r;
// MFMA step 4c + j contracts k = 16c + 4kq + j on both operands.
__device__ __forceinline__ void load_rows(const float *base, int row, int nrows, int kq, f4 (&v)[8]) {
    const int r = row < nrows ? row : nrows - 1;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f4 *>(base + (size_t)r * kD + 16 * c + 4 * kq);
}

__device__ __forceinline__ float row_norm2(const f4 (&v)[8]) {
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c][0] * v[c][0] + v[c][1] * v[c][1] + v[c][2] * v[c][2] + v[c][3] * v[c][3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return s;
}

// nn_idx[i] = argmin_j |a_i - b_j|, nn_ratio[i] = d_first / d_second (NaN/inf when d_second = 0, as the reference)
// Batched over blockIdx.y: pair p compares a + p*stride_a (na_dev[p] rows) with b + p*stride_b (nb_dev[p] rows);
// na_dev / nb_dev == nullptr: one pair with the sizes passed by value.
__global__ __launch_bounds__(64) void match_nn_kernel(const float *a, int na, const int *na_dev, long stride_a,
                                                      const float *b, int nb, const int *nb_dev, long stride_b,
                                                      int *nn_idx, float *nn_ratio, int out_stride) {
    const int lane = threadIdx.x, n = lane & 15, kq = lane >> 4;
    const int pair = blockIdx.y;
    if (na_dev) { na = na_dev[pair]; nb = nb_dev[pair]; }
    a += pair * stride_a; b += pair * stride_b;
    nn_idx += (long)pair * out_stride; nn_ratio += (long)pair * out_stride;
    const int row0 = blockIdx.x * 16;
    if (row0 >= na || nb <= 0) return;
    f4 av[8];
    load_rows(a, row0 + n, na, kq, av);
    const float an = row_norm2(av);                 // |a|^2 of row (row0 + n), same in the 4 kq lanes
    float arow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) arow[r] = __shfl(an, 4 * kq + r);      // accumulator row 4kq + r
    Best best[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) best[r] = Best{INFINITY, INFINITY, 0x7fffffff};
    for (int c0 = 0; c0 < nb; c0 += 16) {
        f4 bv[8];
        load_rows(b, c0 + n, nb, kq, bv);
        const float bn = row_norm2(bv);
        f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][j], bv[c][j], acc, 0, 0, 0);
        if (c0 + n < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) best_insert(best[r], fmaxf(arow[r] + bn - 2.0f * acc[r], 0.0f), c0 + n);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const float od1 = __shfl_xor(best[r].d1, o), od2 = __shfl_xor(best[r].d2, o);
            const int oi1 = __shfl_xor(best[r].i1, o);
            best_merge(best[r], od1, od2, oi1);
        }
        const int row = row0 + 4 * kq + r;
        if (n == 0 && row < na) {
            nn_idx[row] = best[r].i1;
            nn_ratio[row] = sqrtf(best[r].d1) / sqrtf(best[r].d2);
        }
    }
}

__global__ __launch_bounds__(1024) void match_mutual_kernel(const int *idx1, const float *ratio1, int n1, const int *n1_dev,
                                                            int stride1, const int *idx2, const float *ratio2, int n2,
                                                            const int *n2_dev, int stride2, float th, int *out_idx,
                                                            float *out_dist, int *out_count, int cap) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = blockIdx.x;
    if (n1_dev) { n1 = n1_dev[pair]; n2 = n2_dev[pair]; }
    if (n1 < 2 || n2 < 2) th = -1.0f;       // the reference returns no matches when either side has < 2 descriptors
    idx1 += (long)pair * stride1; ratio1 += (long)pair * stride1;
    idx2 += (long)pair * stride2; ratio2 += (long)pair * stride2;
    out_idx += (long)pair * cap * 2; out_dist += (long)pair * cap; out_count += pair;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + tid;
        bool ok = false;
        int j = 0;
        float dist = 0.0f;
        if (i < n1) {
            j = idx1[i];
            const float r1 = ratio1[i];
            if (r1 <= th && j >= 0 && j < n2) {
                const float r2 = ratio2[j];
                ok = r2 <= th && idx2[j] == i;
                dist = fmaxf(r1, r2);
            }
        }
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (ok && off + before < cap) {
            out_idx[2 * (off + before)] = i;
            out_idx[2 * (off + before) + 1] = j;
            out_dist[off + before] = dist;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *out_count = base < cap ? base : cap;
    for (int k = base + tid; k < cap; k += 1024) {
        out_idx[2 * k] = -1; out_idx[2 * k + 1] = -1; out_dist[k] = 0.0f;
    }
}

}  // namespace
}  // namespace balf

using namespace balf;

extern "C" size_t balf_match_smnn_batch_workspace_bytes(int pairs, int k1, int k2) {
    if (pairs <= 0 || k1 <= 0 || k2 <= 0) return 0;
    return balf_align_up((size_t)pairs * k1 * 8, 256) + balf_align_up((size_t)pairs * k2 * 8, 256);
}

// pairs > 1 (or count pointers given): desc1 [pairs,k1,128], desc2 [pairs,k2,128] with n1_dev/n2_dev [pairs] valid rows
static int match_launch(const float *d1, int k1, const int *n1_dev, const float *d2, int k2, const int *n2_dev, int pairs,
                        float th, int32_t *idx_dev, float *dist_dev, int32_t *count_dev, void *workspace_dev,
                        hipStream_t st) {
    char *ws = static_cast<char *>(workspace_dev);
    int *idx1 = reinterpret_cast<int *>(ws);
    float *r1 = reinterpret_cast<float *>(ws + (size_t)pairs * k1 * 4);
    char *ws2 = ws + balf_align_up((size_t)pairs * k1 * 8, 256);
    int *idx2 = reinterpret_cast<int *>(ws2);
    float *r2 = reinterpret_cast<float *>(ws2 + (size_t)pairs * k2 * 4);
    BALF_PROF(balf_prof::kMatchNN, st,
              (match_nn_kernel<<<dim3(balf_ceil_div(k1, 16), pairs), 64, 0, st>>>(d1, k1, n1_dev, (long)k1 * kD, d2, k2, n2_dev,
                                                                                 (long)k2 * kD, idx1, r1, k1)));
    BALF_LAUNCH_CHECK();
    BALF_PROF(balf_prof::kMatchNN, st,
              (match_nn_kernel<<<dim3(balf_ceil_div(k2, 16), pairs), 64, 0, st>>>(d2, k2, n2_dev, (long)k2 * kD, d1, k1, n1_dev,
                                                                                 (long)k1 * kD, idx2, r2, k2)));
    BALF_LAUNCH_CHECK();
    const int cap = k1 < k2 ? k1 : k2;
    BALF_PROF(balf_prof::kMatchMutual, st,
              (match_mutual_kernel<<<pairs, 1024, 0, st>>>(idx1, r1, k1, n1_dev, k1, idx2, r2, k2, n2_dev, k2, th, idx_dev,
                                                            dist_dev, count_dev, cap)));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

extern "C" int balf_match_smnn_batch(const float *desc1_dev, int k1, const int32_t *n1_dev, const float *desc2_dev, int k2,
                                     const int32_t *n2_dev, int pairs, float th, int32_t *idx_dev, float *dist_dev,
                                     int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!desc1_dev || !desc2_dev || !n1_dev || !n2_dev || !idx_dev || !dist_dev || !count_dev || !workspace_dev)
        return BALF_ERR_ARG;
    if (pairs <= 0 || pairs > 65535 || k1 <= 0 || k2 <= 0) return BALF_ERR_ARG;
    if (workspace_bytes < balf_match_smnn_batch_workspace_bytes(pairs, k1, k2)) return BALF_ERR_WORKSPACE;
    return match_launch(desc1_dev, k1, n1_dev, desc2_dev, k2, n2_dev, pairs, th, idx_dev, dist_dev, count_dev, workspace_dev,
                        static_cast<hipStream_t>(stream));
}

extern "C" size_t balf_match_smnn_workspace_bytes(int n1, int n2) {
    return balf_match_smnn_batch_workspace_bytes(1, n1, n2);
}

extern "C" int balf_match_smnn(const float *desc1_dev, int n1, const float *desc2_dev, int n2, float th,
                               int32_t *idx_dev, float *dist_dev, int32_t *count_dev, void *workspace_dev,
                               size_t workspace_bytes, void *stream) {
    if (!desc1_dev || !desc2_dev || !idx_dev || !dist_dev || !count_dev || !workspace_dev || n1 <= 0 || n2 <= 0)
        return BALF_ERR_ARG;
    if (workspace_bytes < balf_match_smnn_workspace_bytes(n1, n2)) return BALF_ERR_WORKSPACE;
    return match_launch(desc1_dev, n1, nullptr, desc2_dev, n2, nullptr, 1, th, idx_dev, dist_dev, count_dev, workspace_dev,
                        static_cast<hipStream_t>(stream));
}
