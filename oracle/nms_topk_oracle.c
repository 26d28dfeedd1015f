/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  Plain-C restatement of the
 * reference's post-processing on one score map, used by the tests at full 1080p sizes and as the
 * NMS/top-K leg of bench.py's cpu_baseline:
 *   remove_borders            /root/reference/balf/utils/test_utils.py:34-47
 *   apply_nms                 /root/reference/balf/utils/test_utils.py:50-54  (SciPy maximum_filter,
 *                             mode='reflect' == clipped window for a max filter)
 *   find_index_higher_scores  /root/reference/balf/utils/test_utils.py:74-95  (full sort, K-th value,
 *                             <= 0 fallback, raster-order scan truncated to K)
 * Pinned against tests/golden/nms_topk.npz by tests/test_oracle_golden.py.
 * Build: make -C oracle   (gcc -O2 -shared -fPIC) */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_desc(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* nms_out: H*W floats (scratch + result).  Returns the number of points written (<= K), -1 on bad
 * arguments (K > H*W is the reference's IndexError). */
int oracle_nms_topk(const float *score, int H, int W, int border, int size, int K, float *nms_out,
                    int32_t *idx_out, float *score_out) {
    if (H <= 0 || W <= 0 || size < 1 || K < 1 || (long)K > (long)H * W) return -1;
    const int lo = size / 2, hi = (size - 1) / 2;
    const long n = (long)H * W;
    float *rb = (float *)calloc(n, sizeof(float));
    float *rowmax = (float *)malloc(n * sizeof(float));
    float *sorted = (float *)malloc(n * sizeof(float));
    if (!rb || !rowmax || !sorted) { free(rb); free(rowmax); free(sorted); return -1; }
    for (int y = border; y < H - border; ++y)
        for (int x = border; x < W - border; ++x) rb[(long)y * W + x] = score[(long)y * W + x];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int x0 = x - lo < 0 ? 0 : x - lo, x1 = x + hi >= W ? W - 1 : x + hi;
            float m = rb[(long)y * W + x0];
            for (int k = x0 + 1; k <= x1; ++k) if (rb[(long)y * W + k] > m) m = rb[(long)y * W + k];
            rowmax[(long)y * W + x] = m;
        }
    for (int y = 0; y < H; ++y) {
        const int y0 = y - lo < 0 ? 0 : y - lo, y1 = y + hi >= H ? H - 1 : y + hi;
        for (int x = 0; x < W; ++x) {
            float m = rowmax[(long)y0 * W + x];
            for (int k = y0 + 1; k <= y1; ++k) if (rowmax[(long)k * W + x] > m) m = rowmax[(long)k * W + x];
            const float v = rb[(long)y * W + x];
            nms_out[(long)y * W + x] = v * (v == m ? 1.0f : 0.0f);
        }
    }
    memcpy(sorted, nms_out, n * sizeof(float));
    qsort(sorted, n, sizeof(float), cmp_desc);
    float thr = sorted[K - 1];
    if (thr <= 0.0f) {
        long last_pos = -1;
        for (long i = 0; i < n && sorted[i] > 0.0f; ++i) last_pos = i;
        thr = last_pos >= 0 ? sorted[last_pos] : 0.0f;
    }
    int cnt = 0;
    for (long i = 0; i < n && cnt < K; ++i)
        if (nms_out[i] >= thr) { idx_out[cnt] = (int32_t)i; score_out[cnt] = nms_out[i]; ++cnt; }
    free(rb); free(rowmax); free(sorted);
    return cnt;
}
