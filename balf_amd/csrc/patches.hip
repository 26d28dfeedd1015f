// Patch extraction of the demo path on gfx950.
//
// Reference call site: /root/reference/demo/demo_match.py:62-70 --
//   laf = kornia.feature.laf_from_center_scale_ori(kp, s_mult * ones, zeros)
//   patches = kornia.feature.extract_patches_from_pyramid(gray / 255, laf, PS=32)
// kornia is a third-party dependency that is NOT installed offline, so this follows kornia's published
// algorithm (kornia/feature/laf.py, kornia/geometry/transform/pyramid.py; PARITY UNPINNED, see oracle.py
// extract_patches): one pyramid level for the whole call (the demo uses one scale for all keypoints),
//   level = clamp(floor(log2(2 s / PS)), 0, max(0, min(H,W)/PS - 1)),
// level images by pyrdown = 5x5 binomial blur with reflect border + bilinear resampling to (h/2, w/2), and a
// PS x PS bilinear sampling grid centred on the keypoint with half-extent s (in level-0 pixels), border clamped.
#include <math.h>

#include "common.h"
#include "prof.h"

namespace balf {
namespace {

constexpr int kPS = 32;

__device__ __forceinline__ int reflect_idx(int i, int n) {          // torch 'reflect': -1 -> 1, n -> n-2
    if (n == 1) return 0;
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

template <typename T>
__device__ __forceinline__ float px(const T *img, int i);
template <>
__device__ __forceinline__ float px<unsigned char>(const unsigned char *img, int i) { return (float)img[i] / 255.0f; }
template <>
__device__ __forceinline__ float px<float>(const float *img, int i) { return img[i]; }

// out[oy][ox] = bilinear(blur5x5(in)) at the align_corners=False source position of (oy, ox)
template <typename T>
__global__ __launch_bounds__(256) void pyrdown_kernel(const T *in, int h, int w, float *out, int ho, int wo) {
    const int ox = blockIdx.x * 16 + (threadIdx.x & 15), oy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (ox >= wo || oy >= ho) return;
    in += (size_t)blockIdx.z * h * w;                       // image of the batch
    out += (size_t)blockIdx.z * ho * wo;
    const float sy = fmaxf(((float)h / (float)ho) * ((float)oy + 0.5f) - 0.5f, 0.0f);
    const float sx = fmaxf(((float)w / (float)wo) * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float k[5] = {1.0f, 4.0f, 6.0f, 4.0f, 1.0f};
    auto blur = [&](int cy, int cx) {
        float acc = 0.0f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = reflect_idx(cy + dy, h);
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx)
                acc += (k[dy + 2] * k[dx + 2] * (1.0f / 256.0f)) * px<T>(in, yy * w + reflect_idx(cx + dx, w));
        }
        return acc;
    };
    const float v00 = blur(y0, x0), v01 = blur(y0, x1), v10 = blur(y1, x0), v11 = blur(y1, x1);
    out[oy * wo + ox] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
}

struct SampleArgs {
    const void *img;            // level image: uint8 (level 0) or float
    int h, w;                   // level size
    const float *xy;            // [B][K][2] level-0 pixel coordinates
    float *patches;             // [B][K][32][32]
    const int *count;           // [B] valid keypoints per image (nullptr: all K)
    int n;                      // K
    float s_l;                  // LAF scale at this level
    float xs, ys;               // level-0 -> level coordinate factors  (w_l - 1)/(w_0 - 1), (h_l - 1)/(h_0 - 1)
};

template <typename T>
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    const int b = blockIdx.y;
    const size_t p = (size_t)b * a.n + blockIdx.x;
    if (a.count && (int)blockIdx.x >= a.count[b]) {         // slot past the image's keypoint count: a zero patch
#pragma unroll
        for (int e = 0; e < 4; ++e) a.patches[p * (kPS * kPS) + threadIdx.x + 256 * e] = 0.0f;
        return;
    }
    const T *img = static_cast<const T *>(a.img) + (size_t)b * a.h * a.w;
    const float xc = a.xy[2 * p] * a.xs, yc = a.xy[2 * p + 1] * a.ys;
    const float wl1 = (float)(a.w - 1), hl1 = (float)(a.h - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = threadIdx.x + 256 * e, iy = i >> 5, ix = i & 31;
        const float bx = (2.0f * (float)ix + 1.0f) / (float)kPS - 1.0f, by = (2.0f * (float)iy + 1.0f) / (float)kPS - 1.0f;
        const float gx = 2.0f * (a.s_l * bx + xc) / wl1 - 1.0f, gy = 2.0f * (a.s_l * by + yc) / hl1 - 1.0f;
        // grid_sample(align_corners=False): pixel = ((g + 1) * size - 1) / 2, clipped to the border
        float fx = ((gx + 1.0f) * (float)a.w - 1.0f) * 0.5f, fy = ((gy + 1.0f) * (float)a.h - 1.0f) * 0.5f;
        fx = fminf(fmaxf(fx, 0.0f), wl1);
        fy = fminf(fmaxf(fy, 0.0f), hl1);
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const float lx = fx - (float)x0, ly = fy - (float)y0;
        const int x1 = x0 + 1 < a.w ? x0 + 1 : x0, y1 = y0 + 1 < a.h ? y0 + 1 : y0;     // weight is 0 when clamped
        const float v00 = px<T>(img, y0 * a.w + x0), v01 = px<T>(img, y0 * a.w + x1);
        const float v10 = px<T>(img, y1 * a.w + x0), v11 = px<T>(img, y1 * a.w + x1);
        a.patches[p * (kPS * kPS) + i] =
            (1.0f - ly) * (1.0f - lx) * v00 + (1.0f - ly) * lx * v01 + ly * (1.0f - lx) * v10 + ly * lx * v11;
    }
}

// PIL's Image.convert('L') (ITU-R 601-2 luma, fixed point): what demo_match.load_im produces (demo_match.py:13-19)
__global__ void rgb_to_gray_kernel(const unsigned char *rgb, unsigned char *gray, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        gray[i] = (unsigned char)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16);
    }
}

int pyramid_level(int h, int w, float scale) {
    const float s = sqrtf(scale * scale + 1e-10f);
    const int max_level = (h < w ? h : w) / kPS;
    float l = floorf(log2f(2.0f * s / (float)kPS));
    const float hi = (float)(max_level - 1 > 0 ? max_level - 1 : 0);
    l = l < 0.0f ? 0.0f : (l > hi ? hi : l);
    return (int)l;
}

}  // namespace
}  // namespace balf

using namespace balf;

extern "C" size_t balf_extract_patches_batch_workspace_bytes(int B, int H, int W, float scale) {
    if (B <= 0 || H <= 0 || W <= 0 || !(scale > 0.0f)) return 0;
    const int level = pyramid_level(H, W, scale);
    size_t bytes = 256;
    int h = H, w = W;
    for (int l = 0; l < level && h >= kPS && w >= kPS; ++l) {
        h /= 2; w /= 2;
        bytes += balf_align_up((size_t)B * h * w * 4, 256);
    }
    return bytes;
}

extern "C" int balf_extract_patches_batch(const unsigned char *gray_dev, int B, int H, int W, const float *xy_dev,
                                          const int32_t *count_dev, int K, float scale, float *patches_dev,
                                          void *workspace_dev, size_t workspace_bytes, void *stream) {
    if (!gray_dev || !xy_dev || !patches_dev || !workspace_dev || B <= 0 || K <= 0 || !(scale > 0.0f)) return BALF_ERR_ARG;
    if (H < 2 || W < 2 || B > 65535) return BALF_ERR_SHAPE;
    if (workspace_bytes < balf_extract_patches_batch_workspace_bytes(B, H, W, scale)) return BALF_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int level = pyramid_level(H, W, scale);
    const void *cur = gray_dev;
    int h = H, w = W, made = 0;
    char *ws = static_cast<char *>(workspace_dev);
    for (int l = 0; l < level && h >= kPS && w >= kPS; ++l) {
        const int ho = h / 2, wo = w / 2;
        float *out = reinterpret_cast<float *>(ws);
        const dim3 grid(balf_ceil_div(wo, 16), balf_ceil_div(ho, 16), B);
        if (made == 0)
            BALF_PROF(balf_prof::kPatchPyr, st,
                      (pyrdown_kernel<unsigned char><<<grid, 256, 0, st>>>(gray_dev, h, w, out, ho, wo)));
        else
            BALF_PROF(balf_prof::kPatchPyr, st,
                      (pyrdown_kernel<float><<<grid, 256, 0, st>>>(static_cast<const float *>(cur), h, w, out, ho, wo)));
        BALF_LAUNCH_CHECK();
        ws += balf_align_up((size_t)B * ho * wo * 4, 256);
        cur = out; h = ho; w = wo; ++made;
    }
    const float ms0 = (float)((H < W ? H : W) - 1), msl = (float)((h < w ? h : w) - 1);
    SampleArgs a{cur, h, w, xy_dev, patches_dev, count_dev, K, scale / ms0 * msl,
                 (float)(w - 1) / (float)(W - 1), (float)(h - 1) / (float)(H - 1)};
    const dim3 grid(K, B);
    if (made == 0)
        BALF_PROF(balf_prof::kPatchSample, st, (sample_kernel<unsigned char><<<grid, 256, 0, st>>>(a)));
    else
        BALF_PROF(balf_prof::kPatchSample, st, (sample_kernel<float><<<grid, 256, 0, st>>>(a)));
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}

extern "C" size_t balf_extract_patches_workspace_bytes(int H, int W, float scale) {
    return balf_extract_patches_batch_workspace_bytes(1, H, W, scale);
}

extern "C" int balf_extract_patches(const unsigned char *gray_dev, int H, int W, const float *xy_dev, int n_points,
                                    float scale, float *patches_dev, void *workspace_dev, size_t workspace_bytes,
                                    void *stream) {
    return balf_extract_patches_batch(gray_dev, 1, H, W, xy_dev, nullptr, n_points, scale, patches_dev, workspace_dev,
                                      workspace_bytes, stream);
}

extern "C" int balf_rgb_to_gray(const unsigned char *rgb_dev, long n_pixels, unsigned char *gray_dev, void *stream) {
    if (!rgb_dev || !gray_dev || n_pixels <= 0) return BALF_ERR_ARG;
    const long blocks = (n_pixels + 255) / 256;
    rgb_to_gray_kernel<<<(unsigned)(blocks < 65536 ? blocks : 65536), 256, 0, static_cast<hipStream_t>(stream)>>>(
        rgb_dev, gray_dev, n_pixels);
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}
