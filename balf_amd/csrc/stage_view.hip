// balf_forward_stage_view: the activations that cross the stage boundaries of the detector forward, copied out of the
// forward's workspace as plain fp32 NHWC tensors.  A validation aid, not part of the data path: the reference's forward
// returns only {'logits', 'prob'} (/root/reference/balf/model/mlp_ma_decoder.py:278-285), but its four Down stages are
// separately observable modules (forward hooks on down1..down4), and a wrong kernel should fail at the stage that has it.
#include "det_common.h"
#include "split16.h"

namespace balf {
namespace {

// fragment formats of the split-f16 path (detector_f16.hip: store_frag_px; stage1_f16.h: store_frag32) -> value
__device__ __forceinline__ float frag_value(const char *base, long pix, int C, int c, bool fmt32) {
    const char *p;
    int lo_off;
    if (fmt32) {            // per pixel and K-step of 16 channels 64 B: [hi: lane half 0, 1 x 8 halves][lo: ...]
        const int s = c >> 4, cc = c & 15, g2 = cc >> 3, h = (cc & 7) >> 2, r = cc & 3;
        p = base + pix * (long)C * 4 + s * 64 + h * 16 + (4 * g2 + r) * 2;
        lo_off = 32;
    } else {                // per pixel and K-step of 32 channels 128 B: [hi: lane quarters 0..3 x 8 halves][lo: ...]
        const int ks = c >> 5, cc = c & 31, t = cc >> 4, q = (cc & 15) >> 2, r = cc & 3;
        p = base + pix * (long)C * 4 + ks * 128 + q * 16 + (4 * t + r) * 2;
        lo_off = 64;
    }
    return (float)*reinterpret_cast<const _Float16 *>(p) + (float)*reinterpret_cast<const _Float16 *>(p + lo_off);
}

__global__ void view_frag_kernel(const char *X, long npix, int C, int fmt32, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    out[i] = frag_value(X, i / C, C, (int)(i % C), fmt32 != 0);
}

__global__ void view_x2_kernel(const float *T, const float *R, const float *scale, long per_img, long n, int C, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long img = (i / C) / per_img;
    out[i] = fmaf(T[i], scale[img * C + c], R[i]);
}

}  // namespace
}  // namespace balf

extern "C" size_t balf_forward_stage_view_numel(int B, int Hp, int Wp, int stage) {
    if (B <= 0 || Hp <= 0 || Wp <= 0 || Hp % 64 || Wp % 64 || stage < 1 || stage > 4) return 0;
    const int sh = stage < 4 ? stage : 3;                       // stages 1-3 pool 2x2, stage 4 keeps its resolution
    return (size_t)B * (Hp >> sh) * (Wp >> sh) * balf::kC[stage - 1];
}

extern "C" int balf_forward_stage_view(int precision, const void *workspace_dev, size_t workspace_bytes, int B, int Hp, int Wp,
                                       int stage, float *out_dev, void *stream) {
    using namespace balf;
    if (!workspace_dev || !out_dev || B <= 0 || stage < 1 || stage > 4) return BALF_ERR_ARG;
    if (precision != BALF_PREC_FP32 && precision != BALF_PREC_FP16) return BALF_ERR_ARG;
    if (Hp <= 0 || Wp <= 0 || Hp % 64 || Wp % 64) return BALF_ERR_SHAPE;
    const Plan pl = make_plan(B, Hp, Wp);
    if (workspace_bytes < pl.total) return BALF_ERR_WORKSPACE;
    if (B > pl.mb) return BALF_ERR_ARG;                          // only the last micro-batch of a forward is resident
    const char *ws = static_cast<const char *>(workspace_dev);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = balf_forward_stage_view_numel(B, Hp, Wp, stage);
    const int C = kC[stage - 1];
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (stage < 4) {
        const char *X = ws + pl.off_X[stage - 1];
        if (precision == BALF_PREC_FP32) {
            if (hipMemcpyAsync(out_dev, X, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return BALF_ERR_LAUNCH;
        } else {
            hipLaunchKernelGGL(view_frag_kernel, dim3(blocks), dim3(256), 0, st, X, (long)(n / C), C, kFmt32[stage] ? 1 : 0, out_dev);
        }
    } else {
        const long per_img = (long)(Hp / 8) * (Wp / 8);
        hipLaunchKernelGGL(view_x2_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float *>(ws + pl.off_T),
                           reinterpret_cast<const float *>(ws + pl.off_R), reinterpret_cast<const float *>(ws + pl.off_scale),
                           per_img, (long)n, C, out_dev);
    }
    BALF_LAUNCH_CHECK();
    return BALF_OK;
}
