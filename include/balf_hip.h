/*
 * balf_hip.h -- C ABI of libbalf_hip.so: the MI355X (gfx950) implementation of BALF's
 * keypoint-detection hot path (detector forward -> score map -> window-max NMS -> top-K).
 *
 * The reference is pure Python and has no FFI of its own; these entry points
 * are what a binding for this path replaces, one per reference call site:
 *
 *   balf_pack_weights        <- MLP_MA_DECODER.load_state_dict, reached from
 *                               balf/model/get_model.py:65-67 (load_test_pretrained_model)
 *   balf_forward             <- MLP_MA_DECODER.forward, balf/model/mlp_ma_decoder.py:278-285
 *                               (+ DetectorHead.forward, balf/model/decoder.py:16-30)
 *   balf_window_nms          <- remove_borders + apply_nms, balf/utils/test_utils.py:34-54
 *   balf_nms_topk            <- crop + remove_borders + apply_nms + get_point_coordinates /
 *                               find_index_higher_scores + final sort,
 *                               balf/utils/train_utils.py:437-452, balf/utils/test_utils.py:50-95
 *
 * Conventions: plain pointers and sizes only.  Every device buffer (inputs, outputs, workspace)
 * is owned by the caller; the library never allocates or frees device memory, never calls
 * hipDeviceSynchronize, and enqueues all work on the hipStream_t passed as `stream`
 * (as void*; NULL = the null stream).  No entry point synchronises the stream or reads anything back: where the amount
 * of work depends on the data (the rounds of balf_greedy_nms, the candidate pairs of balf_repeatability) the kernels
 * decide on the device, and a result that did not fit is reported in the outputs.  Every function returns BALF_OK (0)
 * or a negative BALF_ERR_* code and never throws.  Shapes are validated on the host before any launch.
 */
#ifndef BALF_HIP_H
#define BALF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BALF_ABI_VERSION 1

#define BALF_OK 0
#define BALF_ERR_ARG (-1)        /* null pointer, non-positive size, unsupported parameter        */
#define BALF_ERR_SHAPE (-2)      /* H/W not a multiple of 64, crop outside the padded map, K > H*W */
#define BALF_ERR_WORKSPACE (-3)  /* workspace smaller than *_workspace_bytes() says               */
#define BALF_ERR_ARCH (-4)       /* current device is not gfx950                                  */
#define BALF_ERR_LAUNCH (-5)     /* HIP reported an error at launch                               */

/* limits */
#define BALF_MAX_NMS_SIZE 32     /* window-max NMS footprint side (reference default 15)          */
#define BALF_MAX_TOPK 16384      /* num_points per image (reference configs use 1000..10000)      */

/* precision of the Linear-layer contractions (everything else is always fp32) */
#define BALF_PREC_FP32 0         /* v_mfma_f32_16x16x4_f32, exact fp32 fma chains                  */
#define BALF_PREC_FP16 1         /* f16 MFMA with SPLIT operands: every operand is carried as hi = f16(v) and
                                    lo = f16(v - hi), a product is hi*hi' + lo*hi' + hi*lo' accumulated in fp32
                                    (v_mfma_f32_32x32x16_f16 in the persistent kernels of the early stages,
                                    v_mfma_f32_16x16x32_f16 elsewhere; conv0 of stage 1 as exact f32 MFMA): score
                                    map within ~6e-6 of the fp32 reference.  Operands must stay inside the f16
                                    range (|v| < 6.5e4): see MLP_MA_DECODER.validate_fp16 on the Python side     */

int balf_abi_version(void);
const char *balf_error_string(int code);
/* BALF_OK iff the current HIP device is a gfx950 part. */
int balf_device_check(void);
/* "release BALF_ABLATE_GELU=0 ..." for the library as shipped; "DIAGNOSTIC ..." with the switch values when it was built
 * with any of the timing-ablation / instrumentation switches of csrc/diag.h (wrong results: never to be deployed). */
const char *balf_build_flags(void);

/* ---- weights ------------------------------------------------------------------------------
 * The 166 floating-point state-dict tensors of MLP_MA_DECODER in state_dict() order
 * (num_batches_tracked, the one int64 entry, is skipped).  balf_state_tensor_name/numel let a
 * binding check its table against the library's.  balf_pack_weights runs on the HOST: it
 * reads `n_tensors` host pointers (contiguous fp32, nn.Linear layout [out,in]) and writes the
 * packed blob (MFMA-fragment-ordered weights, BatchNorm folded into the head) that the caller
 * then copies to the device and passes to balf_forward as `packed_dev`. */
int balf_num_state_tensors(void);
const char *balf_state_tensor_name(int i);
size_t balf_state_tensor_numel(int i);
size_t balf_packed_weights_bytes(int precision);
int balf_pack_weights(const float *const *tensors, int n_tensors, int precision,
                      void *packed_host, size_t packed_bytes);

/* ---- detector forward ---------------------------------------------------------------------
 * x_nchw_dev : [B,3,Hp,Wp] fp32, Hp and Wp multiples of 64 (callers pad: test_utils.py:23-32), Hp * Wp <= 2^25 pixels per
 *              image (e.g. 5792 x 5792; BALF_ERR_SHAPE beyond: the kernels use 32-bit byte offsets inside an image)
 * logits_dev : [B,65,Hp/8,Wp/8] fp32 (post-BatchNorm, pre-softmax), may be NULL to skip
 * prob_dev   : [B,Hp,Wp] fp32 score map (softmax over 65, dustbin dropped, pixel-shuffled)
 * workspace  : balf_forward_workspace_bytes(B,Hp,Wp) bytes, 256-byte aligned */
size_t balf_forward_workspace_bytes(int B, int Hp, int Wp);
/* Images per launch: the forward walks a batch in micro-batches of this many images (the workspace holds one micro-batch:
 * 16 images at 1088x1920, more at smaller sizes; B if the batch is smaller).  0: bad arguments. */
int balf_forward_micro_batch(int B, int Hp, int Wp);
int balf_forward(const void *packed_dev, int precision, const float *x_nchw_dev, int B, int Hp, int Wp,
                 float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                 void *stream);

/* Same forward, fed with the raw uint8 image(s): gray [B,H,W] (channels = 1, replicated to the three input
 * channels) or RGB [B,H,W,3] (channels = 3).  The /255, make_shape_even and mod_padding_symmetric(64) of the
 * callers (demo/demo_match.py:22-29, balf/utils/test_utils.py:16-32) happen inside the first kernels; the
 * outputs have the PADDED size: Hp = H rounded up to even then to a multiple of 64 (same for W), the image at
 * rows (Hp-He)/2.., i.e. prob_dev [B,Hp,Wp], logits_dev [B,65,Hp/8,Wp/8], workspace for (B,Hp,Wp).  Results
 * are bit-identical to balf_forward on the host-prepared float input. */
int balf_forward_u8(const void *packed_dev, int precision, const unsigned char *image_dev, int channels, int B,
                    int H, int W, float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                    void *stream);

/* The same two forwards with a STATUS WORD BLOCK: status_dev -> 4 ints, caller-owned, writable by the device (device memory, or
 * pinned host memory mapped into the device's address space so that the host can look at it without a copy), zeroed by the
 * caller.  The split-f16 path (BALF_PREC_FP16) carries every MFMA operand as two f16 halves: a value beyond +-65504
 * saturates the high half and beyond ~1.3e5 turns into inf / NaN without any trap (csrc/split16.h).  With a status block the
 * kernels report that, at no cost while nothing happens -- the library only ever STORES 1 into a word, it never clears one:
 *   status[BALF_STATUS_SCORE] = 1   a pixel's softmax denominator was not a positive finite number (head kernel): the score
 *                                   map of this call holds non-finite values or garbage;
 *   status[BALF_STATUS_RANGE] = 1   a stage output (the next stage's input, the head's input or conv2's output) reached
 *                                   |v| >= 65504: its high half saturated -- the low half alone (11 bits) carries the
 *                                   excess, precision drops from 2^-20 to ~3e-4 relative, and beyond ~1.3e5 it overflows too;
 *   status[BALF_STATUS_SE]    = 1   a squeeze-excite pre-activation was not finite (SE kernel);
 *   status[3]                       reserved (never written).
 * The words are valid once the stream has passed the call.  status_dev may be NULL (then these are balf_forward /
 * balf_forward_u8).  The exact-fp32 path (BALF_PREC_FP32) has no operand range to leave: it reports BALF_STATUS_SCORE and
 * BALF_STATUS_SE (its SE kernel is the same one).  BALF_STATUS_RANGE covers STAGE BOUNDARIES only (a stage's output, the
 * head's input, conv2's output); an overflow inside a stage (token mix, dense2 inputs) is seen only if it reaches a
 * boundary, the SE pre-activation or the softmax denominator; NaNs do not raise RANGE (the comparison is false for them).
 * Host mirror: MLP_MA_DECODER checks the block lazily and re-runs a flagged batch on the fp32 kernels. */
#define BALF_STATUS_SCORE 0
#define BALF_STATUS_RANGE 1
#define BALF_STATUS_SE 2
#define BALF_STATUS_WORDS 4
int balf_forward_status(const void *packed_dev, int precision, const float *x_nchw_dev, int B, int Hp, int Wp,
                        float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                        int *status_dev, void *stream);
int balf_forward_u8_status(const void *packed_dev, int precision, const unsigned char *image_dev, int channels, int B,
                           int H, int W, float *logits_dev, float *prob_dev, void *workspace_dev, size_t workspace_bytes,
                           int *status_dev, void *stream);

/* Validation aid (not part of the data path): the activation that crosses stage boundary `stage` of the forward that last
 * ran on `workspace_dev` with the same (precision, B, Hp, Wp), as plain fp32 NHWC in out_dev:
 *   stage 1..3  Down.forward's return value of down1..down3 (mlp_ma_decoder.py:223-244: after MaxPool2d), i.e. the next
 *               stage's input: [B, Hp/2^s, Wp/2^s, C_s], C = 32/64/128 (the split-f16 path keeps these as hi+lo f16
 *               fragments in the workspace; the view adds the halves);
 *   stage 4     x2 = t * s + x1 + x0 of down4 BEFORE its conv2 (:239-241; conv2 runs inside the head kernel and its output
 *               never exists in memory): [B, Hp/8, Wp/8, 256] -- apply down4.conv2 to compare with the reference's down4.
 * Only the last micro-batch of a forward is resident in the workspace: B must not exceed it (BALF_ERR_ARG otherwise;
 * balf_forward_micro_batch: 16 images at 1088x1920, more at smaller sizes).  balf_forward_stage_view_numel = elements of out_dev (0: bad arguments). */
size_t balf_forward_stage_view_numel(int B, int Hp, int Wp, int stage);
int balf_forward_stage_view(int precision, const void *workspace_dev, size_t workspace_bytes, int B, int Hp, int Wp, int stage,
                            float *out_dev, void *stream);

/* ---- window-max NMS, dense form (apply_nms) ------------------------------------------------
 * score_dev [B,H,W] fp32 -> out_dev [B,H,W] fp32 = rb * (rb == max over the clipped
 * nms_size x nms_size window), rb = score with a `border`-pixel frame zeroed. */
int balf_window_nms(const float *score_dev, int B, int H, int W, int border, int nms_size,
                    float *out_dev, void *stream);

/* ---- crop + border + NMS + exact top-K ------------------------------------------------------
 * prob_dev [B,Hp,Wp]; the score map of image b is prob[b, crop_y:crop_y+H, crop_x:crop_x+W].
 * Outputs per image: idx_dev[b,0:count] flat indices y*W+x of the first K pixels in raster
 * order whose NMS score >= the K-th largest NMS score (reference fallback when that is <= 0),
 * emitted sorted by (score descending, index ascending); score_dev their scores;
 * entries past count are idx -1 / score 0.  K <= H*W (the reference raises IndexError
 * otherwise) and K <= BALF_MAX_TOPK.  Scores are probabilities (>= +0): with negative scores in the map the points
 * with a positive score are still selected exactly as the reference selects them, but the reference's <= 0 fallback
 * then returns -0.0 and negative window maxima (its NMS map is x * (x == max)), which this entry point does not
 * reproduce (it returns the first K raster pixels with score +0). */
size_t balf_nms_topk_workspace_bytes(int B, int H, int W, int K);
int balf_nms_topk(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                  int border, int nms_size, int K, int32_t *idx_dev, float *score_dev,
                  int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* The same selection with a caller-given threshold instead of the K-th largest score: `threshold != -1` of
 * find_index_higher_scores (/root/reference/balf/utils/test_utils.py:74-95): the first K pixels in raster order
 * with NMS score >= threshold (all of them if fewer reach it; count may be 0).  threshold must be > 0 and finite:
 * with a threshold <= 0 every pixel qualifies and the answer is the first K raster indices -- no device work, the
 * host mirror (balf_amd/utils/test_utils.py) returns those directly.  Same workspace as balf_nms_topk. */
int balf_nms_threshold(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W,
                       int border, int nms_size, float threshold, int K, int32_t *idx_dev, float *score_dev,
                       int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- greedy "SuperPoint" NMS of the demo path --------------------------------------
 * Replaces get_points_direct_from_score_map + nms_fast (+ soft_argmax_points), balf/utils/test_utils.py:97-215,
 * as called by demo/demo_match.py:45-57.  The score map of image b is prob[b, crop_y:+H, crop_x:+W] with a
 * `border` frame zeroed; candidates = pixels >= conf_thresh (> 0); a candidate is kept iff no higher-scoring kept
 * candidate lies within Chebyshev distance dist_thresh (<= 16).  Output rows sorted by score descending (flat
 * index ascending among equal scores): idx_dev[B,K] (-1 padded), score_dev[B,K], count_dev[B] = rows returned
 * (<= K), total_dev[B] = points kept before truncation (may be NULL).  subpixel_patch > 0 additionally writes
 * xy_dev[B,K,2] = (x, y) refined by the patch soft-argmax.  The number of suppression rounds depends on the data
 * (4-8 on score maps, ~W / dist_thresh on a monotone ramp): a fixed number of rounds is enqueued, each returning at
 * once when nothing is alive, and a per-image kernel finishes whatever is left -- exact for every input, stream-ordered. */
size_t balf_greedy_nms_workspace_bytes(int B, int H, int W, int K);
int balf_greedy_nms(const float *prob_dev, int B, int Hp, int Wp, int crop_y, int crop_x, int H, int W, int border,
                    float conf_thresh, int dist_thresh, int K, int subpixel_patch, int32_t *idx_dev,
                    float *score_dev, float *xy_dev, int32_t *count_dev, int32_t *total_dev, void *workspace_dev,
                    size_t workspace_bytes, void *stream);

/* ---- HardNet patch descriptor of the demo path ----------------------------------------
 * Replaces HardNet.load_state_dict / HardNet.forward, third_party/hardnet/hardnet_pytorch.py:31-72, as called by
 * demo/demo_match.py:72-93,131-134.  The 21 floating-point state tensors in state_dict() order are, per
 * convolution i in features.{0,3,6,9,12,15,19}: weight [Cout,Cin,k,k], then the following BatchNorm's running_mean
 * and running_var (num_batches_tracked is skipped).  balf_hardnet_pack_weights runs on the HOST (BatchNorm folded,
 * split-f16 MFMA fragment order); the caller copies the blob to the device.
 * balf_hardnet_forward: patches_dev [N,32,32] fp32 (the [N,1,32,32] tensor HardNet.forward takes) ->
 * desc_dev [N,128] fp32, L2-normalised.  Contractions run on v_mfma_f32_16x16x32_f16 with split (hi+lo) operands,
 * fp32 accumulation: descriptors agree with the fp32 reference to ~1e-5. */
int balf_hardnet_num_state_tensors(void);
const char *balf_hardnet_state_tensor_name(int i);
size_t balf_hardnet_state_tensor_numel(int i);
size_t balf_hardnet_packed_weights_bytes(void);
int balf_hardnet_pack_weights(const float *const *tensors, int n_tensors, void *packed_host, size_t packed_bytes);
size_t balf_hardnet_workspace_bytes(int n_patches);
int balf_hardnet_forward(const void *packed_dev, const float *patches_dev, int n_patches, float *desc_dev,
                         void *workspace_dev, size_t workspace_bytes, void *stream);
/* Masked form for fixed-size keypoint slots: the patches are n_patches / group images x `group` slots, and only the
 * first count_dev[image] slots of each image are computed (the others cost nothing and get zero descriptors).
 * count_dev == NULL: as balf_hardnet_forward. */
int balf_hardnet_forward_masked(const void *packed_dev, const float *patches_dev, int n_patches, int group,
                                const int32_t *count_dev, float *desc_dev, void *workspace_dev, size_t workspace_bytes,
                                void *stream);
/* Same with a choice of operand precision: BALF_HARDNET_SPLIT_F16 (default everywhere else: hi+lo operands, three
 * products, descriptors within ~1e-6 of fp32) or BALF_HARDNET_PLAIN_F16 (one f16 product, activations stored as one
 * plane: about twice as fast, descriptors within ~5e-4 -- the accuracy class of the TF32 convolutions PyTorch uses
 * by default for this network on an NVIDIA GPU). */
#define BALF_HARDNET_SPLIT_F16 0
#define BALF_HARDNET_PLAIN_F16 1
int balf_hardnet_forward_ex(const void *packed_dev, const float *patches_dev, int n_patches, int group,
                            const int32_t *count_dev, int precision, float *desc_dev, void *workspace_dev,
                            size_t workspace_bytes, void *stream);

/* ---- patch extraction and descriptor matching of the demo path -------------------------
 * balf_extract_patches replaces kornia.feature.laf_from_center_scale_ori + extract_patches_from_pyramid(PS=32) as
 * called by demo/demo_match.py:62-70: gray_dev uint8 [H,W]; xy_dev [N,2] keypoint (x, y) in pixels; `scale` = the
 * LAF scale (args.s_mult) shared by all keypoints; patches_dev [N,32,32] fp32 in [0,1].
 * balf_match_smnn replaces kornia.feature.match_smnn(desc1, desc2, th) (demo_match.py:104-110): descriptors
 * [n,128] fp32; outputs idx_dev [min(n1,n2),2] (index in desc1, index in desc2; -1 padded, sorted by the first),
 * dist_dev [min(n1,n2)] (max of the two nearest/second-nearest distance ratios), count_dev [1].
 * kornia is a third-party dependency that is absent offline: both follow its published algorithm and are checked
 * against this repo's restatement only (parity unpinned). */
size_t balf_extract_patches_workspace_bytes(int H, int W, float scale);
int balf_extract_patches(const unsigned char *gray_dev, int H, int W, const float *xy_dev, int n_points, float scale,
                         float *patches_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
/* Batched form: gray_dev [B,H,W], xy_dev [B,K,2], count_dev [B] = keypoints actually present per image (NULL: all
 * K; slots past the count get zero patches), patches_dev [B,K,32,32].  One pyramid per image, one launch for all. */
size_t balf_extract_patches_batch_workspace_bytes(int B, int H, int W, float scale);
int balf_extract_patches_batch(const unsigned char *gray_dev, int B, int H, int W, const float *xy_dev,
                               const int32_t *count_dev, int K, float scale, float *patches_dev, void *workspace_dev,
                               size_t workspace_bytes, void *stream);
/* uint8 RGB (interleaved, n_pixels x 3) -> uint8 gray exactly as PIL's Image.convert('L') (demo_match.py:13-19). */
int balf_rgb_to_gray(const unsigned char *rgb_dev, long n_pixels, unsigned char *gray_dev, void *stream);
size_t balf_match_smnn_workspace_bytes(int n1, int n2);
/* Batched form: `pairs` independent problems, desc1_dev [pairs,k1,128] / desc2_dev [pairs,k2,128] with n1_dev / n2_dev
 * [pairs] valid rows each; idx_dev [pairs,min(k1,k2),2], dist_dev [pairs,min(k1,k2)], count_dev [pairs]. */
size_t balf_match_smnn_batch_workspace_bytes(int pairs, int k1, int k2);
int balf_match_smnn_batch(const float *desc1_dev, int k1, const int32_t *n1_dev, const float *desc2_dev, int k2,
                          const int32_t *n2_dev, int pairs, float th, int32_t *idx_dev, float *dist_dev,
                          int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
int balf_match_smnn(const float *desc1_dev, int n1, const float *desc2_dev, int n2, float th, int32_t *idx_dev,
                    float *dist_dev, int32_t *count_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* ---- repeatability evaluation ---------------------------------------------------------
 * balf_repeatability replaces compute_repeatability, balf/benchmark_test/repeatability_tools.py:379-490 (callers:
 * balf/utils/train_utils.py:189,257, balf/datasets/dataset_utils.py:332).  src_dev [ns,3] / dst_dev [nd,3] float64
 * rows (x, y, radius).  Outputs: counts_dev[3] = {num_points_single_scale, num_points_multi_scale,
 * possible_matches}; errors_dev[2] = the two sums of (1 - overlap) over the assigned pairs, in assignment order;
 * corr_s_dev / corr_m_dev [min(ns,nd),2] = (dst index, src index) per assigned pair in assignment order, -1 padded.
 * Among exactly equal overlaps the pair with the lower flat index ns-major wins (the reference's order there is
 * NumPy's unstable argsort).  max_edges bounds the number of pairs whose overlap reaches 1 - overlap_err, per scale;
 * that number is known on the device only: when a scale's list does not fit, its entry of counts_dev is -1 and its
 * correspondences are all -1 (the host mirror raises when it reads that) -- nothing waits for the device.
 * ns, nd <= 65536.
 * balf_apply_homography replaces apply_homography_to_points, balf/benchmark_test/geometry_tools.py:43-86:
 * points_dev [n,4] float64 rows (x, y, radius, score), h_dev[9] row-major -> out_dev [n,4]. */
size_t balf_repeatability_workspace_bytes(int ns, int nd, int max_edges);
int balf_repeatability(const double *src_dev, int ns, const double *dst_dev, int nd, double overlap_err, double eps,
                       double dist_match_thresh, double radius_size, int max_edges, int32_t *counts_dev,
                       double *errors_dev, int32_t *corr_s_dev, int32_t *corr_m_dev, void *workspace_dev,
                       size_t workspace_bytes, void *stream);
int balf_apply_homography(const double *points_dev, int n, const double *h_dev, double *out_dev, void *stream);

/* create_common_region_masks, balf/benchmark_test/geometry_tools.py:7-26 (callers: the HPatches evaluation, and the
 * masks handed to compute_repeatability_with_maximum_filter, balf/utils/train_utils.py:170-196): the region of each
 * image covered by the other under the homography h_dst_2_src (HOST pointer, 9 doubles row-major).  mask_src_dev
 * [h_src, w_src] and mask_dst_dev [h_dst, w_dst] float64 in {0, 1}: cv2.warpPerspective of an all-ones image with a
 * zeroed `border` frame (bilinear, zero outside, source coordinates rounded to 1/32 pixel), >= 0.75, frame zeroed
 * again.  cv2 is absent from the build container: parity with it is unpinned. */
int balf_common_region_masks(const double *h_dst_2_src_host, int h_src, int w_src, int h_dst, int w_dst, int border,
                             double *mask_src_dev, double *mask_dst_dev, void *stream);

/* ---- measurement aid (not part of the data path) ---------------------------------------------
 * Between balf_profile_begin() and balf_profile_end() every kernel launch of the library is bracketed
 * by a hipEvent pair on its launch stream.  balf_profile_end() waits for those events and returns,
 * per slot (balf_profile_slot_name), the summed device time in ms and the number of launches.
 * Global state, not re-entrant; bench.py uses it for the per-kernel roofline figures. */
int balf_profile_num_slots(void);
const char *balf_profile_slot_name(int slot);
int balf_profile_begin(void);
int balf_profile_end(float *ms_total, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* BALF_HIP_H */
