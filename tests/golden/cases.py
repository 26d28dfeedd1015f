"""Case tables shared by make_golden.py (which runs the reference) and the tests (which run the
oracle / the HIP path).  Inputs are regenerated from seeds on both sides; only the reference's
outputs are stored in the .npz fixtures."""
import numpy as np
import torch

WEIGHT_SEED = 20240

# name -> (batch, H, W, input seed)         H, W multiples of 64
FORWARD_SMALL = {
    "b2_64x64": (2, 64, 64, 11),
    "b1_128x192": (1, 128, 192, 12),
    "b1_256x320": (1, 256, 320, 13),
}
TAP_CASE = "b2_64x64"
# per-stage outputs of the two larger cases, sampled (stage_taps.npz): every 3rd row / 5th column of the stage's NCHW output --
# all in-cell offsets of the grid branch and all positions of the 8x8 blocks come up -- plus the float64 sum of every channel
# (every pixel enters)
TAP_SAMPLED = ("b1_128x192", "b1_256x320")


def stage_sample(v):
    """NCHW stage output -> (strided sample, per-channel float64 sums)"""
    v = np.asarray(v)
    return v[:, :, ::3, ::5].copy(), v.astype(np.float64).sum(axis=(0, 2, 3))

# name -> (h, w, K, synthetic image index)  un-padded image sizes, run through pad/crop
FORWARD_CFG = {
    "vga": (480, 640, 1000, 0),
    "720p": (720, 1280, 2000, 1),          # BASELINE configs[2]
    "1080p": (1080, 1920, 2000, 2),        # BASELINE configs[3]/[4]: fh = 136, fw = 240 in the stage-1 grid branch
}


def CFG_ROWS(hp):
    return np.array([0, 1, hp // 3, hp // 2, hp - 2, hp - 1])


def cfg_mix(prob):
    """One pixel per 8x8 cell at a cell-dependent offset, so that all 64 head channels are sampled (prob[::8, ::8] only
    ever sees channel 0)."""
    hc, wc = prob.shape[0] // 8, prob.shape[1] // 8
    i, j = np.mgrid[0:hc, 0:wc]
    return prob[8 * i + (3 * i + 5 * j) % 8, 8 * j + (i + 2 * j) % 8].copy()


def cfg_cellsum(prob):
    """float64 sum of each 8x8 cell (= 1 - dustbin probability): every pixel of the map enters."""
    hc, wc = prob.shape[0] // 8, prob.shape[1] // 8
    return prob.astype(np.float64).reshape(hc, 8, wc, 8).sum(axis=(1, 3))


def forward_input(b, h, w, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random((b, 3, h, w), dtype=np.float32))


# NMS / top-K cases on synthetic score maps
NMS_CASES = {
    "rand_120x160":     dict(h=120, w=160, seed=1, kind="rand", border=15, nms=15, k=100),
    "rand_480x640":     dict(h=480, w=640, seed=2, kind="rand", border=15, nms=15, k=1000),
    "ties_480x640":     dict(h=480, w=640, seed=3, kind="quant", border=15, nms=15, k=1000),
    "ties_coarse":      dict(h=200, w=333, seed=4, kind="quant8", border=15, nms=15, k=500),
    "zeros_96x128":     dict(h=96, w=128, seed=5, kind="zeros", border=15, nms=15, k=50),
    "sparse_few":       dict(h=150, w=170, seed=6, kind="sparse", border=15, nms=15, k=400),
    "const_plateau":    dict(h=90, w=110, seed=7, kind="const", border=15, nms=15, k=300),
    "nms3_odd":         dict(h=101, w=77, seed=8, kind="rand", border=0, nms=3, k=200),
    "nms4_even":        dict(h=64, w=96, seed=9, kind="quant8", border=2, nms=4, k=150),
    "nms16_even":       dict(h=128, w=128, seed=10, kind="rand", border=15, nms=16, k=64),
    "nms5_border1":     dict(h=70, w=70, seed=11, kind="quant", border=1, nms=5, k=4000),
    "rand_720p":        dict(h=720, w=1280, seed=12, kind="rand", border=15, nms=15, k=2000),
    "blobs_1080p":      dict(h=1080, w=1920, seed=13, kind="blobs", border=15, nms=15, k=2000),
    "k_eq_1":           dict(h=80, w=90, seed=14, kind="rand", border=15, nms=15, k=1),
    "thin_border_all":  dict(h=40, w=200, seed=15, kind="rand", border=20, nms=15, k=10),
}


def nms_input(spec):
    h, w, kind = spec["h"], spec["w"], spec["kind"]
    rng = np.random.default_rng(1000 + spec["seed"])
    if kind == "rand":
        return rng.random((h, w), dtype=np.float32)
    if kind == "quant":
        return (np.round(rng.random((h, w), dtype=np.float32) * 200.0) / 200.0).astype(np.float32)
    if kind == "quant8":
        return (np.round(rng.random((h, w), dtype=np.float32) * 8.0) / 8.0).astype(np.float32)
    if kind == "zeros":
        return np.zeros((h, w), np.float32)
    if kind == "const":
        return np.full((h, w), 0.25, np.float32)
    if kind == "sparse":
        m = np.zeros((h, w), np.float32)
        n = 60
        ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
        m[ys, xs] = rng.random(n, dtype=np.float32) + 0.01
        return m
    if kind == "blobs":      # smooth map, softmax-like dynamic range
        g = rng.random((h // 4 + 2, w // 4 + 2), dtype=np.float32)
        up = np.kron(g, np.ones((4, 4), np.float32))[:h, :w]
        noise = rng.random((h, w), dtype=np.float32) * 0.05
        return (np.exp(4.0 * up + noise) / np.exp(4.05) * 0.3).astype(np.float32)
    if kind == "ramp":       # strictly increasing along x: a long dependency chain for the greedy order
        return (np.arange(w, dtype=np.float32)[None, :] * 0.004 + np.arange(h, dtype=np.float32)[:, None] * 1e-5
                + 0.01).astype(np.float32)
    raise ValueError(kind)


PAD_SIZES = [(480, 640), (720, 1280), (1080, 1920), (481, 641), (64, 64), (100, 130), (511, 512), (65, 1)]


# Greedy SuperPoint NMS (nms_fast) cases: get_points_direct_from_score_map(subpixel=False)
GREEDY_CASES = {
    "g_rand_96x128":   dict(h=96, w=128, seed=21, kind="rand", border=15, conf=0.5, nms=15),
    "g_rand_dense":    dict(h=120, w=160, seed=22, kind="rand", border=15, conf=0.001, nms=15),
    "g_blobs_240x320": dict(h=240, w=320, seed=23, kind="blobs", border=15, conf=0.001, nms=15),
    "g_small_radius":  dict(h=100, w=90, seed=24, kind="rand", border=4, conf=0.2, nms=3),
    "g_sparse":        dict(h=150, w=170, seed=6, kind="sparse", border=15, conf=0.015, nms=15),
    "g_none":          dict(h=64, w=64, seed=25, kind="zeros", border=15, conf=0.001, nms=15),
    "g_ramp":          dict(h=80, w=200, seed=26, kind="ramp", border=0, conf=0.001, nms=8),
    "g_vga_blobs":     dict(h=480, w=640, seed=27, kind="blobs", border=15, conf=0.001, nms=15),
}


# sub-pixel refinement: (greedy case, patch size); 5 is the reference's default, 4 an even size (asymmetric patch)
SUBPIXEL_CASES = [("g_rand_96x128", 5), ("g_blobs_240x320", 5), ("g_small_radius", 5), ("g_blobs_240x320", 4), ("g_sparse", 3)]

# stand-alone nms_fast on a corner list: name -> (H, W, number of corners, dist_thresh, seed); float coordinates, distinct
# confidences, several corners per cell in the dense cases
NMS_FAST_CASES = {"c_sparse": (120, 160, 300, 4, 31), "c_dense": (60, 80, 3000, 4, 32), "c_wide": (200, 150, 2000, 9, 33),
                  "c_two": (40, 40, 2, 3, 34), "c_one": (40, 40, 1, 3, 35), "c_none": (40, 40, 0, 3, 36)}


def nms_fast_input(h, w, n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    x = rng.uniform(-0.49, w - 0.51, n)
    y = rng.uniform(-0.49, h - 0.51, n)
    c = rng.permutation(n).astype(np.float64) / max(n, 1) + 0.01             # distinct confidences
    return np.stack([x, y, c]) if n else np.zeros((3, 0))


# HardNet descriptor cases: name -> (number of patches, patch seed); weights = synthetic_hardnet_state_dict(HARDNET_SEED)
HARDNET_SEED = 515
HARDNET_CASES = {"n5": (5, 1), "n70": (70, 2)}
HARDNET_TAP_CASE = "n5"          # per-layer activations of patch 0, every 4th channel


# repeatability evaluation: name -> dict(ns, nd, seed, planted correspondences, kwargs of compute_repeatability)
REPEAT_CASES = {
    "small":      dict(ns=40, nd=55, seed=1, planted=25, kw={}),
    "train_eval": dict(ns=150, nd=130, seed=2, planted=90, kw={}),                                  # train_utils.py:189 defaults
    "hpatches":   dict(ns=300, nd=300, seed=3, planted=200, kw=dict(overlap_err=1 - 0.6, dist_match_thresh=5)),   # dataset_utils.py:332
    "no_match":   dict(ns=20, nd=20, seed=4, planted=0, kw={}),
    "crowded":    dict(ns=120, nd=120, seed=5, planted=120, kw=dict(overlap_err=0.7), spread=60.0),
}


def repeat_inputs(spec):
    """src rows (x, y, radius, score); dst = warped + jittered copies of some src points plus outliers."""
    rng = np.random.default_rng(9000 + spec["seed"])
    spread = spec.get("spread", 600.0)
    src = np.stack([rng.uniform(20, 20 + spread, spec["ns"]), rng.uniform(20, 20 + 0.75 * spread, spec["ns"]),
                    rng.uniform(0.6, 3.0, spec["ns"]), rng.uniform(0, 1, spec["ns"])], axis=1)
    dst = np.stack([rng.uniform(20, 20 + spread, spec["nd"]), rng.uniform(20, 20 + 0.75 * spread, spec["nd"]),
                    rng.uniform(0.6, 3.0, spec["nd"]), rng.uniform(0, 1, spec["nd"])], axis=1)
    k = min(spec["planted"], spec["ns"], spec["nd"])
    if k:
        pick_s, pick_d = rng.permutation(spec["ns"])[:k], rng.permutation(spec["nd"])[:k]
        dst[pick_d, :2] = src[pick_s, :2] + rng.normal(0, 4.0, (k, 2))
        dst[pick_d, 2] = src[pick_s, 2] * rng.uniform(0.7, 1.4, k)
    return src, dst


HOMOGRAPHY = np.array([[1.05, 0.08, -12.0], [-0.06, 0.97, 9.0], [1.2e-4, -0.8e-4, 1.0]])


# evaluation glue (train_utils.compute_repeatability_with_maximum_filter): two smooth score maps related by HOMOGRAPHY
EVAL_CASE = dict(h=240, w=320, seed=31, nms=15, num_points=300)


def eval_inputs(spec):
    """(src score map, dst score map, mask_src, mask_dst): dst is src resampled through HOMOGRAPHY (nearest
    neighbour) plus noise; the masks are rectangles standing in for the reference's cv2.warpPerspective masks."""
    h, w = spec["h"], spec["w"]
    rng = np.random.default_rng(500 + spec["seed"])
    g = rng.random((h // 8 + 2, w // 8 + 2))
    src = np.kron(g, np.ones((8, 8)))[:h, :w] * 0.5 + rng.random((h, w)) * 0.5
    hinv = np.linalg.inv(HOMOGRAPHY)
    ys, xs = np.mgrid[0:h, 0:w]
    # pixel (x, y) of dst shows the src pixel HOMOGRAPHY maps it to (HOMOGRAPHY = dst -> src)
    p = HOMOGRAPHY @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])
    sx = np.clip(np.rint(p[0] / p[2]), 0, w - 1).astype(int)
    sy = np.clip(np.rint(p[1] / p[2]), 0, h - 1).astype(int)
    dst = src[sy, sx].reshape(h, w) * 0.97 + rng.random((h, w)) * 0.03
    mask_src = np.zeros((h, w)); mask_src[20:h - 20, 25:w - 25] = 1.0
    mask_dst = np.zeros((h, w)); mask_dst[22:h - 18, 20:w - 30] = 1.0
    del hinv
    return src.astype(np.float32), dst.astype(np.float32), mask_src, mask_dst


# the demo's caller (demo/demo_match.py:21-57 `detect`), executed from the reference's source: name -> (h, w, image index,
# overrides of config.parse_test_config's defaults).  Inputs are uint8 RGB images as load_im returns them.
DETECT_ARGS = dict(border_size=15, nms_size=15, num_features=2048, s_mult=60, order_coord="xysr",
                   heatmap_confidence_threshold=0.001, sub_pixel=True, patch_size=4)
DETECT_CASES = {
    "d_240x320":     (240, 320, 40, {}),
    "d_nosub":       (240, 320, 40, dict(sub_pixel=False)),
    "d_odd_199x301": (199, 301, 41, dict(sub_pixel=False)),
    "d_few_yx":      (192, 256, 42, dict(sub_pixel=False, num_features=100, order_coord="yxsr")),
    "d_conf_high":   (192, 256, 42, dict(sub_pixel=False, heatmap_confidence_threshold=0.9999)),     # nothing passes
}


def detect_input(h, w, index):
    from balf_amd.utils import synth
    return np.stack([synth.synthetic_gray_u8(h, w, index + 7 * c, blur=5) for c in range(3)], axis=-1)


# the benchmark's caller (balf/utils/train_utils.py:416-454 `extract_detections`) beyond the FORWARD_CFG sizes:
# name -> (h, w, K, image index, border, nms)
EXTRACT_CASES = {
    "e_100x130": (100, 130, 50, 50, 15, 15),
    "e_odd_201x333_nms5": (201, 333, 300, 51, 4, 5),
}


# ---- natural images and a synthetic poster (round 5; fixtures in natural.npz) ----
# name -> (K, border, nms).  im1 / im2 are the two photographs the reference's demo runs on (/root/reference/media, decoded with
# PIL by make_golden.py and stored as uint8 arrays); the poster is generated below.
NATURAL_CASES = {"im1": (1000, 15, 15), "im2": (1000, 15, 15), "poster": (1000, 15, 15)}
NATURAL_FULL_PROB = ("im1", "poster")       # cases whose whole reference score map is stored (identical-input NMS)


def poster_u8():
    """480 x 640 RGB uint8 with what photographs rarely have and screenshots / documents always do: constant black, white and
    saturated-colour rectangles with hard edges (exact ties in the score map, NMS plateaus, all-zero neighbourhoods), a
    one-pixel checkerboard, one-pixel lines, a smooth ramp, and a patch of noise."""
    im = np.full((480, 640, 3), 128, np.uint8)
    im[0:160, 0:200] = 0                                            # black
    im[0:160, 200:400] = 255                                        # white
    im[0:160, 400:480] = (255, 0, 0)
    im[0:160, 480:560] = (0, 255, 0)
    im[0:160, 560:640] = (0, 0, 255)
    yy, xx = np.mgrid[0:160, 0:200]
    im[160:320, 0:200] = (((yy + xx) & 1) * 255).astype(np.uint8)[..., None]      # one-pixel checkerboard
    im[160:320, 200:400] = np.linspace(0, 255, 200).astype(np.uint8)[None, :, None]   # horizontal ramp
    im[160:320, 400:640] = 255
    im[160:320:16, 400:640] = 0                                     # one-pixel horizontal lines on white
    im[160:320, 400:640:16] = 0                                     # ... and vertical ones
    rng = np.random.default_rng(7)
    im[320:480, 0:200] = rng.integers(0, 256, (160, 200, 3), dtype=np.uint8)      # noise
    im[320:480, 200:400] = 0
    im[360:440, 240:360] = 255                                      # a white box on black
    im[320:480, 400:640] = (40, 40, 40)
    for k in range(6):                                              # dark squares of shrinking size
        s = 40 - 6 * k
        im[340:340 + s, 410 + 38 * k:410 + 38 * k + s] = (200, 180, 20)
    return im


POSTER_FLAT = {"black": (20, 140, 20, 180), "white": (20, 140, 220, 380)}   # interiors (y0, y1, x0, x1) of two constant rectangles
