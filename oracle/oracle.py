"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``balf_amd`` never does (the HIP path fails loudly instead of falling
back here).

A restatement, in plain torch/NumPy, of the BALF keypoint-detection hot path:

* detector forward  -- /root/reference/balf/model/mlp_ma_decoder.py:223-285,
  /root/reference/balf/model/decoder.py:16-30, /root/reference/balf/utils/tensor_op.py:15-27
* pad / crop / border -- /root/reference/balf/utils/test_utils.py:16-47,
  /root/reference/balf/utils/train_utils.py:437-442
* window-max NMS      -- /root/reference/balf/utils/test_utils.py:50-54
* K-th-threshold top-K -- /root/reference/balf/utils/test_utils.py:56-95,
  /root/reference/balf/utils/train_utils.py:451-452

Parity pinning: ``tests/golden/*.npz`` were produced by ``tests/golden/make_golden.py``,
which imports the *reference itself* from /root/reference in the build container, loads
the same seeded synthetic weights, and records its outputs.  ``tests/test_oracle_golden.py``
checks every function here against those vectors (no GPU needed).  The arithmetic itself
lives in third-party libraries the reference does not pin (torch, einops, scipy:
requirements.txt:1,9,13); this file calls the same torch primitives for Linear /
LayerNorm / GELU / softmax and re-derives the einops rearranges and SciPy's
``maximum_filter`` by hand.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_RSH = "residual_split_head_multi_axis_gmlp_layer"
_RCAB = "residual_channel_attention_block"


# ----------------------------------------------------------------------------------------
# detector forward
# ----------------------------------------------------------------------------------------
def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _gmlp_branch(sd, p, unit, z, grid: bool):
    """GridGmlpLayer / BlockGmlpLayer (mlp_ma_decoder.py:57-70, 104-117) on NHWC ``z``.

    The reference rearranges to [n, groups, tokens, c], mixes, and rearranges back; here the
    64x64 token mix is applied in place on a 6-D view."""
    n, h, w, c = z.shape
    t = F.gelu(_lin(sd, p + ".dense1", _ln(sd, p + ".norm", z)))
    a, b = t[..., :c], t[..., c:]
    b = _ln(sd, f"{p}.{unit}.norm", b)
    w4 = sd[f"{p}.{unit}.dense.weight"].reshape(8, 8, 8, 8)
    b2 = sd[f"{p}.{unit}.dense.bias"].reshape(8, 8)
    if grid:   # token = which of the 8x8 image regions; (iy, ix) inside the region is batch
        b6 = b.reshape(n, 8, h // 8, 8, w // 8, c)                 # n gy iy gx ix c
        mix = torch.einsum("pqgh,ngihjc->npiqjc", w4, b6) + b2[None, :, None, :, None, None]
    else:      # token = position inside each contiguous 8x8 block
        b6 = b.reshape(n, h // 8, 8, w // 8, 8, c)                 # n by iy bx ix c
        mix = torch.einsum("pqgh,nygxhc->nypxqc", w4, b6) + b2[None, None, :, None, :, None]
    mix = mix.reshape(n, h, w, c)
    return z + _lin(sd, p + ".dense2", a * (mix + 1.0))


def stage_forward(sd: Dict[str, torch.Tensor], d: str, x_nhwc: torch.Tensor, last: bool,
                  taps: Dict[str, torch.Tensor] = None) -> torch.Tensor:
    """One ``Down`` stage (mlp_ma_decoder.py:223-244) on NHWC input, NHWC output."""
    c = sd[f"{d}.conv.0.weight"].shape[0]
    x0 = F.relu(_lin(sd, f"{d}.conv.0", x_nhwc))
    q = f"{d}.{_RSH}"
    y = F.gelu(_lin(sd, q + ".dense1", _ln(sd, q + ".norm", x0)))
    u, v = y[..., :c], y[..., c:]
    u = _gmlp_branch(sd, q + ".grid_gmlp_layer", "grid_gating_unit", u, True)
    v = _gmlp_branch(sd, q + ".block_gmlp_layer", "block_gating_unit", v, False)
    x1 = _lin(sd, q + ".dense2", torch.cat([u, v], dim=-1)) + x0
    r = f"{d}.{_RCAB}"
    t = _lin(sd, r + ".conv2", F.leaky_relu(_lin(sd, r + ".conv1", _ln(sd, r + ".norm", x1)), 0.2))
    m = t.mean(dim=(1, 2))                                          # CALayer squeeze (:166)
    s = torch.sigmoid(_lin(sd, r + ".calayer.excite.2", F.relu(_lin(sd, r + ".calayer.excite.0", m))))
    x2 = t * s[:, None, None, :] + x1 + x0
    if taps is not None:
        taps[d + ".u"], taps[d + ".x1"], taps[d + ".t"], taps[d + ".s"], taps[d + ".x2"] = u, x1, t, s, x2
    if last:
        return _lin(sd, f"{d}.conv2", x2)
    n, h, w, _ = x2.shape
    return x2.reshape(n, h // 2, 2, w // 2, 2, c).amax(dim=(2, 4))   # MaxPool2d(2) (:219,236)


def detector_forward(sd: Dict[str, torch.Tensor], x_nchw: torch.Tensor,
                     taps: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """``MLP_MA_DECODER.forward`` (mlp_ma_decoder.py:278-285): NCHW float input with H, W
    multiples of 64 -> {'logits': [B,65,H/8,W/8], 'prob': [B,H,W]}.  Computes in the dtype of
    ``x_nchw``/``sd`` (fp32 to match the reference, fp64 for a tighter yardstick)."""
    if x_nchw.shape[-1] % 64 or x_nchw.shape[-2] % 64:
        raise ValueError("H and W must be multiples of 64")
    x = x_nchw.permute(0, 2, 3, 1)
    for i in range(4):
        x = stage_forward(sd, f"down{i + 1}", x, last=(i == 3), taps=taps)
    z = _lin(sd, "detector_head.dense", F.relu(x))
    hp = "detector_head.norm."
    z = (z - sd[hp + "running_mean"]) / torch.sqrt(sd[hp + "running_var"] + 1e-5) * sd[hp + "weight"] + sd[hp + "bias"]
    p = torch.softmax(z, dim=-1)[..., :64]                          # drop dustbin (decoder.py:25)
    n, h, w, _ = p.shape
    prob = p.reshape(n, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(n, h * 8, w * 8)
    return {"logits": z.permute(0, 3, 1, 2).contiguous(), "prob": prob.contiguous()}


def cast_state(sd: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------
# pre / post processing (NumPy)
# ----------------------------------------------------------------------------------------
def make_shape_even(image: np.ndarray) -> np.ndarray:
    h, w = image.shape[:2]
    return np.pad(image, ((0, h & 1), (0, w & 1), (0, 0)), mode="constant")


def mod_padding_symmetric(image: np.ndarray, factor: int = 64) -> np.ndarray:
    h, w = image.shape[:2]
    ph = ((h + factor) // factor) * factor - h if h % factor else 0
    pw = ((w + factor) // factor) * factor - w if w % factor else 0
    return np.pad(image, ((ph // 2, ph // 2), (pw // 2, pw // 2), (0, 0)), mode="constant")


def crop_offsets(h: int, w: int, hp: int, wp: int) -> Tuple[int, int]:
    he, we = h + (h & 1), w + (w & 1)
    return hp // 2 - he // 2, wp // 2 - we // 2


def remove_borders(score: np.ndarray, b: int) -> np.ndarray:
    out = np.zeros_like(score)
    h, w = score.shape[:2]
    out[b:h - b, b:w - b] = score[b:h - b, b:w - b]
    return out


def window_max(score: np.ndarray, size: int) -> np.ndarray:
    """Clipped-window maximum over rows/cols [i - size//2, i + (size-1)//2]: what SciPy's
    ``maximum_filter(footprint=ones((size,size)))`` with its default ``mode='reflect'``
    computes (reflected samples always lie inside the clipped window)."""
    lo, hi = size // 2, (size - 1) // 2
    h, w = score.shape
    neg = np.array(-np.inf, dtype=score.dtype)
    p = np.pad(score, ((lo, hi), (0, 0)), mode="constant", constant_values=neg)
    m = p[0:h]
    for k in range(1, size):
        m = np.maximum(m, p[k:k + h])
    p = np.pad(m, ((0, 0), (lo, hi)), mode="constant", constant_values=neg)
    m2 = p[:, 0:w]
    for k in range(1, size):
        m2 = np.maximum(m2, p[:, k:k + w])
    return m2


def apply_nms(score: np.ndarray, size: int) -> np.ndarray:
    return score * (score == window_max(score, size))


def topk_threshold(nms: np.ndarray, k: int) -> float:
    """K-th largest value of the whole map with the reference's fallback when it is <= 0
    (test_utils.py:78-89).  ``k > nms.size`` is an IndexError there and here."""
    flat = nms.ravel()
    if k > flat.size:
        raise IndexError("num_points exceeds the number of pixels")
    thr = np.partition(flat, flat.size - k)[flat.size - k]
    if thr <= 0.0:
        pos = flat[flat > 0.0]
        thr = pos.min() if pos.size else flat.dtype.type(0.0)
    return thr


def select_topk(nms: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Flat indices (row-major, ascending) and scores of the first ``k`` pixels in raster order
    with ``nms >= thr`` (test_utils.py:93-95)."""
    thr = topk_threshold(nms, k)
    idx = np.flatnonzero(nms.ravel() >= thr)[:k]
    return idx.astype(np.int64), nms.ravel()[idx]


def select_threshold(nms: np.ndarray, k: int, threshold: float) -> np.ndarray:
    """``find_index_higher_scores(map, num_points=k, threshold=threshold)`` with ``threshold != -1``
    (test_utils.py:91-95): flat indices, in raster order, of the first k pixels with value >= threshold."""
    return np.flatnonzero(np.asarray(nms).ravel() >= threshold)[:k]


def canonical_order(idx: np.ndarray, score: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Score descending, flat index ascending among equal scores.  The reference's final
    ``argsort(-score)`` (train_utils.py:451) leaves the order of equal scores unspecified;
    this is the canonical representative the HIP path emits."""
    o = np.lexsort((idx, -score.astype(np.float64)))
    return idx[o], score[o]


def detect_from_prob(prob_pad: np.ndarray, h: int, w: int, border: int, nms_size: int, k: int):
    """crop -> remove_borders -> apply_nms -> top-K on one padded score map
    (train_utils.py:437-452).  Returns canonical (idx, score)."""
    top, left = crop_offsets(h, w, *prob_pad.shape)
    score = remove_borders(prob_pad[top:top + h, left:left + w], border)
    idx, sc = select_topk(apply_nms(score, nms_size), k)
    return canonical_order(idx, sc)


def points_xysr(idx: np.ndarray, score: np.ndarray, w: int) -> np.ndarray:
    """Rows ``[x, y, 1.0, score]`` float64, the reference's 'xysr' layout (test_utils.py:63-64)."""
    out = np.empty((idx.size, 4), dtype=np.float64)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = idx % w, idx // w, 1.0, score
    return out


def extract_detections(sd, image_rgb_norm: np.ndarray, nms_size=15, num_points=25, border_size=15):
    """Whole single-image pipeline (train_utils.py:416-454) on the CPU."""
    h, w = image_rgb_norm.shape[:2]
    pad = mod_padding_symmetric(make_shape_even(image_rgb_norm), 64)
    x = torch.tensor(pad, dtype=torch.float32).permute(2, 0, 1).unsqueeze(0)
    with torch.no_grad():
        prob = detector_forward(sd, x)["prob"][0].numpy()
    idx, sc = detect_from_prob(prob, h, w, border_size, nms_size, num_points)
    return points_xysr(idx, sc, w), prob


# ----------------------------------------------------------------------------------------
# demo post-processing: greedy nms_fast (+ sub-pixel) -- SURVEY 8f row f1
# ----------------------------------------------------------------------------------------
def greedy_nms(heatmap: np.ndarray, conf_thresh: float, dist_thresh: int) -> Tuple[np.ndarray, np.ndarray]:
    """``get_points_direct_from_score_map(subpixel=False)`` (test_utils.py:97-168): candidates >= conf_thresh are
    visited in descending score order (raster-first among equal scores); one is kept iff no kept candidate lies
    within Chebyshev distance ``dist_thresh``.  Returns (flat idx, score) sorted by score descending."""
    h, w = heatmap.shape
    ys, xs = np.where(heatmap >= conf_thresh)
    if xs.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, heatmap.dtype)
    sc = heatmap[ys, xs]
    order = np.lexsort((ys * w + xs, -sc.astype(np.float64)))
    blocked = np.zeros((h + 2 * dist_thresh, w + 2 * dist_thresh), bool)
    cand = np.zeros_like(blocked)
    cand[ys + dist_thresh, xs + dist_thresh] = True
    keep = []
    for j in order:
        y, x = ys[j] + dist_thresh, xs[j] + dist_thresh
        if cand[y, x] and not blocked[y, x]:
            keep.append(j)
            blocked[y - dist_thresh:y + dist_thresh + 1, x - dist_thresh:x + dist_thresh + 1] = True
    keep = np.asarray(keep, np.int64)
    return (ys[keep] * w + xs[keep]).astype(np.int64), sc[keep]


def soft_argmax_refine(heatmap: np.ndarray, idx: np.ndarray, patch: int) -> np.ndarray:
    """``soft_argmax_points`` (test_utils.py:170-215) by its definition: the patch, normalised by its sum + 1e-6,
    log-ed and soft-max-ed (= the patch re-normalised), gives the expected (x, y); p += E - patch//2.
    Pinned (tests/golden/subpixel.npz) against the reference's own code run around a restatement of its one torchgeometry
    call, SpatialSoftArgmax2d (not installed in the build container; that call itself stays unpinned)."""
    h, w = heatmap.shape
    pad = patch // 2
    hp = np.pad(heatmap.astype(np.float64), pad, mode="constant")
    out = np.zeros((idx.size, 2))
    for n, p in enumerate(idx):
        y, x = int(p) // w, int(p) % w
        pt = hp[y:y + patch, x:x + patch]
        s = pt.sum()
        gx, gy = np.meshgrid(np.arange(patch), np.arange(patch))
        out[n] = (x + (pt * gx).sum() / s - pad, y + (pt * gy).sum() / s - pad)
    return out


def demo_detect(sd, im_rgb_u8: np.ndarray, border_size=15, nms_size=15, num_features=2048, conf_thresh=0.001,
                sub_pixel=True, patch_size=4, order_coord="xysr", prob_pad: np.ndarray = None):
    """``demo_match.detect`` (demo/demo_match.py:21-57): /255 -> pad -> forward -> crop -> remove_borders ->
    get_points_direct_from_score_map (threshold, greedy NMS, optional sub-pixel) -> strongest num_features rows (x, y, 1).
    ``prob_pad`` replaces the forward (identical-input checks).  The reference returns a PAIR of empty arrays when nothing
    passes (:51-52); so does this."""
    h, w = im_rgb_u8.shape[:2]
    if prob_pad is None:
        pad = mod_padding_symmetric(make_shape_even(im_rgb_u8 / 255.), 64)
        x = torch.tensor(pad, dtype=torch.float32).permute(2, 0, 1).unsqueeze(0)
        with torch.no_grad():
            prob_pad = detector_forward(sd, x)["prob"][0].numpy()
    top, left = crop_offsets(h, w, *prob_pad.shape)
    heat = remove_borders(prob_pad[top:top + h, left:left + w], border_size)
    idx, sc = greedy_nms(heat, conf_thresh, nms_size)
    if idx.size == 0:
        return np.zeros([0, 3]), np.zeros([0, 1])
    xy = soft_argmax_refine(heat, idx, patch_size) if sub_pixel else np.stack([idx % w, idx // w], axis=1).astype(np.float64)
    xy = xy[:num_features]
    if order_coord == "yxsr":
        xy = xy[:, ::-1]
    return np.concatenate([xy, np.ones((xy.shape[0], 1))], axis=1)


# ----------------------------------------------------------------------------------------
# demo descriptor / matching path
# ----------------------------------------------------------------------------------------
_HARDNET_CONVS = ((0, 1, 1), (3, 1, 1), (6, 2, 1), (9, 1, 1), (12, 2, 1), (15, 1, 1), (19, 1, 0))   # (index, stride, padding)


def hardnet_forward(sd: Dict[str, torch.Tensor], patches: torch.Tensor, taps: dict = None) -> torch.Tensor:
    """HardNet descriptor of [N,1,32,32] patches -> [N,128], L2-normalised.
    /root/reference/third_party/hardnet/hardnet_pytorch.py:58-72: per-patch (mean, unbiased std + 1e-7)
    normalisation, 6 x [conv3x3 (no bias) -> BatchNorm(eval, affine=False, eps 1e-5) -> ReLU], Dropout (identity in
    eval), conv8x8 -> BatchNorm, then x / sqrt(sum x^2 + 1e-10) (:6-15).  Pinned by tests/golden/hardnet.npz."""
    x = patches.to(sd["features.0.weight"].dtype)
    flat = x.reshape(x.shape[0], -1)
    mp = flat.mean(dim=1)
    sp = flat.std(dim=1) + 1e-7
    x = (x - mp.view(-1, 1, 1, 1)) / sp.view(-1, 1, 1, 1)
    for j, (idx, stride, pad) in enumerate(_HARDNET_CONVS):
        x = F.conv2d(x, sd[f"features.{idx}.weight"], None, stride=stride, padding=pad)
        mean, var = sd[f"features.{idx + 1}.running_mean"], sd[f"features.{idx + 1}.running_var"]
        x = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5)
        if j < 6:
            x = torch.relu(x)
        if taps is not None:
            taps[j] = x
    x = x.reshape(x.shape[0], -1)
    return x / torch.sqrt((x * x).sum(dim=1) + 1e-10).unsqueeze(-1)


def pyrdown(img: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.transform.pyrdown (PARITY UNPINNED: kornia is not installed offline): 5x5 binomial blur
    ([1,4,6,4,1] x [1,4,6,4,1] / 256, reflect border) followed by bilinear resampling to (H//2, W//2),
    align_corners=False.  img [B,C,H,W]."""
    k1 = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0], dtype=img.dtype)
    k = (k1[:, None] * k1[None, :] / 256.0).view(1, 1, 5, 5)
    b, c, h, w = img.shape
    xp = F.pad(img, (2, 2, 2, 2), mode="reflect")
    blur = F.conv2d(xp.reshape(b * c, 1, h + 4, w + 4), k).reshape(b, c, h, w)
    return F.interpolate(blur, size=(h // 2, w // 2), mode="bilinear", align_corners=False)


def extract_patches(gray: torch.Tensor, xy: torch.Tensor, scale: float, ps: int = 32) -> torch.Tensor:
    """What demo_match.extract_features:62-70 gets from kornia: laf_from_center_scale_ori(xy, scale, 0) ->
    extract_patches_from_pyramid(img, laf, PS).  PARITY UNPINNED (kornia absent); restated from kornia's
    published source (kornia/feature/laf.py, 0.6-0.7 series):
      * pyramid level = clamp(floor(log2(2 * scale / PS)), 0, max(0, min(H, W) // PS - 1)), level l image =
        pyrdown applied l times (the loop also stops once a level is smaller than PS);
      * LAF [[s,0,x],[0,s,y]] normalised by (min(H,W)-1 | W-1 | H-1) of the full image and de-normalised with the
        level's size; sampling grid = affine_grid(LAF, PS x PS, align_corners=False), i.e. base coordinates
        (2i + 1)/PS - 1, mapped to [-1,1] by 2 g / (w_l - 1) - 1 and sampled with
        grid_sample(bilinear, padding_mode='border', align_corners=False).
    gray [H,W] float in [0,1]; xy [N,2] (x, y) pixel coordinates.  Returns [N,1,ps,ps]."""
    img = gray.view(1, 1, *gray.shape).to(torch.float32)
    n = xy.shape[0]
    h0, w0 = gray.shape
    max_level = min(h0, w0) // ps
    level = int(np.clip(np.floor(np.log2(2.0 * np.sqrt(scale * scale + 1e-10) / ps)), 0.0, max(0, max_level - 1)))
    cur = img
    for _ in range(level):
        if min(cur.shape[2], cur.shape[3]) < ps:
            break
        cur = pyrdown(cur)
    hl, wl = cur.shape[2], cur.shape[3]
    ms0, msl = float(min(h0 - 1, w0 - 1)), float(min(hl - 1, wl - 1))
    s_l = scale / ms0 * msl
    x_l = xy[:, 0].to(torch.float32) / float(w0 - 1) * float(wl - 1)
    y_l = xy[:, 1].to(torch.float32) / float(h0 - 1) * float(hl - 1)
    base = (2.0 * torch.arange(ps, dtype=torch.float32) + 1.0) / ps - 1.0
    gx = s_l * base.view(1, 1, ps) + x_l.view(n, 1, 1)                  # [n,1,ps]
    gy = s_l * base.view(1, ps, 1) + y_l.view(n, 1, 1)                  # [n,ps,1]
    grid = torch.stack([(2.0 * gx / float(wl - 1) - 1.0).expand(n, ps, ps),
                        (2.0 * gy / float(hl - 1) - 1.0).expand(n, ps, ps)], dim=-1)
    return F.grid_sample(cur.expand(n, 1, hl, wl), grid, mode="bilinear", padding_mode="border", align_corners=False)


def match_snn(d1: torch.Tensor, d2: torch.Tensor, th: float):
    """kornia.feature.match_snn restated (PARITY UNPINNED): Euclidean distance matrix, the two smallest per row,
    keep rows with d_first / d_second <= th.  Returns (ratio [M], idx [M,2])."""
    if d2.shape[0] < 2 or d1.shape[0] == 0:
        return torch.zeros(0, dtype=d1.dtype), torch.zeros(0, 2, dtype=torch.int64)
    dm = torch.cdist(d1.double(), d2.double()).to(d1.dtype)
    vals, idx = torch.topk(dm, 2, dim=1, largest=False)
    ratio = vals[:, 0] / vals[:, 1]
    mask = ratio <= th
    i1 = torch.arange(d1.shape[0])[mask]
    return ratio[mask], torch.stack([i1, idx[:, 0][mask]], dim=1)


def match_smnn(d1: torch.Tensor, d2: torch.Tensor, th: float = 0.99):
    """kornia.feature.match_smnn restated (demo_match.py:105-107; PARITY UNPINNED): ratio-test matches in both
    directions, keep the mutual ones; distance = max of the two ratios; sorted by the index in d1."""
    r1, m1 = match_snn(d1, d2, th)
    r2, m2 = match_snn(d2, d1, th)
    if len(r1) == 0 or len(r2) == 0:
        return torch.zeros(0, dtype=d1.dtype), torch.zeros(0, 2, dtype=torch.int64)
    back = {int(j): (int(i), float(r)) for (j, i), r in zip(m2.tolist(), r2.tolist())}    # d2 index -> (d1 index, ratio)
    out_i, out_r = [], []
    for (i, j), r in zip(m1.tolist(), r1.tolist()):
        if j in back and back[j][0] == i:
            out_i.append((i, j))
            out_r.append(max(r, back[j][1]))
    if not out_i:
        return torch.zeros(0, dtype=d1.dtype), torch.zeros(0, 2, dtype=torch.int64)
    return torch.tensor(out_r, dtype=d1.dtype), torch.tensor(out_i, dtype=torch.int64)


# ----------------------------------------------------------------------------------------
# repeatability evaluation
# ----------------------------------------------------------------------------------------
def _circle_intersection(R, r, d):
    """/root/reference/balf/benchmark_test/repeatability_tools.py:492-508, vectorised (float64)."""
    R, r, d = np.broadcast_arrays(np.asarray(R, np.float64), np.asarray(r, np.float64), np.asarray(d, np.float64))
    out = np.zeros(d.shape, np.float64)
    inside = d <= np.abs(R - r)
    out[inside] = np.pi * np.minimum(R, r)[inside] ** 2
    mid = ~inside & ~(d >= r + R)
    r2, R2, d2 = r[mid] ** 2, R[mid] ** 2, d[mid] ** 2
    alpha = np.arccos((d2 + r2 - R2) / (2 * d[mid] * r[mid]))
    beta = np.arccos((d2 + R2 - r2) / (2 * d[mid] * R[mid]))
    out[mid] = r2 * alpha + R2 * beta - 0.5 * (r2 * np.sin(2 * alpha) + R2 * np.sin(2 * beta))
    return out


def _greedy_assign(overlaps: np.ndarray, thr: float):
    """Greedy one-to-one assignment in descending overlap (repeatability_tools.py:424-441): ties broken by flat
    index (the reference's order among exactly equal overlaps is NumPy's unstable argsort)."""
    n_dst = overlaps.shape[1]
    flat = overlaps.ravel()
    cand = np.flatnonzero(flat >= thr)
    order = cand[np.lexsort((cand, -flat[cand]))]
    y_vis = np.zeros(overlaps.shape[0], bool)
    x_vis = np.zeros(n_dst, bool)
    found, err, corr = 0, 0.0, []
    for idx in order:
        y, x = idx // n_dst, idx % n_dst
        if x_vis[x] or y_vis[y]:
            continue
        found += 1
        err += 1 - flat[idx]
        corr.append([x, y])
        x_vis[x] = y_vis[y] = True
    return found, err, np.asarray(corr)


def compute_repeatability(src, dst, overlap_err=0.4, eps=1e-6, dist_match_thresh=3, radious_size=30.0):
    """/root/reference/balf/benchmark_test/repeatability_tools.py:379-490 restated with NumPy: src/dst rows
    (x, y, radius, ...).  Pinned by tests/golden/repeatability.npz (recorded by executing the reference's own
    function bodies, extracted with ``ast`` because the module imports torchvision)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    ns, nd = len(src), len(dst)
    dx = src[:, None, 0] - dst[None, :, 0]
    dy = src[:, None, 1] - dst[None, :, 1]
    dist = (dx ** 2 + dy ** 2) ** 0.5
    possible = int((dist <= dist_match_thresh).any(axis=1).sum()) if nd else 0
    near = dist <= 4 * radious_size
    rr, rd = np.broadcast_arrays(src[:, None, 2], dst[None, :, 2])
    factor = radious_size / (np.maximum(rr, rd) + np.finfo(float).eps)
    inter = _circle_intersection(factor * rr, factor * rd, dist)
    union = np.pi * (factor * rr) ** 2 + np.pi * (factor * rd) ** 2 - inter + eps
    multi = np.where(near, inter / union, 0.0)
    inter = _circle_intersection(radious_size, radious_size, dist)
    union = np.pi * radious_size ** 2 + np.pi * radious_size ** 2 - inter + eps
    single = np.where(near, inter / union, 0.0)
    thr = 1 - overlap_err
    fs, es, cs = _greedy_assign(single, thr)
    fm, em, cm = _greedy_assign(multi, thr)
    points = min(ns, nd)
    return {"rep_single_scale": fs / np.asarray(points, float) * 100.0,
            "rep_multi_scale": fm / np.asarray(points, float) * 100.0,
            "num_points_single_scale": fs, "num_points_multi_scale": fm,
            "error_overlap_single_scale": 0.0 if fs == 0 else es / float(fs + np.finfo(float).eps),
            "error_overlap_multi_scale": 0.0 if fm == 0 else em / float(fm + np.finfo(float).eps),
            "total_num_points": points, "correspondences": cs, "possible_matches": possible,
            "correspondences_m": cm}


def apply_homography_to_points(points, h):
    """/root/reference/balf/benchmark_test/geometry_tools.py:43-86 restated in closed form: (x, y) through the
    homography; the radius is scaled by the local affine approximation A of the warp.  The reference builds
    B = inv(A (r^2 + eps32) I A^T) and returns 1 / sqrt(sqrt(eig1 eig2)) = sqrt((r^2 + eps32) |det A|)."""
    pts = np.asarray(points, np.float64)
    if len(pts) == 0:
        return np.zeros((0, 4))
    h = np.asarray(h, np.float64)
    x, y = pts[:, 0], pts[:, 1]
    den = h[2, 0] * x + h[2, 1] * y + h[2, 2]
    nx = h[0, 0] * x + h[0, 1] * y + h[0, 2]
    ny = h[1, 0] * x + h[1, 1] * y + h[1, 2]
    fxdx = h[0, 0] / den - nx * h[2, 0] / den ** 2
    fxdy = h[0, 1] / den - nx * h[2, 1] / den ** 2
    fydx = h[1, 0] / den - ny * h[2, 0] / den ** 2
    fydy = h[1, 1] / den - ny * h[2, 1] / den ** 2
    tmp = pts[:, 2] ** 2 + np.finfo(np.float32).eps
    rad = np.sqrt(tmp * np.abs(fxdx * fydy - fxdy * fydx))
    return np.stack([nx / den, ny / den, rad, pts[:, 3]], axis=1)


def compute_repeatability_with_maximum_filter(src_scores, dst_scores, homography, mask_src, mask_dst, nms_size, num_points):
    """/root/reference/balf/utils/train_utils.py:170-196 from the pieces above."""
    pts = []
    for score, mask in ((src_scores, mask_src), (dst_scores, mask_dst)):
        nms = np.multiply(apply_nms(np.asarray(score), nms_size), mask)
        idx, sc = select_topk(nms, num_points)                       # raster order, like argwhere
        pts.append(points_xysr(idx, sc, nms.shape[1]))
    r = compute_repeatability(pts[0], apply_homography_to_points(pts[1], homography))
    return ([r["rep_single_scale"]], [r["rep_multi_scale"]], [r["error_overlap_single_scale"]],
            [r["error_overlap_multi_scale"]], [r["possible_matches"]])


# ----------------------------------------------------------------------------------------
# common-region masks of the evaluation (geometry_tools.py:7-26)
# ----------------------------------------------------------------------------------------
def invert3(m: np.ndarray) -> np.ndarray:
    """Closed-form 3x3 inverse (adjugate / determinant) in individually rounded fp64 operations -- the form OpenCV's
    cv::invert takes for n <= 3, and operation for operation what balf_common_region_masks computes on the host."""
    a, b, c, d, e, f, g, h, i = (float(v) for v in np.asarray(m, dtype=np.float64).reshape(9))
    A, B, C = e * i - f * h, -(d * i - f * g), d * h - e * g
    det = a * A + b * B + c * C
    if det == 0.0:
        raise np.linalg.LinAlgError("singular homography")
    r = 1.0 / det
    return np.array([[A * r, -(b * i - c * h) * r, (b * f - c * e) * r],
                     [B * r, (a * i - c * g) * r, -(a * f - c * d) * r],
                     [C * r, -(a * h - b * g) * r, (a * e - b * d) * r]], dtype=np.float64)


def warp_perspective_linear(src: np.ndarray, m: np.ndarray, dsize) -> np.ndarray:
    """cv2.warpPerspective(src, m, dsize) with its default flags (bilinear, BORDER_CONSTANT 0), restated from OpenCV's
    algorithm for a float64 image: dst(x, y) = src(M^-1 (x, y, 1)); source coordinates rounded to 1/32 pixel
    (INTER_TAB_SIZE 32, round half to even).  cv2 is not installed in the build container: this restatement is
    PARITY UNPINNED against cv2 itself."""
    w, h = int(dsize[0]), int(dsize[1])
    mi = invert3(m)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    wv = mi[2, 0] * xs + mi[2, 1] * ys + mi[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(wv != 0, 32.0 / wv, 0.0)
    fx = np.clip((mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2]) * sc, -2147483648.0, 2147483647.0)
    fy = np.clip((mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]) * sc, -2147483648.0, 2147483647.0)
    X, Y = np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)
    sx, sy = X >> 5, Y >> 5
    ax, ay = (X & 31) / 32.0, (Y & 31) / 32.0
    sh, sw = src.shape

    def at(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        return np.where(ok, src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], 0.0)

    return (at(sy, sx) * ((1 - ax) * (1 - ay)) + at(sy, sx + 1) * (ax * (1 - ay)) +
            at(sy + 1, sx) * ((1 - ax) * ay) + at(sy + 1, sx + 1) * (ax * ay))


def create_common_region_masks(h_dst_2_src, shape_src, shape_dst, numpy_inverse: bool = True):
    """geometry_tools.py:7-26 with ``warp_perspective_linear`` in the place of cv2.warpPerspective.
    numpy_inverse=True (default) is the reference's own statement, ``inv_h = np.linalg.inv(h_dst_2_src)`` (:9, LAPACK's LU);
    False takes the closed-form ``invert3`` there too -- operation for operation what balf_common_region_masks computes, so
    that kernel and oracle can be compared for EQUALITY.  The two inverses differ in the last bits, which can move a source
    coordinate across a 1/32-pixel rounding tie: a handful of mask pixels at most (tests/test_repeat_gpu.py checks both)."""
    h_dst_2_src = np.asarray(h_dst_2_src, dtype=np.float64)
    inv_h = np.linalg.inv(h_dst_2_src) if numpy_inverse else invert3(h_dst_2_src)
    inv_h = inv_h / inv_h[2, 2]
    ones_dst = remove_borders(np.ones((shape_dst[0], shape_dst[1])), 15)
    mask_src = warp_perspective_linear(ones_dst, h_dst_2_src, (shape_src[1], shape_src[0]))
    mask_src = remove_borders(np.where(mask_src >= 0.75, 1.0, 0.0), 15)
    ones_src = remove_borders(np.ones((shape_src[0], shape_src[1])), 15)
    mask_dst = warp_perspective_linear(ones_src, inv_h, (shape_dst[1], shape_dst[0]))
    mask_dst = remove_borders(np.where(mask_dst >= 0.75, 1.0, 0.0), 15)
    return mask_src, mask_dst
