"""Host-side mirror of the reference's inference helpers, backed by the HIP library.

Same names, argument meaning and error behaviour as /root/reference/balf/utils/test_utils.py
(``make_shape_even:16``, ``mod_padding_symmetric:23``, ``remove_borders:34``, ``apply_nms:50``,
``get_point_coordinates:56``, ``find_index_higher_scores:74``) so callers such as
``train_utils.extract_detections`` (/root/reference/balf/utils/train_utils.py:416-454) can import
this module instead.  NumPy arrays in, NumPy arrays out; the window-max NMS and the top-K selection
run on the GPU and there is no CPU fallback -- without a GPU these functions raise.
The pad/border helpers only move bytes and stay on the host, as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch
import yaml

from .. import ops
from .._lib import BalfHipError


def get_cfg_from_yaml_file(cfg_file):
    with open(cfg_file, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def make_shape_even(image):
    h, w = image.shape[0], image.shape[1]
    return np.pad(image, ((0, h % 2), (0, w % 2), (0, 0)), mode="constant", constant_values=0)


def mod_padding_symmetric(image, factor=64):
    h, w = image.shape[0], image.shape[1]
    ph = (h // factor + 1) * factor - h if h % factor != 0 else 0
    pw = (w // factor + 1) * factor - w if w % factor != 0 else 0
    return np.pad(image, ((ph // 2, ph // 2), (pw // 2, pw // 2), (0, 0)), mode="constant", constant_values=0)


def remove_borders(image, borders):
    out = np.zeros_like(image)
    h, w = image.shape[0], image.shape[1]
    out[borders:h - borders, borders:w - borders] = image[borders:h - borders, borders:w - borders]
    return out


def _device():
    if not torch.cuda.is_available():
        raise BalfHipError("no GPU visible: balf_amd.utils.test_utils has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    from .._lib import require_mi355x
    require_mi355x(dev)
    return dev


def _as_map(score_map) -> np.ndarray:
    m = np.ascontiguousarray(score_map, dtype=np.float32)
    if m.ndim != 2:
        raise ValueError("score map must be [H, W]")
    return m


def apply_nms(score_map, size):
    m = _as_map(score_map)
    t = torch.from_numpy(m).to(_device()).unsqueeze(0)
    return ops.window_nms(t, 0, int(size))[0].cpu().numpy()


def find_index_higher_scores(map, num_points=1000, threshold=-1):
    """[row, col] pairs, in raster order, of the first ``num_points`` pixels whose value reaches the
    ``num_points``-th largest value of the map (with the reference's <= 0 fallback), or -- ``threshold != -1`` --
    the given threshold (/root/reference/balf/utils/test_utils.py:74-95)."""
    m = _as_map(map)
    h, w = m.shape
    if threshold != -1 and not threshold > 0:
        # `map >= threshold` holds everywhere on a non-negative score map: the first num_points raster pixels
        # (test_utils.py:91-95); index arithmetic only, nothing to compute
        flat = np.flatnonzero(m.ravel() >= threshold)[: int(num_points)]
        return np.stack([flat // w, flat % w], axis=1)
    t = torch.from_numpy(m).to(_device()).unsqueeze(0)
    # a 1x1 window keeps every pixel, so this is the threshold selection alone (K-th largest value, or the given one)
    idx, _, cnt = ops.nms_topk(t, 0, 0, h, w, 0, 1, min(int(num_points), h * w) if threshold != -1 else int(num_points),
                               threshold=float(threshold))
    flat = np.sort(idx[0, : int(cnt[0])].cpu().numpy().astype(np.int64))
    return np.stack([flat // w, flat % w], axis=1)


def get_point_coordinates(map, scale_value=1., num_points=1000, threshold=-1, order_coord='xysr'):
    m = _as_map(map)
    ind = find_index_higher_scores(m, num_points=num_points, threshold=threshold)
    sc = m[ind[:, 0], ind[:, 1]].astype(np.float64)
    out = np.empty((ind.shape[0], 4), dtype=np.float64)
    if order_coord == 'xysr':
        out[:, 0], out[:, 1] = ind[:, 1], ind[:, 0]
    elif order_coord == 'yxsr':
        out[:, 0], out[:, 1] = ind[:, 0], ind[:, 1]
    else:
        raise ValueError(order_coord)
    out[:, 2], out[:, 3] = scale_value, sc
    return out


def get_points_direct_from_score_map(heatmap, conf_thresh=0.015, nms_size=15, subpixel=True, patch_size=5,
                                     scale_value=1., order_coord='xysr'):
    """Mirror of the demo's post-processing (/root/reference/balf/utils/test_utils.py:97-128): confidence
    threshold, greedy ``nms_fast`` with a (2*nms_size+1)^2 suppression window, sort by confidence, optional
    sub-pixel soft-argmax -> rows ``[x, y, scale, score]`` float64 (``np.zeros((0, 4))`` when nothing passes).
    Among exactly equal scores the raster-first point wins (the reference's order there is NumPy's unstable
    argsort).  The sub-pixel step is pinned against the reference's own code around its one torchgeometry call (that call:
    documented definition, unpinned; tests/golden/subpixel.npz)."""
    m = _as_map(heatmap)
    h, w = m.shape
    t = torch.from_numpy(m).to(_device()).unsqueeze(0)
    k = min(ops._lib.MAX_TOPK, h * w)
    idx, score, xy, count, total = ops.greedy_nms(t, 0, 0, h, w, 0, float(conf_thresh), int(nms_size), k,
                                                  int(patch_size) if subpixel else 0)
    n = int(count[0])
    if int(total[0]) > n:
        raise NotImplementedError(f"{int(total[0])} points survive the NMS, more than the {k} the kernel returns")
    if n == 0:
        return np.zeros((0, 4))
    i = idx[0, :n].cpu().numpy().astype(np.int64)
    out = np.empty((n, 4), dtype=np.float64)
    if subpixel:
        p = xy[0, :n].cpu().numpy().astype(np.float64)
        xs, ys = p[:, 0], p[:, 1]
    else:
        xs, ys = (i % w).astype(np.float64), (i // w).astype(np.float64)
    if order_coord == 'xysr':
        out[:, 0], out[:, 1] = xs, ys
    elif order_coord == 'yxsr':
        out[:, 0], out[:, 1] = ys, xs
    else:
        raise ValueError(order_coord)
    out[:, 2], out[:, 3] = scale_value, score[0, :n].cpu().numpy()
    return out


def nms_fast(in_corners, H, W, dist_thresh):
    """Mirror of the stand-alone greedy NMS on a corner list (/root/reference/balf/utils/test_utils.py:130-168,
    SuperPoint's ``nms_fast``): ``in_corners`` is ``3 x N`` ``[x, y, confidence]``; returns ``(out 3 x M, out_inds M)``
    -- the surviving corners sorted by confidence and their indices into ``in_corners``.

    The sweep itself runs in ``balf_greedy_nms`` on a rasterised map whose values are the corners' RANKS in the
    confidence order (exact in fp32 up to 2^24 corners, so that the suppression order is the reference's order whatever
    the confidences' precision); everything else is index bookkeeping on the host, quirks included: coordinates are
    rounded half-to-even, a cell holding several corners takes part with the best of them but REPORTS the worst one
    (the reference's ``inds`` grid keeps the last writer).  Among exactly equal confidences the order is the stable one
    (NumPy's default argsort, which the reference uses, is unstable there)."""
    in_corners = np.asarray(in_corners)
    n = in_corners.shape[1]
    inds1 = np.argsort(-in_corners[2, :], kind="stable")
    corners = in_corners[:, inds1]
    rcorners = corners[:2, :].round().astype(int)
    if n == 0:
        return np.zeros((3, 0)).astype(int), np.zeros(0).astype(int)
    if n == 1:
        return np.vstack((rcorners, in_corners[2])).reshape(3, 1), np.zeros((1)).astype(int)
    if rcorners[0].min() < 0 or rcorners[0].max() >= W or rcorners[1].min() < 0 or rcorners[1].max() >= H:
        raise IndexError("corner outside the H x W grid")
    if n >= 1 << 24:
        raise NotImplementedError("more than 2^24 corners")
    flat = rcorners[1].astype(np.int64) * W + rcorners[0]
    rank_map = np.zeros(H * W, dtype=np.float32)
    np.maximum.at(rank_map, flat, np.arange(n, 0, -1, dtype=np.float32))    # a cell takes part with its best corner: n - position
    inds = np.zeros(H * W, dtype=np.int64)
    np.maximum.at(inds, flat, np.arange(n))                                 # ... and reports its worst one (the reference's last writer)
    t = torch.from_numpy(rank_map.reshape(1, H, W)).to(_device())
    k = min(ops._lib.MAX_TOPK, H * W)
    idx, _, _, count, total = ops.greedy_nms(t, 0, 0, H, W, 0, 0.5, int(dist_thresh), k, 0)
    m = int(count[0])
    if int(total[0]) > m:
        raise NotImplementedError(f"{int(total[0])} corners survive, more than the {k} the kernel returns")
    keep = np.sort(idx[0, :m].cpu().numpy().astype(np.int64))               # raster order, as np.where gives it
    inds_keep = inds[keep]
    out = corners[:, inds_keep]
    inds2 = np.argsort(-out[-1, :], kind="stable")
    return out[:, inds2], inds1[inds_keep[inds2]]
