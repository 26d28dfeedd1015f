"""``compute_repeatability`` on the GPU: same arguments and result dict as
/root/reference/balf/benchmark_test/repeatability_tools.py:379-490 (callers: train_utils.py:189,257,
dataset_utils.py:332).  The reference's Ns x Nd Python double loop, two dense overlap matrices and their argsorts
become one call into ``balf_repeatability`` (include/balf_hip.h); float64 throughout.  No CPU path.

``apply_nms`` of the same reference module (:19-23) is the window-max NMS: use ``balf_amd.utils.test_utils.apply_nms``.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops
from .._lib import BalfHipError, check, current_stream_ptr, lib

MAX_EDGES = 1 << 22


def _device():
    if not torch.cuda.is_available():
        raise BalfHipError("balf_amd has no CPU path: compute_repeatability needs the GPU")
    dev = torch.device("cuda", torch.cuda.current_device())
    from .._lib import require_mi355x
    require_mi355x(dev)
    return dev


def compute_repeatability(src_indexes, dst_indexes, overlap_err=0.4, eps=1e-6, dist_match_thresh=3, radious_size=30.):
    src = np.asarray(src_indexes, dtype=np.float64)
    dst = np.asarray(dst_indexes, dtype=np.float64)
    ns, nd = len(src), len(dst)
    points = min(ns, nd)
    found = [0, 0]
    errs = [0.0, 0.0]
    possible = 0
    corr = [np.asarray([]), np.asarray([])]
    if ns > 0 and nd > 0:
        dev = _device()
        s = torch.from_numpy(np.ascontiguousarray(src[:, :3])).to(dev)
        d = torch.from_numpy(np.ascontiguousarray(dst[:, :3])).to(dev)
        counts = torch.zeros(4, dtype=torch.int32, device=dev)
        errors = torch.zeros(2, dtype=torch.float64, device=dev)
        cs = torch.empty((points, 2), dtype=torch.int32, device=dev)
        cm = torch.empty((points, 2), dtype=torch.int32, device=dev)
        cap = int(min(ns * nd, MAX_EDGES))
        ws = ops._workspace("repeat", dev, lib().balf_repeatability_workspace_bytes(ns, nd, cap))
        with torch.cuda.device(dev):
            check(lib().balf_repeatability(s.data_ptr(), ns, d.data_ptr(), nd, float(overlap_err), float(eps),
                                           float(dist_match_thresh), float(radious_size), cap, counts.data_ptr(),
                                           errors.data_ptr(), cs.data_ptr(), cm.data_ptr(), ws.data_ptr(), ws.numel(),
                                           current_stream_ptr(dev)), "balf_repeatability")
        c = counts.cpu().numpy()
        e = errors.cpu().numpy()
        if c[0] < 0 or c[1] < 0:            # the candidate list of a scale did not fit max_edges (reported by the device)
            raise BalfHipError(f"balf_repeatability: more than {cap} candidate pairs (BALF_ERR_WORKSPACE)")
        found = [int(c[0]), int(c[1])]
        possible = int(c[2])
        errs = [float(e[0]), float(e[1])]
        corr = [cs[:found[0]].cpu().numpy().astype(np.int64) if found[0] else np.asarray([]),
                cm[:found[1]].cpu().numpy().astype(np.int64) if found[1] else np.asarray([])]
    rep_s = (found[0] / np.asarray(points, float)) * 100.0
    rep_m = (found[1] / np.asarray(points, float)) * 100.0
    err_s = 0.0 if found[0] == 0 else errs[0] / float(found[0] + np.finfo(float).eps)
    err_m = 0.0 if found[1] == 0 else errs[1] / float(found[1] + np.finfo(float).eps)
    return {'rep_single_scale': rep_s, 'rep_multi_scale': rep_m, 'num_points_single_scale': found[0],
            'num_points_multi_scale': found[1], 'error_overlap_single_scale': err_s,
            'error_overlap_multi_scale': err_m, 'total_num_points': points,
            'correspondences': corr[0], 'possible_matches': possible, 'correspondences_m': corr[1]}


def check_common_points(kpts, mask):
    """Indices of the key points (rows ``[y, x, ...]``) that fall inside ``mask`` (repeatability_tools.py:8-13; index
    bookkeeping on the host, like the reference: note its off-by-one ``mask[round(y) - 1, round(x) - 1]``)."""
    kpts = np.asarray(kpts)
    if len(kpts) == 0:
        return np.asarray([])
    r = np.rint(kpts[:, :2]).astype(np.int64) - 1
    return np.flatnonzero(np.asarray(mask)[r[:, 0], r[:, 1]] != 0)


def select_top_k(kpts, k=1000):
    """Indices of the ``k`` highest-scoring rows (score in column 3; repeatability_tools.py:15-17)."""
    return np.argsort(-1 * np.asarray(kpts)[:, 3])[:k]
