"""GPU counterpart of the point geometry in /root/reference/balf/benchmark_test/geometry_tools.py:43-86."""
from __future__ import annotations

import numpy as np
import torch

from .._lib import check, current_stream_ptr, lib
from .repeatability_tools import _device


def apply_homography_to_points(points, h):
    """rows (x, y, radius, score) -> the points warped by ``h`` with the radius rescaled by the warp's local
    affine approximation (geometry_tools.py:43-64), float64; ``balf_apply_homography`` in include/balf_hip.h."""
    pts = np.asarray(points, dtype=np.float64)
    if len(pts) == 0:
        return np.asarray([])
    dev = _device()
    p = torch.from_numpy(np.ascontiguousarray(pts[:, :4])).to(dev)
    hm = torch.from_numpy(np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(9))).to(dev)
    out = torch.empty_like(p)
    with torch.cuda.device(dev):
        check(lib().balf_apply_homography(p.data_ptr(), len(pts), hm.data_ptr(), out.data_ptr(), current_stream_ptr(dev)),
              "balf_apply_homography")
    return out.cpu().numpy()
