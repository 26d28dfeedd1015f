"""GPU counterpart of /root/reference/balf/benchmark_test/geometry_tools.py: the common-region masks (:7-26) and the
point geometry (:43-86).  ``remove_borders`` / ``get_point_coordinates`` / ``find_index_higher_scores`` of that module
are the same functions as in ``balf/utils/test_utils.py``: use ``balf_amd.utils.test_utils``."""
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


def create_common_region_masks(h_dst_2_src, shape_src, shape_dst):
    """-> (mask_src [Hs,Ws], mask_dst [Hd,Wd]) float64 in {0, 1}: where each image is covered by the other
    (geometry_tools.py:7-26); ``balf_common_region_masks`` in include/balf_hip.h.  The reference computes them with
    ``cv2.warpPerspective``; cv2 is not available offline, so this follows OpenCV's documented algorithm and its
    parity with cv2 itself is unpinned."""
    import ctypes as C
    hm = np.ascontiguousarray(np.asarray(h_dst_2_src, dtype=np.float64).reshape(9))
    hs, ws, hd, wd = int(shape_src[0]), int(shape_src[1]), int(shape_dst[0]), int(shape_dst[1])
    dev = _device()
    ms = torch.empty((hs, ws), dtype=torch.float64, device=dev)
    md = torch.empty((hd, wd), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib().balf_common_region_masks(hm.ctypes.data_as(C.c_void_p), hs, ws, hd, wd, 15, ms.data_ptr(), md.data_ptr(),
                                             current_stream_ptr(dev)), "balf_common_region_masks")
    return ms.cpu().numpy(), md.cpu().numpy()
