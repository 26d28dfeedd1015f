"""The inference / evaluation helpers of /root/reference/balf/utils/train_utils.py with every stage on the GPU.
Training (``train_model``, losses, optimiser; train_utils.py:20-160) is out of scope.

* ``extract_detections`` (train_utils.py:416-454) -- see :mod:`balf_amd.pipeline`.
* ``compute_repeatability_with_maximum_filter`` (train_utils.py:170-196): window-max NMS of both score maps, common-
  region masks (supplied by the caller: the reference builds them with ``cv2.warpPerspective``, which is not rebuilt
  here), top-K points, homography of the destination points, repeatability.
"""
from __future__ import annotations

import numpy as np

from ..benchmark_test import geometry_tools, repeatability_tools
from ..pipeline import extract_detections  # noqa: F401
from . import test_utils


def compute_repeatability_with_maximum_filter(src_scores_np, dst_scores_np, homography, mask_src, mask_dst, nms_size,
                                              num_points):
    src_scores_common_nms = np.multiply(test_utils.apply_nms(src_scores_np, nms_size), mask_src)
    dst_scores_common_nms = np.multiply(test_utils.apply_nms(dst_scores_np, nms_size), mask_dst)
    src_pts_nms = test_utils.get_point_coordinates(src_scores_common_nms, num_points=num_points, order_coord='xysr')
    dst_pts_nms = test_utils.get_point_coordinates(dst_scores_common_nms, num_points=num_points, order_coord='xysr')
    dst_to_src_pts_nms = geometry_tools.apply_homography_to_points(dst_pts_nms, homography)
    r = repeatability_tools.compute_repeatability(src_pts_nms, dst_to_src_pts_nms)
    return ([r['rep_single_scale']], [r['rep_multi_scale']], [r['error_overlap_single_scale']],
            [r['error_overlap_multi_scale']], [r['possible_matches']])

