"""Repeatability evaluation: balf_repeatability / balf_apply_homography through the host mirror
against the goldens recorded from the reference's own function bodies, and against the oracle on larger inputs."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from balf_amd.benchmark_test import geometry_tools, repeatability_tools       # noqa: E402
from oracle import oracle                                                      # noqa: E402
from tests.golden import cases                                                 # noqa: E402

pytestmark = pytest.mark.gpu
INT_KEYS = ("num_points_single_scale", "num_points_multi_scale", "total_num_points", "possible_matches",
            "correspondences", "correspondences_m")
FLOAT_KEYS = ("rep_single_scale", "rep_multi_scale", "error_overlap_single_scale", "error_overlap_multi_scale")


def _same(res, ref):
    for k in INT_KEYS:
        assert np.array_equal(np.asarray(res[k]).reshape(-1), np.asarray(ref[k]).reshape(-1)), k
    for k in FLOAT_KEYS:
        assert abs(float(res[k]) - float(ref[k])) < 1e-12, k


@pytest.mark.parametrize("name", list(cases.REPEAT_CASES))
def test_matches_reference_goldens(name):
    g = np.load(os.path.join(HERE, "golden", "repeatability.npz"))
    spec = cases.REPEAT_CASES[name]
    src, dst = cases.repeat_inputs(spec)
    res = repeatability_tools.compute_repeatability(src, dst, **spec["kw"])
    _same(res, {k: g[f"{name}.{k}"] for k in INT_KEYS + FLOAT_KEYS})


@pytest.mark.parametrize("ns,nd,planted", [(1000, 1000, 700), (2000, 1500, 1200), (1, 1, 1), (3, 700, 2)])
def test_vs_oracle_at_benchmark_sizes(ns, nd, planted):
    """1000 x 1000 is what the reference's HPatches evaluation feeds in (a 10^6-iteration Python loop there)."""
    spec = dict(ns=ns, nd=nd, seed=77, planted=planted, kw={})
    src, dst = cases.repeat_inputs(spec)
    _same(repeatability_tools.compute_repeatability(src, dst), oracle.compute_repeatability(src, dst))


def test_exact_ties_resolve_by_flat_index():
    """Integer grids give many exactly equal distances; the kernel's documented tie-break (lower flat index first) is
    the oracle's."""
    ys, xs = np.mgrid[0:12, 0:12]
    src = np.stack([xs.ravel() * 20.0, ys.ravel() * 20.0, np.ones(144), np.ones(144)], axis=1)
    dst = src.copy()
    dst[:, 0] += 6.0
    _same(repeatability_tools.compute_repeatability(src, dst), oracle.compute_repeatability(src, dst))


def test_empty_inputs():
    r = repeatability_tools.compute_repeatability(np.zeros((0, 4)), np.zeros((5, 4)))
    assert r["num_points_single_scale"] == 0 and r["total_num_points"] == 0 and r["possible_matches"] == 0


def test_apply_homography_matches_reference_golden():
    g = np.load(os.path.join(HERE, "golden", "repeatability.npz"))
    src, _ = cases.repeat_inputs(cases.REPEAT_CASES["small"])
    out = geometry_tools.apply_homography_to_points(src, cases.HOMOGRAPHY)
    assert np.abs(out - g["homography.points"]).max() < 1e-11


def test_evaluation_glue_matches_reference_golden():
    """train_utils.compute_repeatability_with_maximum_filter end to end on the GPU (window-max NMS, masks, top-K
    points, homography, repeatability) against the result recorded from the reference's own function bodies."""
    from balf_amd.utils import train_utils
    g = np.load(os.path.join(HERE, "golden", "repeatability.npz"))
    es, ed, ms, md = cases.eval_inputs(cases.EVAL_CASE)
    res = train_utils.compute_repeatability_with_maximum_filter(es, ed, cases.HOMOGRAPHY, ms, md, cases.EVAL_CASE["nms"],
                                                                cases.EVAL_CASE["num_points"])
    assert len(res) == 5 and all(isinstance(v, list) and len(v) == 1 for v in res)
    assert np.allclose([float(np.asarray(v[0])) for v in res], g["eval.result"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape_src,shape_dst,hm", [
    ((240, 320), (240, 320), cases.HOMOGRAPHY),
    ((480, 640), (400, 600), [[0.93, -0.11, 31.0], [0.08, 1.04, -12.5], [1.2e-4, -6.0e-5, 1.0]]),
    ((200, 260), (300, 280), [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]),
    ((120, 160), (120, 160), [[0.5, 0.0, 200.0], [0.0, 0.5, 200.0], [0.0, 0.0, 1.0]]),       # no overlap at all
    ((480, 640), (480, 640), [[1.0, 0.0, 7.03125], [0.0, 1.0, -3.515625], [0.0, 0.0, 1.0]]),  # every coordinate ON a 1/32-px tie
    ((1080, 1920), (1080, 1920), [[0.98, 0.03, 11.0], [-0.02, 1.01, 5.0], [2.0e-5, -1.0e-5, 1.0]]),
])
def test_common_region_masks_vs_oracle(shape_src, shape_dst, hm):
    """create_common_region_masks (geometry_tools.py:7-26).  The reference computes the masks with cv2.warpPerspective,
    which is not installed here: the HIP kernel and the oracle both restate OpenCV's algorithm (parity with cv2 itself
    UNPINNED) in the same individually rounded fp64 operations (closed-form inverse, no FMA contraction), so the two {0,1}
    masks must be EQUAL, 1/32-pixel ties included."""
    hm = np.asarray(hm, dtype=np.float64)
    ms, md = geometry_tools.create_common_region_masks(hm, shape_src, shape_dst)
    rs, rd = oracle.create_common_region_masks(hm, shape_src, shape_dst, numpy_inverse=False)
    assert ms.shape == tuple(shape_src) and md.shape == tuple(shape_dst) and ms.dtype == np.float64
    assert set(np.unique(ms)) <= {0.0, 1.0} and set(np.unique(md)) <= {0.0, 1.0}
    assert ms[:15].sum() == 0 and ms[:, :15].sum() == 0 and md[-15:].sum() == 0 and md[:, -15:].sum() == 0
    assert np.array_equal(ms, rs) and np.array_equal(md, rd)
    # and against the reference's own inverse (np.linalg.inv, geometry_tools.py:9): an independent statement of the
    # reference's line, where the last bits of the inverse may flip a 1/32-pixel rounding tie
    fs, fd = oracle.create_common_region_masks(hm, shape_src, shape_dst, numpy_inverse=True)
    assert int((ms != fs).sum()) <= 4 and int((md != fd).sum()) <= 4, (int((ms != fs).sum()), int((md != fd).sum()))


def test_common_region_masks_identity_is_the_inner_frame():
    ms, md = geometry_tools.create_common_region_masks(np.eye(3), (100, 140), (100, 140))
    ref = np.zeros((100, 140)); ref[15:85, 15:125] = 1.0
    assert np.array_equal(ms, ref) and np.array_equal(md, ref)


def test_repeatability_overflow_is_reported_without_a_sync(monkeypatch):
    """Round 6: balf_repeatability no longer reads the candidate count back (no hipStreamSynchronize); a list longer than
    max_edges is cut on the device and reported as count -1, which the host mirror turns into an error."""
    import time
    import torch
    from balf_amd._lib import BalfHipError
    from balf_amd.benchmark_test import repeatability_tools as R
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.uniform(50, 60, (300, 2)), np.full((300, 1), 20.0)], axis=1)      # 300 x 300 overlapping discs
    ref = R.compute_repeatability(pts, pts + 0.25)
    assert ref["num_points_single_scale"] > 0
    monkeypatch.setattr(R, "MAX_EDGES", 1000)
    with pytest.raises(BalfHipError, match="candidate pairs"):
        R.compute_repeatability(pts, pts + 0.25)
    monkeypatch.setattr(R, "MAX_EDGES", 1 << 22)
    # stream order: behind a busy stream the library call returns at once (the mirror's .cpu() is what waits)
    from balf_amd import ops
    from balf_amd._lib import check, current_stream_ptr, lib
    dev = torch.device("cuda:0")
    s = torch.from_numpy(pts).to(dev)
    d = torch.from_numpy(pts + 0.25).to(dev)
    counts = torch.zeros(4, dtype=torch.int32, device=dev)
    errors = torch.zeros(2, dtype=torch.float64, device=dev)
    cs = torch.empty((300, 2), dtype=torch.int32, device=dev)
    cm = torch.empty((300, 2), dtype=torch.int32, device=dev)
    cap = 300 * 300
    ws = ops._workspace("repeat", dev, lib().balf_repeatability_workspace_bytes(300, 300, cap))
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    torch.cuda._sleep(int(1.5e9))
    t0 = time.perf_counter()
    check(lib().balf_repeatability(s.data_ptr(), 300, d.data_ptr(), 300, 0.4, 1e-6, 3.0, 30.0, cap, counts.data_ptr(),
                                   errors.data_ptr(), cs.data_ptr(), cm.data_ptr(), ws.data_ptr(), ws.numel(),
                                   current_stream_ptr(dev)), "balf_repeatability")
    dt = time.perf_counter() - t0
    done.record()
    busy = not done.query()
    torch.cuda.synchronize()
    assert busy and dt < 0.1, (busy, dt)
    assert int(counts[0]) == ref["num_points_single_scale"] and int(counts[1]) == ref["num_points_multi_scale"]
