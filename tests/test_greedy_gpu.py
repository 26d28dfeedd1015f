"""HIP greedy NMS (demo post-processing) against the reference's golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", list(cases.GREEDY_CASES))
def test_greedy_golden(name):
    from balf_amd import ops
    f = np.load(os.path.join(G, "greedy_nms.npz"))
    spec = cases.GREEDY_CASES[name]
    score = cases.nms_input(spec)
    t = torch.from_numpy(score).cuda().unsqueeze(0)
    idx, sc, _, cnt, tot = ops.greedy_nms(t, 0, 0, spec["h"], spec["w"], spec["border"], spec["conf"], spec["nms"], 4096)
    n = int(cnt[0])
    assert n == int(tot[0]) == f[name + ".idx"].size
    assert np.array_equal(idx[0, :n].cpu().numpy(), f[name + ".idx"])            # same points, same order
    assert np.array_equal(sc[0, :n].cpu().numpy().view(np.uint32), f[name + ".score"].view(np.uint32))
    assert np.all(idx[0, n:].cpu().numpy() == -1)


def test_greedy_batch_crop_vs_oracle_1080p():
    from balf_amd import ops
    rng = np.random.default_rng(3)
    hp, wp, h, w, top, left = 1088, 1920, 1080, 1920, 4, 0
    prob = rng.random((2, hp, wp), dtype=np.float32)
    t = torch.from_numpy(prob).cuda()
    idx, sc, xy, cnt, tot = ops.greedy_nms(t, top, left, h, w, 15, 0.001, 15, 2048, subpixel_patch=4)
    for b in range(2):
        rb = O.remove_borders(prob[b, top:top + h, left:left + w], 15)
        ri, rs = O.greedy_nms(rb, 0.001, 15)
        n = int(cnt[b])
        assert int(tot[b]) == ri.size and n == min(2048, ri.size)
        assert np.array_equal(idx[b, :n].cpu().numpy(), ri[:n].astype(np.int32))
        ref_xy = O.soft_argmax_refine(rb, ri[:n], 4)
        assert np.abs(xy[b, :n].cpu().numpy() - ref_xy).max() < 1e-3


def test_mirror_function():
    from balf_amd.utils import test_utils as T
    f = np.load(os.path.join(G, "greedy_nms.npz"))
    name = "g_blobs_240x320"
    spec = cases.GREEDY_CASES[name]
    rb = T.remove_borders(cases.nms_input(spec), spec["border"])
    pts = T.get_points_direct_from_score_map(heatmap=rb, conf_thresh=spec["conf"], nms_size=spec["nms"], subpixel=False)
    assert pts.dtype == np.float64 and pts.shape == (f[name + ".idx"].size, 4)
    assert np.array_equal((pts[:, 1] * spec["w"] + pts[:, 0]).astype(np.int32), f[name + ".idx"])
    assert T.get_points_direct_from_score_map(np.zeros((64, 64), np.float32), conf_thresh=0.001).shape == (0, 4)


def test_greedy_random_sweep_vs_oracle():
    """40 seeded random maps (odd sizes, thresholds, suppression radii 1..16, ties from quantisation) through
    balf_greedy_nms against the oracle's sequential nms_fast restatement: same points, same order, same score bits."""
    import torch
    from balf_amd import ops
    rng = np.random.default_rng(4242)
    for case in range(40):
        h, w = int(rng.integers(8, 90)), int(rng.integers(8, 110))
        dist = int(rng.integers(1, 17))
        border = int(rng.integers(0, 6))
        conf = float(rng.choice([0.001, 0.015, 0.2, 0.6]))
        m = rng.random((h, w), dtype=np.float32)
        if case % 3 == 1:
            m = (np.round(m * 20) / 20).astype(np.float32)
        if case % 5 == 2:
            m = np.where(rng.random((h, w)) < 0.05, m, 0.0).astype(np.float32)
        rb = O.remove_borders(m, border)
        ri, rs = O.greedy_nms(rb, conf, dist)
        k = h * w
        idx, sc, xy, cnt, tot = ops.greedy_nms(torch.from_numpy(m).cuda().unsqueeze(0), 0, 0, h, w, border, conf, dist, k, 0)
        n = int(cnt[0])
        tag = (case, h, w, dist, border, conf)
        assert n == len(ri) == int(tot[0]), tag
        assert np.array_equal(idx[0, :n].cpu().numpy(), np.asarray(ri, np.int32)), tag
        assert np.array_equal(sc[0, :n].cpu().numpy().view(np.uint32), np.asarray(rs, np.float32).view(np.uint32)), tag


@pytest.mark.parametrize("name", list(cases.NMS_FAST_CASES))
def test_nms_fast_corner_list_golden(name):
    """The stand-alone ``nms_fast`` on a corner list (reference test_utils.py:130-168; goldens recorded from the
    reference's own function): same surviving corners, same order, same indices -- float coordinates, several corners per
    cell (the reference reports the WORST corner of a cell that wins with its best one), 0 / 1 / 2 corners."""
    from balf_amd.utils import test_utils as T
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "nms_fast.npz"))
    h, w, n, dist, seed = cases.NMS_FAST_CASES[name]
    out, inds = T.nms_fast(cases.nms_fast_input(h, w, n, seed), h, w, dist)
    assert out.shape == f[name + ".out"].shape
    assert np.array_equal(np.asarray(out, dtype=np.float64), f[name + ".out"])
    assert np.array_equal(np.asarray(inds, dtype=np.int64), f[name + ".inds"])


@pytest.mark.parametrize("name,patch", cases.SUBPIXEL_CASES)
def test_subpixel_against_reference_code(name, patch):
    """Sub-pixel refinement of the demo path against vectors recorded from the REFERENCE'S OWN code
    (get_points_direct_from_score_map(subpixel=True): threshold, nms_fast, patch extraction, norm_patches, do_log,
    coordinate update -- test_utils.py:97-215) with its one third-party call, torchgeometry's SpatialSoftArgmax2d (not
    installable offline), supplied by a restatement of the published definition (tests/golden/make_golden.py).  Pins the
    reference's side of the path: same points, same order, same scores, coordinates within 2e-5 px (the reference runs that
    part in fp32 torch, with torchgeometry's eps = 1e-6 in the soft-max normaliser)."""
    from balf_amd.utils import test_utils as T
    ref = np.load(os.path.join(G, "subpixel.npz"))[f"{name}.p{patch}"]
    spec = cases.GREEDY_CASES[name]
    rb = O.remove_borders(cases.nms_input(spec), spec["border"])
    pts = T.get_points_direct_from_score_map(heatmap=rb, conf_thresh=spec["conf"], nms_size=spec["nms"], subpixel=True,
                                             patch_size=patch, order_coord="xysr")
    assert pts.shape == ref.shape
    assert np.array_equal(pts[:, 3].astype(np.float32).view(np.uint32), ref[:, 3].astype(np.float32).view(np.uint32))
    assert np.array_equal(pts[:, 2], ref[:, 2])
    err = float(np.abs(pts[:, :2] - ref[:, :2]).max())
    assert err < 2e-5, err


def _greedy_vs_oracle(m, conf, dist, border=0):
    from balf_amd import ops
    h, w = m.shape
    ri, rs = O.greedy_nms(O.remove_borders(m, border), conf, dist)
    k = min(h * w, 16384)
    idx, sc, _, cnt, tot = ops.greedy_nms(torch.from_numpy(m).cuda().unsqueeze(0), 0, 0, h, w, border, conf, dist, k, 0)
    n = int(cnt[0])
    assert int(tot[0]) == len(ri) and n == min(k, len(ri))
    assert np.array_equal(idx[0, :n].cpu().numpy(), np.asarray(ri[:n], np.int32))
    assert np.array_equal(sc[0, :n].cpu().numpy().view(np.uint32), np.asarray(rs[:n], np.float32).view(np.uint32))


@pytest.mark.parametrize("rounds", ["1", "2", "3"])
def test_greedy_tail_kernel_finishes_the_rounds(rounds, monkeypatch):
    """Round 6: the rounds are enqueued without looking at the data; what is alive after them is finished by the per-image
    tail kernel.  With BALF_GREEDY_ROUNDS = 1..3 the tail does (nearly) all the work: goldens and random maps bit-exact."""
    from balf_amd import ops
    monkeypatch.setenv("BALF_GREEDY_ROUNDS", rounds)
    f = np.load(os.path.join(G, "greedy_nms.npz"))
    for name, spec in cases.GREEDY_CASES.items():
        t = torch.from_numpy(cases.nms_input(spec)).cuda().unsqueeze(0)
        idx, sc, _, cnt, tot = ops.greedy_nms(t, 0, 0, spec["h"], spec["w"], spec["border"], spec["conf"], spec["nms"], 4096)
        n = int(cnt[0])
        assert n == int(tot[0]) == f[name + ".idx"].size, name
        assert np.array_equal(idx[0, :n].cpu().numpy(), f[name + ".idx"]), name
    rng = np.random.default_rng(77)
    for case in range(12):
        h, w = int(rng.integers(40, 200)), int(rng.integers(40, 300))
        m = rng.random((h, w), dtype=np.float32)
        if case % 2:
            m = (np.round(m * 10) / 10).astype(np.float32)
        _greedy_vs_oracle(m, 0.05, int(rng.integers(1, 17)), int(rng.integers(0, 4)))


def test_greedy_monotone_ramp_and_plateau():
    """The adversarial inputs of the parallel form: a monotone ramp (one kept point per window and round along the slope:
    ~W/d rounds, far beyond the enqueued ones, all tiles in window mode) and a constant plateau (every comparison a tie,
    raster-first wins)."""
    h, w = 96, 700
    yy, xx = np.mgrid[0:h, 0:w]
    for ramp in (xx + 0.001 * yy, -xx - 0.001 * yy, yy * w + xx, -(yy * w + xx)):
        m = (0.1 + 0.8 * (ramp - ramp.min()) / (ramp.max() - ramp.min())).astype(np.float32)
        _greedy_vs_oracle(m, 0.05, 5)
    _greedy_vs_oracle(np.full((h, w), 0.5, np.float32), 0.05, 7)
    _greedy_vs_oracle(np.full((70, 130), 0.5, np.float32), 0.05, 16, border=3)


def test_greedy_is_stream_ordered():
    """balf_greedy_nms enqueues and returns: behind a kernel that keeps the stream busy for a while the call comes back
    long before that kernel ends (round 5 synchronised the stream once per four rounds)."""
    import time
    from balf_amd import ops
    rng = np.random.default_rng(5)
    t = torch.from_numpy(rng.random((4, 256, 320), dtype=np.float32)).cuda()
    ops.greedy_nms(t, 0, 0, 256, 320, 4, 0.015, 15, 512, 5)               # warm: workspace, attributes
    torch.cuda.synchronize()
    done = torch.cuda.Event()
    torch.cuda._sleep(int(1.5e9))                                          # ~0.7 s of device time
    t0 = time.perf_counter()
    idx, sc, xy, cnt, tot = ops.greedy_nms(t, 0, 0, 256, 320, 4, 0.015, 15, 512, 5)
    dt = time.perf_counter() - t0
    done.record()
    busy = not done.query()
    torch.cuda.synchronize()
    assert busy, "the stream had drained when balf_greedy_nms returned: it waited for the device"
    assert dt < 0.1, f"balf_greedy_nms took {dt * 1e3:.1f} ms of host time behind a busy stream"
    ri, _ = O.greedy_nms(O.remove_borders(t[1].cpu().numpy(), 4), 0.015, 15)
    assert np.array_equal(idx[1, :int(cnt[1])].cpu().numpy(), np.asarray(ri[:512], np.int32))


def test_greedy_batched_crops_odd_shapes_vs_oracle():
    """Round 6 (new bit-map / tile-list kernels): batches of DIFFERENT images cut out of a padded map at an offset, widths that
    are not multiples of the 64-pixel word, heights below one 32-row tile, radii 0..16, borders, quantised (tie-heavy) and
    sparse maps -- every image against the oracle's sequential sweep: same points, same order, same score bits, and `total`."""
    from balf_amd import ops
    rng = np.random.default_rng(20266)
    shapes = [(5, 9), (31, 64), (33, 65), (64, 63), (97, 257), (130, 300), (200, 129)]
    for case, (h, w) in enumerate(shapes * 3):
        b = int(rng.integers(1, 4))
        top, left = int(rng.integers(0, 5)), int(rng.integers(0, 7))
        hp, wp = h + top + int(rng.integers(0, 4)), w + left + int(rng.integers(0, 4))
        dist = int(rng.integers(0, 17))
        border = int(rng.integers(0, min(h, w) // 2 + 1)) if case % 4 == 0 else int(rng.integers(0, 3))
        conf = float(rng.choice([0.015, 0.3, 0.9]))
        m = rng.random((b, hp, wp), dtype=np.float32)
        if case % 3 == 1:
            m = (np.round(m * 8) / 8).astype(np.float32)
        if case % 3 == 2:
            m = np.where(rng.random((b, hp, wp)) < 0.03, m, 0.0).astype(np.float32)
        k = min(h * w, 16384)
        idx, sc, _, cnt, tot = ops.greedy_nms(torch.from_numpy(m).cuda(), top, left, h, w, border, conf, dist, k, 0)
        for i in range(b):
            rb = O.remove_borders(m[i, top:top + h, left:left + w], border)
            ri, rs = O.greedy_nms(rb, conf, dist)
            n = int(cnt[i])
            tag = (case, i, h, w, hp, wp, top, left, dist, border, conf)
            assert int(tot[i]) == len(ri) and n == min(k, len(ri)), tag
            assert np.array_equal(idx[i, :n].cpu().numpy(), np.asarray(ri[:n], np.int32)), tag
            assert np.array_equal(sc[i, :n].cpu().numpy().view(np.uint32), np.asarray(rs[:n], np.float32).view(np.uint32)), tag
            assert np.all(idx[i, n:].cpu().numpy() == -1), tag


def test_greedy_is_deterministic_under_repetition():
    """The survivor list is appended in whatever order the workgroups finish; the top-K kernel sorts it (score desc, index asc), so
    the outputs must be bit-identical call after call (200 calls on a tie-heavy 4 x 256 x 320 batch)."""
    from balf_amd import ops
    rng = np.random.default_rng(9)
    m = (np.round(rng.random((4, 256, 320), dtype=np.float32) * 50) / 50).astype(np.float32)
    t = torch.from_numpy(m).cuda()
    ref = ops.greedy_nms(t, 0, 0, 256, 320, 3, 0.1, 9, 2048, 5)
    ref = [r.clone() for r in ref]
    for _ in range(200):
        out = ops.greedy_nms(t, 0, 0, 256, 320, 3, 0.1, 9, 2048, 5)
        assert all(torch.equal(a, b_) for a, b_ in zip(out, ref))
