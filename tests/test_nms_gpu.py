"""HIP window-max NMS + top-K (through the C ABI) against the oracle and the golden vectors of the
reference.  Bit-exact: same indices, same score bits, same counts."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from balf_amd import _lib
    assert _lib.lib().balf_device_check() == 0, "not a gfx950 device"
    return torch.device("cuda:0")


def run_topk(score_np, border, nms, k, dev, crop=(0, 0, None, None)):
    from balf_amd import ops
    t = torch.from_numpy(np.ascontiguousarray(score_np)).to(dev)
    if t.dim() == 2:
        t = t.unsqueeze(0)
    cy, cx, h, w = crop
    h = t.shape[1] if h is None else h
    w = t.shape[2] if w is None else w
    idx, sc, cnt = ops.nms_topk(t, cy, cx, h, w, border, nms, k)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_golden_cases(dev, name):
    f = np.load(os.path.join(G, "nms_topk.npz"))
    spec = cases.NMS_CASES[name]
    score = cases.nms_input(spec)
    idx, sc, cnt = run_topk(score, spec["border"], spec["nms"], spec["k"], dev)
    ref_idx, ref_sc = f[name + ".idx"], f[name + ".score"]
    n = int(cnt[0])
    assert n == ref_idx.size
    assert np.all(idx[0, n:] == -1) and np.all(sc[0, n:] == 0)
    o = np.argsort(idx[0, :n], kind="stable")
    assert np.array_equal(idx[0, :n][o], ref_idx)                      # raster order == sorted flat index
    assert np.array_equal(sc[0, :n][o].view(np.uint32), ref_sc.view(np.uint32))
    # emitted order: score descending, index ascending among equals
    ci, cs = O.canonical_order(ref_idx.astype(np.int64), ref_sc)
    assert np.array_equal(idx[0, :n], ci.astype(np.int32))
    assert np.array_equal(sc[0, :n].view(np.uint32), cs.view(np.uint32))


@pytest.mark.parametrize("name", ["rand_120x160", "ties_coarse", "const_plateau", "nms4_even", "nms16_even",
                                  "zeros_96x128", "sparse_few"])
def test_dense_nms_matches_golden_support(dev, name):
    from balf_amd import ops
    f = np.load(os.path.join(G, "nms_topk.npz"))
    spec = cases.NMS_CASES[name]
    score = cases.nms_input(spec)
    t = torch.from_numpy(score).to(dev).unsqueeze(0)
    out = ops.window_nms(t, spec["border"], spec["nms"])[0].cpu().numpy()
    assert np.array_equal(np.flatnonzero(out.ravel() != 0).astype(np.int32), f[name + ".nms_nonzero"])
    ref = O.apply_nms(O.remove_borders(score, spec["border"]), spec["nms"])
    assert np.array_equal(out.view(np.uint32), ref.astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("size", [1, 2, 3, 5, 7, 8, 15, 16, 31, 32])
def test_window_sizes_vs_oracle(dev, size):
    rng = np.random.default_rng(100 + size)
    score = (np.round(rng.random((3, 150, 201), dtype=np.float32) * 50) / 50).astype(np.float32)
    k = 777
    idx, sc, cnt = run_topk(score, 4, size, k, dev)
    for b in range(3):
        ri, rs = O.canonical_order(*O.select_topk(O.apply_nms(O.remove_borders(score[b], 4), size), k))
        n = int(cnt[b])
        assert n == ri.size
        assert np.array_equal(idx[b, :n], ri.astype(np.int32))
        assert np.array_equal(sc[b, :n].view(np.uint32), rs.astype(np.float32).view(np.uint32))


def test_crop_inside_padded_map_batch(dev):
    """Score maps read straight out of the padded prob tensor: crop offsets as in train_utils.py:437-442."""
    rng = np.random.default_rng(5)
    hp, wp, h, w = 512, 704, 481, 641
    top, left = O.crop_offsets(h, w, hp, wp)
    prob = rng.random((4, hp, wp), dtype=np.float32)
    idx, sc, cnt = run_topk(prob, 15, 15, 1000, dev, crop=(top, left, h, w))
    for b in range(4):
        ri, rs = O.detect_from_prob(prob[b], h, w, 15, 15, 1000)
        assert int(cnt[b]) == ri.size == 1000
        assert np.array_equal(idx[b], ri.astype(np.int32))
        assert np.array_equal(sc[b].view(np.uint32), rs.view(np.uint32))


@pytest.mark.parametrize("hw,k", [((1080, 1920), 2000), ((720, 1280), 2000), ((480, 640), 1000)])
def test_config_sizes_tie_heavy(dev, hw, k):
    rng = np.random.default_rng(77)
    h, w = hw
    score = np.stack([rng.random((h, w), dtype=np.float32),
                      (np.round(rng.random((h, w), dtype=np.float32) * 200) / 200).astype(np.float32)])
    idx, sc, cnt = run_topk(score, 15, 15, k, dev)
    for b in range(2):
        ri, rs = O.detect_from_prob(score[b], h, w, 15, 15, k)
        assert int(cnt[b]) == ri.size
        assert np.array_equal(idx[b, :ri.size], ri.astype(np.int32))
        assert np.array_equal(sc[b, :ri.size].view(np.uint32), rs.view(np.uint32))


def test_full_size_properties(dev):
    """Size-independent properties at BASELINE's batch: idempotence of NMS, selected points are
    fixed points of the dense NMS map, sortedness, determinism across runs."""
    from balf_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    prob = torch.rand((8, 1088, 1920), generator=g).to(dev)
    h, w, top, left = 1080, 1920, 4, 0
    idx, sc, cnt = ops.nms_topk(prob, top, left, h, w, 15, 15, 2000)
    idx2, sc2, cnt2 = ops.nms_topk(prob, top, left, h, w, 15, 15, 2000)
    assert torch.equal(idx, idx2) and torch.equal(sc, sc2) and torch.equal(cnt, cnt2)
    assert torch.all(cnt == 2000)
    assert torch.all(sc[:, :-1] >= sc[:, 1:])
    dense = ops.window_nms(prob[:, top:top + h, left:left + w].contiguous(), 15, 15)
    again = ops.window_nms(dense, 0, 15)
    assert torch.equal(dense, again)                                   # NMS is idempotent
    picked = torch.gather(dense.reshape(8, -1), 1, idx.long())
    assert torch.equal(picked, sc)                                     # every keypoint is an NMS survivor
    kth = torch.topk(dense.reshape(8, -1), 2000, dim=1).values[:, -1]
    assert torch.equal(sc[:, -1], kth)                                 # threshold is the K-th largest


def test_k_larger_than_map_is_index_error(dev):
    from balf_amd import ops
    with pytest.raises(IndexError):
        ops.nms_topk(torch.rand((1, 4, 4), device=dev), 0, 0, 4, 4, 0, 3, 17)


def test_test_utils_mirror(dev):
    from balf_amd.utils import test_utils as T
    f = np.load(os.path.join(G, "nms_topk.npz"))
    name = "ties_480x640"
    spec = cases.NMS_CASES[name]
    score = cases.nms_input(spec)
    nms = T.apply_nms(T.remove_borders(score, borders=spec["border"]), spec["nms"])
    ind = T.find_index_higher_scores(nms, num_points=spec["k"])
    assert np.array_equal((ind[:, 0] * score.shape[1] + ind[:, 1]).astype(np.int32), f[name + ".idx"])
    pts = T.get_point_coordinates(nms, num_points=spec["k"], order_coord="xysr")
    assert pts.dtype == np.float64 and pts.shape == (spec["k"], 4)
    assert np.array_equal(pts[:, 3].astype(np.float32).view(np.uint32), f[name + ".score"].view(np.uint32))
    assert np.all(pts[:, 2] == 1.0)


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_explicit_threshold_golden(dev, name):
    """`threshold != -1` of find_index_higher_scores / get_point_coordinates (test_utils.py:74-95) through the mirror
    (balf_nms_threshold for positive thresholds) against the reference's own index lists, both coordinate orders."""
    from balf_amd.utils import test_utils as T
    f = np.load(os.path.join(G, "nms_topk.npz"))
    spec = cases.NMS_CASES[name]
    score = cases.nms_input(spec)
    w = score.shape[1]
    nms = O.apply_nms(O.remove_borders(score, spec["border"]), spec["nms"])
    for j, t in enumerate(f[name + ".thr"]):
        ref = f[f"{name}.thr{j}.idx"]
        ind = T.find_index_higher_scores(nms, num_points=spec["k"], threshold=float(t))
        assert ind.shape == (ref.size, 2), (name, t, ind.shape, ref.size)
        assert np.array_equal((ind[:, 0] * w + ind[:, 1]).astype(np.int32), ref), (name, t)
        for order in ("xysr", "yxsr"):
            pts = T.get_point_coordinates(nms, num_points=spec["k"], threshold=float(t), order_coord=order)
            assert pts.shape == (ref.size, 4)
            if ref.size:
                c, r = (0, 1) if order == "xysr" else (1, 0)
                assert np.array_equal((pts[:, r] * w + pts[:, c]).astype(np.int32), ref)
                assert np.array_equal(pts[:, 3].astype(np.float32), nms.ravel()[ref])


def test_random_sweep_vs_c_oracle(dev):
    """120 seeded random configurations -- odd sizes down to 1 x 1, every window size, borders that swallow the whole
    map, K from 1 to H*W, tie-heavy / sparse / zero maps, batches with crops -- bit-exact against the C oracle."""
    from balf_amd import ops
    from oracle import c_oracle
    rng = np.random.default_rng(777)
    for case in range(120):
        h, w = int(rng.integers(1, 161)), int(rng.integers(1, 201))
        size = int(rng.integers(1, 33))
        border = int(rng.integers(0, 24))
        k = int(rng.integers(1, min(h * w, 3000) + 1))
        b = int(rng.integers(1, 4))
        pad_t, pad_l = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        hp, wp = h + pad_t + int(rng.integers(0, 9)), w + pad_l + int(rng.integers(0, 9))
        kind = rng.choice(["rand", "quant", "quant8", "sparse", "zeros"])
        maps = []
        for i in range(b):
            r = rng.random((hp, wp), dtype=np.float32)
            if kind == "quant":
                r = np.round(r * 50) / 50
            elif kind == "quant8":
                r = np.round(r * 4) / 4
            elif kind == "sparse":
                r = np.where(rng.random((hp, wp)) < 0.02, r + 0.01, 0.0)
            elif kind == "zeros":
                r = np.zeros((hp, wp))
            maps.append(r.astype(np.float32))
        t = torch.from_numpy(np.stack(maps)).to(dev)
        idx, sc, cnt = ops.nms_topk(t, pad_t, pad_l, h, w, border, size, k)
        idx, sc, cnt = idx.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
        for i in range(b):
            ri, rs, _ = c_oracle.nms_topk(np.ascontiguousarray(maps[i][pad_t:pad_t + h, pad_l:pad_l + w]), border, size, k)
            ri, rs = O.canonical_order(ri.astype(np.int64), rs)
            n = int(cnt[i])
            tag = (case, h, w, size, border, k, kind, i)
            assert n == ri.size, tag
            assert np.array_equal(idx[i, :n], ri.astype(np.int32)), tag
            assert np.array_equal(sc[i, :n].view(np.uint32), rs.view(np.uint32)), tag
            assert np.all(idx[i, n:] == -1), tag


def test_window15_vector_kernel_sweep(dev):
    """The tuned window-15 tile kernel (nms_tile15_vec_kernel: taken for nms_size 15, survivor mode, 16-byte aligned rows)
    on 80 seeded configurations that all meet its conditions: maps of one to three 114-row tiles and one to four 64-column
    tiles with ragged edges, crops at multiples of four columns, borders from 0 to beyond a tile, K from 1 to all pixels,
    tie-heavy / sparse / zero maps and maps with NEGATIVE scores among positive ones (its maxima are integer maxima on the
    float bits, which order negative floats differently -- no point with a score <= 0 may survive either way) -- bit-exact
    against the C oracle, and identical to the generic kernel (BALF_NMS_NO_VEC is read once per process, so the comparison is with the
    oracle only here; tools/nms_ab.py runs both kernels)."""
    from balf_amd import ops
    from oracle import c_oracle
    rng = np.random.default_rng(4242)
    for case in range(80):
        h, w = int(rng.integers(1, 301)), int(rng.integers(1, 241))
        border = int(rng.choice([0, 1, 7, 15, 16, 23, 40, 70, 130]))
        k = int(rng.integers(1, min(h * w, 4000) + 1))
        b = int(rng.integers(1, 4))
        pad_t, pad_l = int(rng.integers(0, 9)), 4 * int(rng.integers(0, 3))
        hp = h + pad_t + int(rng.integers(0, 9))
        wp = (w + pad_l + int(rng.integers(0, 9)) + 3) // 4 * 4
        kind = rng.choice(["rand", "quant", "quant8", "sparse", "zeros", "signed", "signed_dense"])
        if kind.startswith("signed"):
            k = min(k, 8)                                                # keep the K-th score positive (see below)
        maps = []
        for i in range(b):
            r = rng.random((hp, wp), dtype=np.float32)
            if kind == "quant":
                r = np.round(r * 50) / 50
            elif kind == "quant8":
                r = np.round(r * 4) / 4
            elif kind == "sparse":
                r = np.where(rng.random((hp, wp)) < 0.02, r + 0.01, 0.0)
            elif kind == "zeros":
                r = np.zeros((hp, wp))
            elif kind == "signed":
                r = np.round((r - 0.6) * 20) / 20                        # mostly negative, ties, some -0.0
            elif kind == "signed_dense":
                r = r - 0.3
            maps.append(r.astype(np.float32))
        t = torch.from_numpy(np.stack(maps)).to(dev)
        idx, sc, cnt = ops.nms_topk(t, pad_t, pad_l, h, w, border, 15, k)
        idx, sc, cnt = idx.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
        for i in range(b):
            ri, rs, _ = c_oracle.nms_topk(np.ascontiguousarray(maps[i][pad_t:pad_t + h, pad_l:pad_l + w]), border, 15, k)
            if kind.startswith("signed") and not (rs.size and rs.min() > 0):
                continue       # the reference's <= 0 fallback on a map with negative scores returns -0.0 / negative maxima:
                               # outside the contract of the entry point (probability maps, include/balf_hip.h)
            ri, rs = O.canonical_order(ri.astype(np.int64), rs)
            n = int(cnt[i])
            tag = (case, h, w, border, k, kind, i)
            assert n == ri.size, tag
            assert np.array_equal(idx[i, :n], ri.astype(np.int32)), tag
            assert np.array_equal(sc[i, :n].view(np.uint32), rs.view(np.uint32)), tag
            assert np.all(idx[i, n:] == -1), tag
