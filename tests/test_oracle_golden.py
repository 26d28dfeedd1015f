"""The oracle (oracle/oracle.py) against the golden vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from balf_amd import arch
from balf_amd.utils import synth
from oracle import oracle as O
from tests.golden import cases

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def sd():
    return synth.synthetic_state_dict(cases.WEIGHT_SEED)


@pytest.fixture(scope="module")
def fsmall():
    return np.load(os.path.join(G, "forward_small.npz"))


@pytest.mark.parametrize("name", list(cases.FORWARD_SMALL))
def test_forward_small_fp32(sd, fsmall, name):
    b, h, w, seed = cases.FORWARD_SMALL[name]
    taps = {}
    with torch.no_grad():
        out = O.detector_forward(sd, cases.forward_input(b, h, w, seed), taps)
    # same torch primitives, different (mathematically equal) token-mix formulation: fp32 rounding only
    assert np.abs(out["logits"].numpy() - fsmall[name + ".logits"]).max() < 2e-4
    assert np.abs(out["prob"].numpy() - fsmall[name + ".prob"]).max() < 2e-6
    assert out["logits"].shape == (b, 65, h // 8, w // 8) and out["prob"].shape == (b, h, w)


def test_forward_fp64_is_closer_than_tolerance(sd, fsmall):
    name = "b1_128x192"
    b, h, w, seed = cases.FORWARD_SMALL[name]
    with torch.no_grad():
        out = O.detector_forward(O.cast_state(sd, torch.float64), cases.forward_input(b, h, w, seed).double())
    assert np.abs(out["prob"].numpy() - fsmall[name + ".prob"]).max() < 2e-6
    assert np.abs(out["logits"].numpy() - fsmall[name + ".logits"]).max() < 2e-4


def test_stage_outputs(sd, fsmall):
    name = cases.TAP_CASE
    b, h, w, seed = cases.FORWARD_SMALL[name]
    x = cases.forward_input(b, h, w, seed).permute(0, 2, 3, 1)
    with torch.no_grad():
        for i in range(4):
            x = O.stage_forward(sd, f"down{i + 1}", x, last=(i == 3))
            ref = fsmall[f"{name}.down{i + 1}"]                     # NCHW from the reference
            got = x.permute(0, 3, 1, 2).numpy()
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", cases.TAP_SAMPLED)
def test_stage_outputs_sampled(sd, name):
    """The oracle's stages against the reference's own down1..down4 outputs at 128x192 and 256x320 (stage_taps.npz: strided
    samples + per-channel sums; recorded through forward hooks on the reference's modules)."""
    f = np.load(os.path.join(G, "stage_taps.npz"))
    b, h, w, seed = cases.FORWARD_SMALL[name]
    x = cases.forward_input(b, h, w, seed).permute(0, 2, 3, 1)
    with torch.no_grad():
        for i in range(4):
            x = O.stage_forward(sd, f"down{i + 1}", x, last=(i == 3))
            got, gsum = cases.stage_sample(x.permute(0, 3, 1, 2).numpy())
            ref, rsum = f[f"{name}.down{i + 1}.sample"], f[f"{name}.down{i + 1}.chansum"]
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
            assert np.abs(gsum - rsum).max() < 1e-5 * max(1.0, np.abs(rsum).max()) * x.shape[1] * x.shape[2] ** 0.5


def test_forward_cfg_vga(sd):
    f = np.load(os.path.join(G, "forward_cfg.npz"))
    h, w, k, img_index = cases.FORWARD_CFG["vga"]
    img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(h, w, img_index))
    pts, prob = O.extract_detections(sd, img, nms_size=15, num_points=k, border_size=15)
    assert np.abs(prob[::8, ::8] - f["vga.prob_s8"]).max() < 2e-6
    assert np.abs(prob[cases.CFG_ROWS(prob.shape[0])] - f["vga.prob_rows"]).max() < 2e-6
    got = np.sort((pts[:, 1] * w + pts[:, 0]).astype(np.int64))
    ref = f["vga.idx"].astype(np.int64)
    # the oracle's prob differs from the reference's by fp32 rounding, so near-ties may flip
    # (SURVEY.md 7.2); identical-input index parity is test_nms_topk_cases below
    overlap = np.intersect1d(got, ref).size / ref.size
    assert overlap >= 0.99, overlap
    assert pts.shape == (k, 4) and np.all(np.diff(pts[:, 3]) <= 0) and np.all(pts[:, 2] == 1.0)


@pytest.mark.parametrize("name", ["720p", "1080p"])
def test_forward_cfg_headline_sizes(sd, name):
    """The oracle against the REFERENCE at the headline geometry (1088x1920: fh = 136, fw = 240 in the stage-1 grid
    branch), fixtures recorded from inside the reference's own extract_detections (make_golden.py)."""
    f = np.load(os.path.join(G, "forward_cfg.npz"))
    h, w, k, img_index = cases.FORWARD_CFG[name]
    img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(h, w, img_index))
    pts, prob = O.extract_detections(sd, img, nms_size=15, num_points=k, border_size=15)
    assert np.abs(prob[::8, ::8] - f[name + ".prob_s8"]).max() < 2e-6
    assert np.abs(prob[cases.CFG_ROWS(prob.shape[0])] - f[name + ".prob_rows"]).max() < 2e-6
    assert np.abs(cases.cfg_mix(prob) - f[name + ".prob_mix"]).max() < 2e-6
    assert np.abs(cases.cfg_cellsum(prob) - f[name + ".prob_cellsum"]).max() < 2e-5
    got = np.sort((pts[:, 1] * w + pts[:, 0]).astype(np.int64))
    overlap = np.intersect1d(got, f[name + ".idx"].astype(np.int64)).size / k
    assert overlap >= 0.99, overlap
    ref = f[name + ".pts"]
    assert ref.shape == pts.shape == (k, 4) and np.all(np.diff(ref[:, 3]) <= 0)


@pytest.mark.parametrize("name", list(cases.EXTRACT_CASES))
def test_extract_detections_identical_input(name):
    """train_utils.extract_detections run from the reference's source: on the very score map the reference's model
    produced, the oracle's crop / border / NMS / top-K gives the reference's points exactly (same set, same score bits)."""
    c = np.load(os.path.join(G, "callers.npz"))
    h, w, k, _, border, nms = cases.EXTRACT_CASES[name]
    idx, sc = O.detect_from_prob(c[name + ".prob"], h, w, border, nms, k)
    ref = c[name + ".pts"]
    ri = (ref[:, 1] * w + ref[:, 0]).astype(np.int64)
    o = np.lexsort((ri, -ref[:, 3]))
    assert np.array_equal(idx, ri[o]) and np.array_equal(sc.astype(np.float64), ref[o, 3])
    assert np.all(ref[:, 2] == 1.0) and np.all(np.diff(ref[:, 3]) <= 0)


def test_reference_extract_detections_raises_as_published():
    """Recorded fact: the published function indexes the 3-D `prob` with four subscripts (train_utils.py:444)."""
    geo = json.load(open(os.path.join(G, "geometry.json")))
    assert geo["extract_detections_unmodified"].startswith("IndexError")


@pytest.mark.parametrize("name", list(cases.DETECT_CASES))
def test_demo_detect_cases(sd, name):
    """demo_match.detect run from the reference's source against the oracle's restatement, end to end from the uint8
    image.  The oracle's score map differs from the reference's by fp32 rounding (1e-7), so integer pixel positions are
    compared as sets with a near-tie allowance and sub-pixel coordinates within 1e-3 px."""
    c = np.load(os.path.join(G, "callers.npz"))
    h, w, img_index, over = cases.DETECT_CASES[name]
    a = dict(cases.DETECT_ARGS, **over)
    res = O.demo_detect(sd, cases.detect_input(h, w, img_index), a["border_size"], a["nms_size"], a["num_features"],
                        a["heatmap_confidence_threshold"], a["sub_pixel"], a["patch_size"], a["order_coord"])
    if name + ".empty_pair_shapes" in c.files:
        assert isinstance(res, tuple) and [list(r.shape) for r in res] == c[name + ".empty_pair_shapes"].tolist()
        return
    ref = c[name + ".pts"]
    assert res.shape == ref.shape and np.all(res[:, 2] == 1.0)
    if a["sub_pixel"]:
        assert np.abs(res - ref).max() < 1e-3
    else:
        same = (res == ref).all(axis=1).mean()
        assert same >= 0.98, same
        assert set(map(tuple, res)) == set(map(tuple, ref)) or same >= 0.98


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_nms_topk_cases(name):
    f = np.load(os.path.join(G, "nms_topk.npz"))
    spec = cases.NMS_CASES[name]
    score = cases.nms_input(spec)
    nms = O.apply_nms(O.remove_borders(score, spec["border"]), spec["nms"])
    assert np.array_equal(np.flatnonzero(nms.ravel() != 0).astype(np.int32), f[name + ".nms_nonzero"])
    idx, sc = O.select_topk(nms, spec["k"])
    assert np.array_equal(idx.astype(np.int32), f[name + ".idx"])
    assert np.array_equal(sc.astype(np.float32).view(np.uint32), f[name + ".score"].view(np.uint32))
    ci, cs = O.canonical_order(idx, sc)
    assert np.array_equal(np.sort(ci), idx) and np.all(np.diff(cs.astype(np.float64)) <= 0)


@pytest.mark.parametrize("name", list(cases.NMS_CASES))
def test_explicit_threshold_cases(name):
    """`threshold != -1` of find_index_higher_scores (test_utils.py:91-95) against the reference's own outputs."""
    f = np.load(os.path.join(G, "nms_topk.npz"))
    spec = cases.NMS_CASES[name]
    nms = O.apply_nms(O.remove_borders(cases.nms_input(spec), spec["border"]), spec["nms"])
    for j, t in enumerate(f[name + ".thr"]):
        assert np.array_equal(O.select_threshold(nms, spec["k"], float(t)).astype(np.int32), f[f"{name}.thr{j}.idx"]), (name, t)


def test_topk_more_points_than_pixels_is_index_error():
    with pytest.raises(IndexError):
        O.select_topk(np.ones((4, 4), np.float32), 17)


def test_geometry_and_state_table():
    geo = json.load(open(os.path.join(G, "geometry.json")))
    for key, g in geo["pad"].items():
        h, w = map(int, key.split("x"))
        img = np.zeros((h, w, 3)); img[0, 0, 0] = 1.0
        ev = O.make_shape_even(img)
        pd = O.mod_padding_symmetric(ev, 64)
        assert list(ev.shape[:2]) == g["even"] and list(pd.shape[:2]) == g["padded"]
        assert list(np.argwhere(pd[:, :, 0] == 1.0)[0]) == g["origin"]
        assert list(O.crop_offsets(h, w, *pd.shape[:2])) == [g["h_start"], g["w_start"]]
        assert arch.padded_hw(h, w) == (g["padded"][0], g["padded"][1], g["h_start"], g["w_start"])
    ents = arch.state_entries()
    assert [[n, list(s), d] for n, s, d in ents] == geo["state"]
    assert len(ents) == 167


@pytest.mark.parametrize("name", list(cases.GREEDY_CASES))
def test_greedy_nms_cases(name):
    """The demo's get_points_direct_from_score_map(subpixel=False), captured from the reference."""
    f = np.load(os.path.join(G, "greedy_nms.npz"))
    spec = cases.GREEDY_CASES[name]
    rb = O.remove_borders(cases.nms_input(spec), spec["border"])
    idx, sc = O.greedy_nms(rb, spec["conf"], spec["nms"])
    assert np.array_equal(idx.astype(np.int32), f[name + ".idx"])
    assert np.array_equal(sc.astype(np.float32).view(np.uint32), f[name + ".score"].view(np.uint32))


@pytest.mark.parametrize("name,patch", cases.SUBPIXEL_CASES)
def test_subpixel_refinement_cases(name, patch):
    """O.soft_argmax_refine against the reference's own sub-pixel path (subpixel.npz: the reference's code around a
    restated torchgeometry call, see make_golden.py): within 2e-5 px (fp32 torch + eps 1e-6 there, fp64 here)."""
    ref = np.load(os.path.join(G, "subpixel.npz"))[f"{name}.p{patch}"]
    spec = cases.GREEDY_CASES[name]
    rb = O.remove_borders(cases.nms_input(spec), spec["border"])
    idx, sc = O.greedy_nms(rb, spec["conf"], spec["nms"])
    xy = O.soft_argmax_refine(rb, idx, patch)
    assert xy.shape[0] == ref.shape[0]
    assert np.array_equal(sc.astype(np.float32).view(np.uint32), ref[:, 3].astype(np.float32).view(np.uint32))
    assert float(np.abs(xy - ref[:, :2]).max()) < 2e-5


# ---------------- HardNet descriptor ----------------
def test_hardnet_oracle_matches_reference_goldens():
    """O.hardnet_forward against descriptors (and per-layer activations) recorded from the reference's own
    HardNet class (third_party/hardnet/hardnet_pytorch.py) with the seeded synthetic weights."""
    g = np.load(os.path.join(G, "hardnet.npz"))
    sd = synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED)
    assert list(g["state_keys"]) == list(sd.keys())
    for name, (n, seed) in cases.HARDNET_CASES.items():
        taps = {}
        d = O.hardnet_forward(sd, synth.synthetic_patches(n, seed), taps).numpy()
        assert np.abs(d - g[name + ".desc"]).max() < 1e-6
        if name == cases.HARDNET_TAP_CASE:
            for j in range(7):
                assert np.abs(taps[j][0, ::4].numpy() - g[f"{name}.act{j}"]).max() < 2e-5


def test_match_smnn_oracle_hand_case():
    """Mutual ratio test on a case small enough to check by hand."""
    e = np.eye(128, dtype=np.float32)
    d1 = torch.from_numpy(np.stack([e[0], e[1], e[2]]))
    d2 = torch.from_numpy(np.stack([e[1] * 0.9 + e[5] * 0.1, e[0], e[7], e[8]]))
    dist, idx = O.match_smnn(d1, d2, 0.9)
    # d1[0] <-> d2[1] (distance 0, ratio 0); d1[1] <-> d2[0]: forward 0.1414/1.4142 = 0.1, backward
    # 0.1414/sqrt(1.82) = 0.10483 -> max; d1[2] has no clear neighbour
    assert idx.tolist() == [[0, 1], [1, 0]]
    assert abs(float(dist[0])) < 1e-6 and abs(float(dist[1]) - (0.02 / 1.82) ** 0.5) < 1e-6
    assert O.match_smnn(d1, d2[:1], 0.9)[1].shape == (0, 2)          # fewer than two candidates: no matches


def test_extract_patches_oracle_identity_scale():
    """With scale = PS/2 at pyramid level 0 and a keypoint on the pixel grid, the 32 sampling positions are
    x + (2i + 1)/2 - 16 scaled by W/(W-1): the centre sample pair straddles the keypoint."""
    h, w = 96, 128
    img = torch.arange(h * w, dtype=torch.float32).view(h, w) / (h * w)
    p = O.extract_patches(img, torch.tensor([[64.0, 48.0]]), 16.0)
    assert p.shape == (1, 1, 32, 32)
    mid = 0.25 * (p[0, 0, 15, 15] + p[0, 0, 15, 16] + p[0, 0, 16, 15] + p[0, 0, 16, 16])
    gx, gy = 64.0 * w / (w - 1) - 0.5, 48.0 * h / (h - 1) - 0.5           # where kornia's mixed conventions land
    assert abs(float(mid) - (gy * w + gx) / (h * w)) < 1e-4


# ---------------- repeatability evaluation ----------------
@pytest.mark.parametrize("name", list(cases.REPEAT_CASES))
def test_repeatability_oracle_matches_reference_goldens(name):
    g = np.load(os.path.join(G, "repeatability.npz"))
    spec = cases.REPEAT_CASES[name]
    src, dst = cases.repeat_inputs(spec)
    r = O.compute_repeatability(src, dst, **spec["kw"])
    for k, v in r.items():
        ref, v = g[f"{name}.{k}"], np.asarray(v)
        if v.dtype.kind in "iu" or ref.dtype.kind in "iu":
            assert np.array_equal(v.reshape(-1), ref.reshape(-1)), k
        else:
            assert np.allclose(v, ref, rtol=0, atol=1e-12), k


def test_homography_oracle_matches_reference_golden():
    g = np.load(os.path.join(G, "repeatability.npz"))
    src, _ = cases.repeat_inputs(cases.REPEAT_CASES["small"])
    assert np.abs(O.apply_homography_to_points(src, cases.HOMOGRAPHY) - g["homography.points"]).max() < 1e-11


def test_evaluation_glue_oracle_matches_reference_golden():
    g = np.load(os.path.join(G, "repeatability.npz"))
    es, ed, ms, md = cases.eval_inputs(cases.EVAL_CASE)
    res = O.compute_repeatability_with_maximum_filter(es, ed, cases.HOMOGRAPHY, ms, md, cases.EVAL_CASE["nms"],
                                                      cases.EVAL_CASE["num_points"])
    assert np.allclose([float(np.asarray(v[0])) for v in res], g["eval.result"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["poster", "im1"])
def test_natural_images_oracle_matches_reference(sd, name):
    """Round 5: the oracle on a photograph and on the poster (natural.npz, recorded from the reference): the forward's score map
    against the reference's samples, and the NMS / top-K restatement on the reference's own map against the reference's points
    -- constant regions, hard edges and exact ties included."""
    from tests.golden import cases
    f = np.load(os.path.join(G, "natural.npz"))
    im = cases.poster_u8() if name == "poster" else f[name + ".u8"]
    h, w = im.shape[:2]
    k, border, nms = cases.NATURAL_CASES[name]
    pts, prob = O.extract_detections(sd, im / 255.0, nms_size=nms, num_points=k, border_size=border)
    assert np.abs(prob[::8, ::8] - f[name + ".prob_s8"]).max() < 1e-6
    assert np.abs(cases.cfg_mix(prob) - f[name + ".prob_mix"]).max() < 1e-6
    assert np.abs(prob - f[name + ".prob"]).max() < 1e-6
    idx, sc = O.detect_from_prob(f[name + ".prob"], h, w, border, nms, k)
    ref = f[name + ".pts"]
    ri = (ref[:, 1] * w + ref[:, 0]).astype(np.int64)
    o = np.argsort(ri)
    assert np.array_equal(np.sort(idx), ri[o]) and np.array_equal(sc[np.argsort(idx)].astype(np.float64), ref[o, 3])
    sm = O.apply_nms(O.remove_borders(f[name + ".prob"][(prob.shape[0] - h - (h & 1)) // 2:, :][:h, :w], border), nms)
    assert int((sm > 0).sum()) == int(f[name + ".nms_survivors"])
