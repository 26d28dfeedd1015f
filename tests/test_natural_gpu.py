"""Photographs and a poster.  Every other forward fixture is box-blurred noise; these are the two images the
reference's demo is run on (/root/reference/media/im1.jpg, im2.jpg -- demo/demo_match.py:122-142 -- decoded with PIL by
tests/golden/make_golden.py and stored as uint8 arrays) and a synthetic poster with what photographs rarely have: constant black /
white / saturated rectangles with hard edges, a one-pixel checkerboard, one-pixel lines (tests/golden/cases.py: poster_u8).
The expected values come from the imported reference with the synthetic weights: score-map samples, the reference's own
extract_detections (balf/utils/train_utils.py:416-454) and demo_match.detect (demo/demo_match.py:21-57) executed from its source."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from balf_amd import arch, ops, pipeline
from balf_amd.demo import demo_match
from balf_amd.model import get_model
from balf_amd.utils import synth
from tests.golden import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TIGHT = 1e-5           # both precisions measure 2e-6 ... 6e-6 against the reference


def _image(f, name):
    return cases.poster_u8() if name == "poster" else f[name + ".u8"]


@pytest.fixture(scope="module")
def models():
    out = {}
    for prec in ("fp32", "fp16"):
        m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
        m.load_state_dict(synth.synthetic_state_dict(cases.WEIGHT_SEED))
        m.precision = prec
        out[prec] = m.eval().to(DEV)
    return out


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("name", list(cases.NATURAL_CASES))
def test_score_map_vs_reference(models, name, precision):
    """Float input as the reference's callers prepare it AND the fused uint8 input: both within 1e-5 of the reference's map."""
    f = np.load(os.path.join(G, "natural.npz"))
    im = _image(f, name)
    x = pipeline.pad_batch((im / 255.0)[None]).to(DEV)
    with torch.inference_mode():
        prob = models[precision](x, want_logits=False)["prob"][0].cpu().numpy()
        prob8 = models[precision].forward_u8(torch.from_numpy(np.ascontiguousarray(im))[None].to(DEV), want_logits=False)["prob"][0].cpu().numpy()
    assert models[precision].effective_precision == precision
    errs = {"s8": np.abs(prob[::8, ::8] - f[name + ".prob_s8"]).max(),
            "rows": np.abs(prob[cases.CFG_ROWS(prob.shape[0])] - f[name + ".prob_rows"]).max(),
            "mix": np.abs(cases.cfg_mix(prob) - f[name + ".prob_mix"]).max(),
            "cellsum": np.abs(cases.cfg_cellsum(prob) - f[name + ".prob_cellsum"]).max() / 8.0}
    print(name, precision, {k: float(v) for k, v in errs.items()}, "u8 vs float input:", float(np.abs(prob8 - prob).max()))
    assert max(errs.values()) < TIGHT
    assert np.abs(prob8 - prob).max() < 2e-6            # (/255 in float32 on the device vs float64 on the host, then float32)
    if name + ".prob" in f.files:
        assert np.abs(prob - f[name + ".prob"]).max() < TIGHT


@pytest.mark.parametrize("name", list(cases.NATURAL_FULL_PROB))
def test_nms_topk_on_the_reference_score_map(name):
    """crop / border / NMS / top-K kernels on the map the reference's model produced inside its own extract_detections: the
    reference's points exactly (same set, same score bits) -- including the poster's exact ties and constant regions."""
    f = np.load(os.path.join(G, "natural.npz"))
    im = _image(f, name)
    h, w = im.shape[:2]
    k, border, nms = cases.NATURAL_CASES[name]
    prob = torch.from_numpy(f[name + ".prob"])[None].to(DEV)
    _, _, top, left = arch.padded_hw(h, w)
    idx, score, count = ops.nms_topk(prob, top, left, h, w, border, nms, k)
    ref = f[name + ".pts"]
    ri = (ref[:, 1] * w + ref[:, 0]).astype(np.int64)
    o = np.lexsort((ri, -ref[:, 3]))
    n = int(count[0])
    assert n == ref.shape[0]
    assert np.array_equal(idx[0, :n].cpu().numpy().astype(np.int64), ri[o])
    assert np.array_equal(score[0, :n].cpu().numpy().astype(np.float64), ref[o, 3])
    # the dense NMS map: as many survivors as the reference's apply_nms left, in the flat regions too
    nm = ops.window_nms(prob[:, top:top + h, left:left + w].contiguous(), border, nms)[0].cpu().numpy()
    assert int((nm > 0).sum()) == int(f[name + ".nms_survivors"])
    if name == "poster":
        for reg, (y0, y1, x0, x1) in cases.POSTER_FLAT.items():
            assert int((nm[y0:y1, x0:x1] > 0).sum()) == int(f[f"poster.flat_{reg}_survivors"]), reg


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("name", list(cases.NATURAL_CASES))
def test_extract_detections_end_to_end(models, name, precision):
    """The whole caller on the GPU against the reference's caller.  The maps differ by ~3e-6, so points whose score sits within
    that of the K-th score may swap: the gate is the share of the reference's points that are clear of the threshold."""
    f = np.load(os.path.join(G, "natural.npz"))
    im = _image(f, name)
    h, w = im.shape[:2]
    k, border, nms = cases.NATURAL_CASES[name]
    pts, _ = pipeline.extract_detections(im / 255.0, models[precision], DEV, nms_size=nms, num_points=k, border_size=border)
    ref = f[name + ".pts"]
    assert pts.shape == ref.shape and np.all(np.diff(pts[:, 3]) <= 0)
    gi, ri = (pts[:, 1] * w + pts[:, 0]).astype(np.int64), (ref[:, 1] * w + ref[:, 0]).astype(np.int64)
    overlap = np.intersect1d(gi, ri).size / ri.size
    clear = float((ref[:, 3] > ref[-1, 3] + 2 * TIGHT).mean())           # reference points that a 1e-5 change cannot push out
    print(name, precision, "overlap", overlap, "clear of the threshold", clear)
    assert overlap >= min(0.995, clear - 0.002), (overlap, clear)
    gs = dict(zip(gi.tolist(), pts[:, 3])); rs = dict(zip(ri.tolist(), ref[:, 3]))
    assert max(abs(gs[i] - rs[i]) for i in np.intersect1d(gi, ri).tolist()) < TIGHT
    if name == "poster":
        # constant regions: the same number of NMS survivors as in the reference's map (a plateau that broke up or merged would
        # change it)
        x = pipeline.pad_batch((im / 255.0)[None]).to(DEV)
        _, _, top, left = arch.padded_hw(h, w)
        with torch.inference_mode():
            prob = models[precision](x, want_logits=False)["prob"]
        nm = ops.window_nms(prob[:, top:top + h, left:left + w].contiguous(), border, nms)[0].cpu().numpy()
        for reg, (y0, y1, x0, x1) in cases.POSTER_FLAT.items():
            got, want = int((nm[y0:y1, x0:x1] > 0).sum()), int(f[f"poster.flat_{reg}_survivors"])
            print("poster", reg, "survivors", got, "reference", want)
            assert abs(got - want) <= max(1, want // 50), (reg, got, want)


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
@pytest.mark.parametrize("name", list(cases.NATURAL_CASES))
def test_demo_detect_vs_reference(models, name, precision):
    """demo_match.detect (greedy nms_fast path, no sub-pixel step) on the uint8 image, as demo/demo_match.py:122-142 calls it."""
    f = np.load(os.path.join(G, "natural.npz"))
    im = _image(f, name)
    args = SimpleNamespace(**dict(cases.DETECT_ARGS, sub_pixel=False))
    res = demo_match.detect(args, im, models[precision], DEV)
    ref = f[name + ".detect_pts"]
    assert res.shape[1] == ref.shape[1] and abs(res.shape[0] - ref.shape[0]) <= max(2, ref.shape[0] // 50), (res.shape, ref.shape)
    a = {(float(r[0]), float(r[1])) for r in res}
    b = {(float(r[0]), float(r[1])) for r in ref}
    same = len(a & b) / len(b)
    print(name, precision, "demo detect: points", res.shape[0], "reference", ref.shape[0], "same positions", same)
    assert same >= 0.995, same
