#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (the reference is mounted read-only at /root/reference
and never travels to the GPU box):

    python tests/golden/make_golden.py

What is recorded is data only -- inputs are regenerated from seeds, outputs are arrays:
  forward_small.npz   logits/prob (+ per-stage outputs) of the reference model with the
                      seeded synthetic weights of balf_amd.utils.synth on small inputs
  stage_taps.npz      (round 4) per-stage outputs of the reference's down1..down4 (forward hooks) on the two larger small
                      cases, sampled (cases.stage_sample) + per-channel float64 sums
  forward_cfg.npz     strided prob samples + reference top-K index lists at 512x640
  nms_topk.npz        remove_borders/apply_nms/find_index_higher_scores results on synthetic
                      score maps (random, tie-heavy, all-zero, sparse, even window sizes)
  geometry.json       pad/crop shapes+offsets, state-dict table, checkpoint-loader behaviour
  repeatability.npz   compute_repeatability / apply_homography_to_points results; those two modules import
                      torchvision / cv2 / torchgeometry (absent offline), so the reference's own function bodies are
                      extracted with ast from the files under /root/reference and executed as they are
  natural.npz         (round 5) the two photographs of /root/reference/media (decoded with PIL, stored as uint8 arrays: data) and
                      a synthetic poster (cases.poster_u8): score-map samples, extract_detections and demo_match.detect results of
                      the reference on them, the whole score map for two of them; `python tests/golden/make_golden.py natural`
                      writes this file alone
  hardnet.npz         descriptors (+ per-layer activations of one patch) of the reference's HardNet class with the
                      seeded synthetic weights, on synthetic patches
"""
import json
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
warnings.filterwarnings("ignore")

from balf.model import get_model as ref_get_model          # noqa: E402  (reference)
from balf.utils import test_utils as RT                     # noqa: E402  (reference)
from third_party.hardnet.hardnet_pytorch import HardNet     # noqa: E402  (reference)

from balf_amd.utils import synth                            # noqa: E402
from tests.golden import cases                              # noqa: E402

torch.set_num_threads(8)


def ref_functions(path, names, extra=None, drop_assign=None):
    """Compile the named top-level functions of a reference source file, unchanged, into a fresh namespace (the
    module itself cannot be imported here: it pulls in packages that are not installed).  `drop_assign` = a variable
    name whose (single) plain assignment statement is left out -- used for ONE statement of the reference that raises on
    the reference's own model (see ref_extract_detections)."""
    import ast
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names)
    if drop_assign:
        hit = 0
        for fn in keep:
            body = [st for st in fn.body if not (isinstance(st, ast.Assign) and len(st.targets) == 1 and
                                                 isinstance(st.targets[0], ast.Name) and st.targets[0].id == drop_assign)]
            hit += len(fn.body) - len(body)
            fn.body = body
        assert hit == 1, hit
    ns = {"np": np}
    ns.update(extra or {})
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def ref_model(seed):
    cfg = RT.get_cfg_from_yaml_file("/root/reference/balf/configs/test.yaml")
    m = ref_get_model.load_model(cfg["model"]).eval()
    sd = synth.synthetic_state_dict(seed)
    assert list(sd.keys()) == list(m.state_dict().keys())
    m.load_state_dict(sd)
    return m, cfg


def ref_extract_detections(unmodified=False):
    """train_utils.extract_detections (balf/utils/train_utils.py:416-454) compiled from the reference's source, with the
    modules it calls into (dataset_utils, geometry_tools, repeatability_tools -- none importable here) supplied the same way.

    As published the function RAISES on the reference's own model: :444 indexes `output['prob']` -- which DetectorHead
    squeezes to [B,H,W] (decoder.py:27) -- with four subscripts (IndexError).  That statement only feeds the second return
    value; `unmodified=False` leaves that one assignment out (the name then resolves to a module-level None) and executes
    everything the keypoints depend on exactly as written."""
    import types
    from scipy.ndimage import maximum_filter
    du = ref_functions("/root/reference/balf/datasets/dataset_utils.py", ["make_shape_even", "mod_padding_symmetric"])
    gt = ref_functions("/root/reference/balf/benchmark_test/geometry_tools.py",
                       ["remove_borders", "get_point_coordinates", "find_index_higher_scores"])
    rt = ref_functions("/root/reference/balf/benchmark_test/repeatability_tools.py", ["apply_nms"],
                       {"maximum_filter": maximum_filter})
    ns = {"torch": torch, "dataset_utils": types.SimpleNamespace(**du), "geometry_tools": types.SimpleNamespace(**gt),
          "repeatability_tools": types.SimpleNamespace(**rt), "score_map_batch": None}
    return ref_functions("/root/reference/balf/utils/train_utils.py", ["extract_detections"], ns,
                         drop_assign=None if unmodified else "score_map_batch")["extract_detections"]


def ref_demo_detect():
    """demo_match.detect (demo/demo_match.py:21-57) compiled from the reference's source; its one module dependency,
    balf.utils.test_utils, is the importable reference module itself."""
    return ref_functions("/root/reference/demo/demo_match.py", ["detect"], {"torch": torch, "test_utils": RT})["detect"]


def pts_to_idx(pts, w):
    """rows [x, y, 1, score] -> (flat indices sorted ascending, their scores)"""
    if pts.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    idx = (pts[:, 1].astype(np.int64) * w + pts[:, 0].astype(np.int64))
    sc = pts[:, 3].astype(np.float32)
    o = np.argsort(idx)
    return idx[o], sc[o]


def make_torchgeometry_stub():
    """A module object standing in for torchgeometry with a restatement of its one call on this path (see main)."""
    import types

    class SpatialSoftArgmax2d(torch.nn.Module):
        def __init__(self, normalized_coordinates=True):
            super().__init__()
            self.normalized_coordinates = normalized_coordinates
            self.eps = 1e-6

        def forward(self, input):
            b, c, hh, ww = input.shape
            x = input.view(b, c, -1)
            exp_x = torch.exp(x - torch.max(x, dim=-1, keepdim=True)[0])
            exp_x_sum = 1.0 / (exp_x.sum(dim=-1, keepdim=True) + self.eps)
            if self.normalized_coordinates:
                xs, ys = torch.linspace(-1, 1, ww), torch.linspace(-1, 1, hh)
            else:
                xs, ys = torch.linspace(0, ww - 1, ww), torch.linspace(0, hh - 1, hh)
            pos_y, pos_x = torch.meshgrid(ys, xs, indexing="ij")
            pos_x, pos_y = pos_x.reshape(-1).to(input.dtype), pos_y.reshape(-1).to(input.dtype)
            expected_y = torch.sum((pos_y * exp_x) * exp_x_sum, dim=-1, keepdim=True)
            expected_x = torch.sum((pos_x * exp_x) * exp_x_sum, dim=-1, keepdim=True)
            return torch.cat([expected_x, expected_y], dim=-1).view(b, c, 2)

    tgm = types.ModuleType("torchgeometry")
    tgm.contrib = types.ModuleType("torchgeometry.contrib")
    tgm.contrib.SpatialSoftArgmax2d = SpatialSoftArgmax2d
    return tgm


def natural_section(m):
    """Photographs and a poster through the reference's own callers."""
    import types
    from PIL import Image
    extract, detect = ref_extract_detections(), ref_demo_detect()
    nat, seen = {}, {}
    hook = m.register_forward_hook(lambda mod, i, o: seen.__setitem__("prob", o["prob"][0].detach().numpy().copy()))
    for name, (k, border, nms) in cases.NATURAL_CASES.items():
        if name == "poster":
            im = cases.poster_u8()
        else:
            im = np.asarray(Image.open(f"/root/reference/media/{name}.jpg").convert("RGB"))
            nat[name + ".u8"] = im                       # the decoded pixels: data, not source
        h, w = im.shape[:2]
        pts, _ = extract(im / 255.0, m, "cpu", nms_size=nms, num_points=k, border_size=border)
        prob = seen.pop("prob")
        nat[name + ".pts"] = pts
        nat[name + ".prob_s8"] = prob[::8, ::8].copy()
        nat[name + ".prob_rows"] = prob[cases.CFG_ROWS(prob.shape[0])].copy()
        nat[name + ".prob_mix"] = cases.cfg_mix(prob)
        nat[name + ".prob_cellsum"] = cases.cfg_cellsum(prob)
        if name in cases.NATURAL_FULL_PROB:
            nat[name + ".prob"] = prob
        # the NMS map the reference computes on the way (crop -> remove_borders -> apply_nms): survivors and exact ties
        hp, wp = prob.shape
        top, left = (hp - (h + (h & 1))) // 2, (wp - (w + (w & 1))) // 2
        sm = RT.apply_nms(RT.remove_borders(prob[top:top + h, left:left + w], borders=border), nms)
        nat[name + ".nms_survivors"] = np.int64((sm > 0).sum())
        vals = np.sort(sm[sm > 0])
        nat[name + ".nms_tied_survivors"] = np.int64((np.diff(vals) == 0).sum())
        if name == "poster":
            for reg, (y0, y1, x0, x1) in cases.POSTER_FLAT.items():
                nat[f"{name}.flat_{reg}_survivors"] = np.int64((sm[y0:y1, x0:x1] > 0).sum())
        args = types.SimpleNamespace(**dict(cases.DETECT_ARGS, sub_pixel=False))
        res = detect(args, im, m, "cpu")
        nat[name + ".detect_pts"] = np.asarray(res, dtype=np.float64)
        print(name, im.shape, "survivors", int(nat[name + ".nms_survivors"]), "tied", int(nat[name + ".nms_tied_survivors"]),
              "K-th score", float(pts[-1, 3]), "detect", nat[name + ".detect_pts"].shape)
    hook.remove()
    np.savez_compressed(os.path.join(HERE, "natural.npz"), **nat)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "natural":
        natural_section(ref_model(cases.WEIGHT_SEED)[0])
        return
    out = {}
    # ---------------- forward, small ----------------
    m, cfg = ref_model(cases.WEIGHT_SEED)
    fw = {}
    taps = {}
    for name, (b, h, w, seed) in cases.FORWARD_SMALL.items():
        x = cases.forward_input(b, h, w, seed)
        stage_out = {}
        hooks = []
        if name == cases.TAP_CASE or name in cases.TAP_SAMPLED:
            for s in ("down1", "down2", "down3", "down4"):
                hooks.append(getattr(m, s).register_forward_hook(
                    lambda mod, i, o, s=s: stage_out.__setitem__(s, o.detach().numpy().copy())))
        with torch.inference_mode():
            o = m(x)
        for hk in hooks:
            hk.remove()
        fw[name + ".logits"] = o["logits"].numpy()
        fw[name + ".prob"] = o["prob"].numpy()
        for s, v in stage_out.items():
            if name == cases.TAP_CASE:
                fw[f"{name}.{s}"] = v      # NCHW, as the reference's Down returns it
            else:                          # (round 4) the larger cases: sampled, in a file of their own
                taps[f"{name}.{s}.sample"], taps[f"{name}.{s}.chansum"] = cases.stage_sample(v)
    np.savez_compressed(os.path.join(HERE, "forward_small.npz"), **fw)
    np.savez_compressed(os.path.join(HERE, "stage_taps.npz"), **taps)

    natural_section(m)

    # ---------------- forward at a config size (strided samples + detections) ----------------
    # The score map is taken from inside the reference's own caller: extract_detections runs pad -> model -> crop ->
    # remove_borders -> apply_nms -> get_point_coordinates -> sort itself, a forward hook on the model records the padded
    # score map it produced on the way (one forward per size: 1.3 / 3 / 7 s).
    extract = ref_extract_detections()
    fc = {}
    seen = {}
    hook = m.register_forward_hook(lambda mod, i, o: seen.__setitem__("prob", o["prob"][0].detach().numpy().copy()))
    for name, (h, w, k, img_index) in cases.FORWARD_CFG.items():
        img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(h, w, img_index))
        pts, none = extract(img, m, "cpu", nms_size=15, num_points=k, border_size=15)
        assert none is None and pts.shape == (k, 4)
        prob = seen.pop("prob")
        fc[name + ".prob_s8"] = prob[::8, ::8].copy()
        fc[name + ".prob_rows"] = prob[cases.CFG_ROWS(prob.shape[0])].copy()
        idx, sc = pts_to_idx(pts, w)
        fc[name + ".idx"] = idx.astype(np.int32)
        fc[name + ".score"] = sc
        if name != "vga":                   # (the vga entry keeps its round-1 key set)
            fc[name + ".prob_mix"] = cases.cfg_mix(prob)
            fc[name + ".prob_cellsum"] = cases.cfg_cellsum(prob)
            fc[name + ".pts"] = pts         # the caller's own return value: rows [x, y, 1.0, score] float64, score descending
    np.savez_compressed(os.path.join(HERE, "forward_cfg.npz"), **fc)

    # ---------------- the two callers, executed from the reference's source ----------------
    import types
    ck2 = {}
    for name, (h, w, k, img_index, border, nms) in cases.EXTRACT_CASES.items():
        img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(h, w, img_index))
        pts, _ = extract(img, m, "cpu", nms_size=nms, num_points=k, border_size=border)
        ck2[name + ".pts"] = pts
        ck2[name + ".prob"] = seen.pop("prob")
    hook.remove()
    try:
        ref_extract_detections(unmodified=True)(synth.gray_to_rgb_norm(synth.synthetic_gray_u8(64, 64, 0)), m, "cpu")
        unmodified = "ok"
    except IndexError as e:
        unmodified = f"IndexError: {e}"
    detect = ref_demo_detect()
    # sub_pixel=True reaches torchgeometry (absent): the restatement of its one call, defined below for subpixel.npz, is
    # installed for these cases too -- same caveat: the reference's side is pinned, the third-party call is not
    tgm_stub = make_torchgeometry_stub()
    sys.modules["torchgeometry"], sys.modules["torchgeometry.contrib"] = tgm_stub, tgm_stub.contrib
    for name, (h, w, img_index, over) in cases.DETECT_CASES.items():
        args = types.SimpleNamespace(**dict(cases.DETECT_ARGS, **over))
        res = detect(args, cases.detect_input(h, w, img_index), m, "cpu")
        if isinstance(res, tuple):          # the empty case returns a PAIR (np.zeros([0,3]), np.zeros([0,1])), demo_match.py:51-52
            ck2[name + ".empty_pair_shapes"] = np.asarray([r.shape for r in res])
        else:
            ck2[name + ".pts"] = np.asarray(res, dtype=np.float64)
    del sys.modules["torchgeometry"], sys.modules["torchgeometry.contrib"]
    np.savez_compressed(os.path.join(HERE, "callers.npz"), **ck2)

    # ---------------- NMS / top-K on synthetic score maps ----------------
    nk = {}
    for name, spec in cases.NMS_CASES.items():
        score = cases.nms_input(spec)
        h, w = score.shape
        rb = RT.remove_borders(score, borders=spec["border"])
        nms = RT.apply_nms(rb, spec["nms"])
        nk[name + ".nms_nonzero"] = np.flatnonzero(nms.ravel() != 0).astype(np.int32)
        idxs = RT.find_index_higher_scores(nms, num_points=spec["k"])
        flat = (idxs[:, 0].astype(np.int64) * w + idxs[:, 1]).astype(np.int32)
        nk[name + ".idx"] = flat                       # raster order, as argwhere returns it
        nk[name + ".score"] = nms.ravel()[flat].astype(np.float32)
        # the same selection with an explicit threshold (`threshold != -1`, test_utils.py:91-95): quantiles of the
        # positive NMS values (fewer / more than k pixels reach them), a value nothing reaches, and <= 0 (every pixel)
        pos = np.sort(nms.ravel()[nms.ravel() > 0])
        thr = [float(pos[int(q * (pos.size - 1))]) for q in (0.1, 0.5, 0.9)] if pos.size else []
        thr += [2.0, 0.0, -0.5]
        nk[name + ".thr"] = np.asarray(thr, dtype=np.float64)
        for j, t in enumerate(thr):
            idt = RT.find_index_higher_scores(nms, num_points=spec["k"], threshold=t)
            nk[f"{name}.thr{j}.idx"] = (idt[:, 0].astype(np.int64) * w + idt[:, 1]).astype(np.int32) if len(idt) else np.zeros(0, np.int32)
    np.savez_compressed(os.path.join(HERE, "nms_topk.npz"), **nk)

    # ---------------- greedy nms_fast path of the demo (no sub-pixel: torchgeometry is not installed) --------
    gk = {}
    for name, spec in cases.GREEDY_CASES.items():
        score = cases.nms_input(spec)
        rb = RT.remove_borders(score, borders=spec["border"])
        pts = RT.get_points_direct_from_score_map(heatmap=rb, conf_thresh=spec["conf"], nms_size=spec["nms"],
                                                  subpixel=False, order_coord="xysr")
        if pts.size == 0:
            gk[name + ".idx"] = np.zeros(0, np.int32); gk[name + ".score"] = np.zeros(0, np.float32)
        else:
            gk[name + ".idx"] = (pts[:, 1].astype(np.int64) * score.shape[1] + pts[:, 0].astype(np.int64)).astype(np.int32)
            gk[name + ".score"] = pts[:, 3].astype(np.float32)      # emitted sorted by confidence, descending
    np.savez_compressed(os.path.join(HERE, "greedy_nms.npz"), **gk)

    # ---------------- sub-pixel refinement of the demo path ----------------
    # torchgeometry (requirements.txt:7, ">=0.1.2", not installed and not installable offline) supplies ONE call of this
    # path, contrib.SpatialSoftArgmax2d (test_utils.py:198-201).  Everything around it -- threshold, nms_fast, the patch
    # extraction with its padding and truncation, norm_patches, do_log, the coordinate update -- is the reference's own code
    # and is executed here unchanged; the one missing call is supplied by the restatement below of its published definition
    # (torchgeometry 0.1.2, contrib/spatial_soft_argmax2d.py: soft-max over the flattened patch with the max-subtraction
    # trick and eps = 1e-6 in the normaliser, expectation of the pixel grid, output (x, y)).  So these vectors pin the
    # reference's side of the sub-pixel path; the third-party call itself stays unpinned.
    tgm = make_torchgeometry_stub()
    sys.modules["torchgeometry"], sys.modules["torchgeometry.contrib"] = tgm, tgm.contrib
    sk = {}
    for name, patch in cases.SUBPIXEL_CASES:
        spec = cases.GREEDY_CASES[name]
        rb = RT.remove_borders(cases.nms_input(spec), borders=spec["border"])
        pts = RT.get_points_direct_from_score_map(heatmap=rb, conf_thresh=spec["conf"], nms_size=spec["nms"],
                                                  subpixel=True, patch_size=patch, order_coord="xysr")
        sk[f"{name}.p{patch}"] = np.asarray(pts, dtype=np.float64)          # rows [x, y, 1, score]
    del sys.modules["torchgeometry"], sys.modules["torchgeometry.contrib"]
    np.savez_compressed(os.path.join(HERE, "subpixel.npz"), **sk)

    # ---------------- stand-alone nms_fast on a corner list ----------------
    ck = {}
    for name, (h, w, n, dist, seed) in cases.NMS_FAST_CASES.items():
        out, out_inds = RT.nms_fast(cases.nms_fast_input(h, w, n, seed), h, w, dist)
        ck[name + ".out"] = np.asarray(out, dtype=np.float64)
        ck[name + ".inds"] = np.asarray(out_inds, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "nms_fast.npz"), **ck)

    # ---------------- HardNet descriptor (demo path) ----------------
    hn = HardNet().eval()
    hsd = synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED)
    assert list(hsd.keys()) == list(hn.state_dict().keys())
    hn.load_state_dict(hsd)
    hk = {}
    for name, (n, seed) in cases.HARDNET_CASES.items():
        x = synth.synthetic_patches(n, seed)
        acts = []
        hooks = []
        if name == cases.HARDNET_TAP_CASE:
            for i in (2, 5, 8, 11, 14, 17, 20):          # the ReLU after each conv+BN, and the last BN
                hooks.append(hn.features[i].register_forward_hook(lambda mod, inp, out: acts.append(out.detach().numpy().copy())))
        with torch.inference_mode():
            hk[name + ".desc"] = hn(x).numpy()
        for h_ in hooks:
            h_.remove()
        for j, a in enumerate(acts):
            hk[f"{name}.act{j}"] = a[0, ::4].copy()
    hk["state_keys"] = np.array(list(hn.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "hardnet.npz"), **hk)

    # ---------------- repeatability evaluation ----------------
    rt = ref_functions("/root/reference/balf/benchmark_test/repeatability_tools.py",
                       ["compute_repeatability", "intersection_area", "union_area"])
    gt = ref_functions("/root/reference/balf/benchmark_test/geometry_tools.py", ["apply_homography_to_points", "getAff"])
    rp = {}
    for name, spec in cases.REPEAT_CASES.items():
        src, dst = cases.repeat_inputs(spec)
        res = rt["compute_repeatability"](src, dst, **spec["kw"])
        for k_, v_ in res.items():
            rp[f"{name}.{k_}"] = np.asarray(v_)
    src, _ = cases.repeat_inputs(cases.REPEAT_CASES["small"])
    rp["homography.points"] = gt["apply_homography_to_points"](src, cases.HOMOGRAPHY)
    # the evaluation glue around them (train_utils.py:170-196), also executed from the reference's source
    import types
    from scipy.ndimage import maximum_filter
    rt2 = ref_functions("/root/reference/balf/benchmark_test/repeatability_tools.py",
                        ["compute_repeatability", "intersection_area", "union_area", "apply_nms"],
                        {"maximum_filter": maximum_filter})
    gt2 = ref_functions("/root/reference/balf/benchmark_test/geometry_tools.py",
                        ["apply_homography_to_points", "getAff", "get_point_coordinates", "find_index_higher_scores"])
    tu = ref_functions("/root/reference/balf/utils/train_utils.py", ["compute_repeatability_with_maximum_filter"],
                       {"repeatability_tools": types.SimpleNamespace(**rt2), "geometry_tools": types.SimpleNamespace(**gt2)})
    es, ed, ms, md = cases.eval_inputs(cases.EVAL_CASE)
    res = tu["compute_repeatability_with_maximum_filter"](es, ed, cases.HOMOGRAPHY, ms, md, cases.EVAL_CASE["nms"],
                                                          cases.EVAL_CASE["num_points"])
    rp["eval.result"] = np.asarray([float(np.asarray(v[0])) for v in res])
    hl = ref_functions("/root/reference/balf/benchmark_test/repeatability_tools.py", ["check_common_points", "select_top_k"])
    src, _ = cases.repeat_inputs(cases.REPEAT_CASES["train_eval"])
    kp = np.stack([src[:, 1] * 0.3 + 1, src[:, 0] * 0.3 + 1, src[:, 2], src[:, 3]], axis=1)      # rows [y, x, s, score]
    rp["helpers.common"] = hl["check_common_points"](kp, ms)
    rp["helpers.topk"] = hl["select_top_k"](kp, 40)
    np.savez_compressed(os.path.join(HERE, "repeatability.npz"), **rp)

    # ---------------- geometry, state-dict table, loader behaviour ----------------
    geo = {"pad": {}, "state": [], "loader": {}}
    for (h, w) in cases.PAD_SIZES:
        img = np.zeros((h, w, 3))
        ev = RT.make_shape_even(img)
        pd = RT.mod_padding_symmetric(ev, factor=64)
        hp, wp = pd.shape[:2]
        # where the original image's (0,0) lands in the padded array
        probe = np.zeros((h, w, 3)); probe[0, 0, 0] = 1.0
        pp = RT.mod_padding_symmetric(RT.make_shape_even(probe), factor=64)
        y0, x0 = np.argwhere(pp[:, :, 0] == 1.0)[0]
        geo["pad"][f"{h}x{w}"] = {"even": list(ev.shape[:2]), "padded": [hp, wp], "origin": [int(y0), int(x0)],
                                   "h_start": hp // 2 - ev.shape[0] // 2, "w_start": wp // 2 - ev.shape[1] // 2}
    geo["state"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]

    with tempfile.TemporaryDirectory() as td:
        sd = synth.synthetic_state_dict(7)
        full = os.path.join(td, "full.pth")
        torch.save({"epoch": 12, "repeatability": 0.5, "model_state": sd, "optimizer_state": None}, full)
        m2, _ = ref_model(0)
        geo["loader"]["full"] = list(ref_get_model.load_test_pretrained_model(m2, full, device="cpu"))
        geo["loader"]["full_probe"] = float(m2.state_dict()["down3.conv2.bias"][5])
        bare = os.path.join(td, "bare.pth")
        torch.save({"model_state": sd}, bare)
        geo["loader"]["bare"] = list(ref_get_model.load_test_pretrained_model(m2, bare, device="cpu"))
        for tag, mut in (("missing_key", lambda d: d.pop("down2.conv2.bias")),
                         ("wrong_shape", lambda d: d.__setitem__("down1.conv.0.weight", torch.zeros(32, 4))),
                         ("extra_key", lambda d: d.__setitem__("not.a.key", torch.zeros(1)))):
            d = dict(sd); mut(d)
            p = os.path.join(td, tag + ".pth")
            torch.save({"model_state": d}, p)
            try:
                ref_get_model.load_test_pretrained_model(m2, p, device="cpu")
                geo["loader"][tag] = "ok"
            except AssertionError:
                geo["loader"][tag] = "AssertionError"
        try:
            ref_get_model.load_test_pretrained_model(m2, os.path.join(td, "nope.pth"), device="cpu")
            geo["loader"]["missing_file"] = "ok"
        except FileNotFoundError:
            geo["loader"]["missing_file"] = "FileNotFoundError"
    geo["extract_detections_unmodified"] = unmodified        # what train_utils.extract_detections does as published
    geo["versions"] = {"torch": torch.__version__, "numpy": np.__version__}
    with open(os.path.join(HERE, "geometry.json"), "w") as f:
        json.dump(geo, f, indent=1)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
