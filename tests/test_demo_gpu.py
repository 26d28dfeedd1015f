"""Demo path beyond the detector: patch extraction and mutual-NN matching through the C ABI
against the oracle's restatement of kornia (parity unpinned: kornia is absent offline), and the whole
demo_match pipeline stage by stage."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from balf_amd import arch, ops                                               # noqa: E402
from balf_amd.demo import demo_match                                   # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet      # noqa: E402
from balf_amd.utils import synth                           # noqa: E402
from oracle import oracle                                             # noqa: E402
from tests.golden import cases                                        # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _points(h, w, n, seed):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], axis=1).astype(np.float32)
    xy[:4] = [[0, 0], [w - 1, h - 1], [0.25, h - 1.5], [w - 1, 0]]            # corners: border clamping
    return xy


@pytest.mark.parametrize("h,w,scale", [(480, 640, 60.0), (480, 640, 12.0), (301, 517, 60.0), (256, 256, 130.0),
                                         (70, 90, 60.0)])
def test_extract_patches_vs_oracle(h, w, scale):
    gray = synth.synthetic_gray_u8(h, w, 3)
    xy = _points(h, w, 200, h + w)
    ref = oracle.extract_patches(torch.from_numpy(gray.astype(np.float32) / 255.0), torch.from_numpy(xy), scale)
    got = ops.extract_patches(torch.from_numpy(gray).to(DEV), torch.from_numpy(xy).to(DEV), scale).cpu()
    assert got.shape == ref.shape == (200, 1, 32, 32)
    # sampling positions are fp32 numbers up to ~W (ulp 6e-5 at 640): a different but equivalent order of the
    # coordinate arithmetic moves a bilinear weight by that much, i.e. the sample by ~1e-5 x local contrast
    assert (got - ref).abs().max() < 1e-4


def _descs(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 128)).astype(np.float32)
    return d / np.linalg.norm(d, axis=1, keepdims=True)


@pytest.mark.parametrize("n1,n2,th", [(500, 700, 0.99), (2048, 2048, 0.99), (37, 1000, 0.8), (1000, 17, 0.95),
                                      (2, 2, 0.99), (1, 50, 0.99), (50, 1, 0.99)])
def test_match_smnn_vs_oracle(n1, n2, th):
    d1 = _descs(n1, 1)
    rng = np.random.default_rng(2)
    d2 = _descs(n2, 3)
    m = min(n1, n2) // 2                                  # plant m noisy correspondences at shuffled positions
    if m:
        src, dst = rng.permutation(n1)[:m], rng.permutation(n2)[:m]
        noisy = d1[src] + 0.15 * rng.standard_normal((m, 128)).astype(np.float32)
        d2[dst] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    rd, ri = oracle.match_smnn(torch.from_numpy(d1), torch.from_numpy(d2), th)
    gd, gi = ops.match_smnn(torch.from_numpy(d1).to(DEV), torch.from_numpy(d2).to(DEV), th)
    assert gi.dtype == torch.int64 and gd.shape == (gi.shape[0], 1)
    assert torch.equal(gi.cpu(), ri)
    if len(rd):
        assert (gd.cpu().view(-1) - rd).abs().max() < 1e-5
        assert m == 0 or len(ri) >= m // 2


def test_match_smnn_duplicate_descriptors():
    """Two identical candidates: d_second = 0 for a perfect match gives 0/0 = NaN, which fails the ratio test in the
    reference's formulation too; the lowest index wins the arg-min."""
    d1 = _descs(20, 5)
    d2 = np.concatenate([d1[:10], d1[:10], _descs(10, 6)], axis=0)
    rd, ri = oracle.match_smnn(torch.from_numpy(d1), torch.from_numpy(d2), 0.99)
    gd, gi = ops.match_smnn(torch.from_numpy(d1).to(DEV), torch.from_numpy(d2).to(DEV), 0.99)
    assert torch.equal(gi.cpu(), ri)


@pytest.fixture(scope="module")
def models():
    det = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    det.load_state_dict(synth.synthetic_state_dict(cases.WEIGHT_SEED))
    det = det.eval().to(DEV)
    hn = HardNet()
    hn.load_state_dict(synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED))
    return det, hn.eval().to(DEV)


def test_demo_pipeline_stage_by_stage(models):
    det, hn = models
    args = demo_match.DEFAULT_ARGS
    g1 = synth.synthetic_gray_u8(240, 320, 11, blur=7)
    g2 = np.roll(g1, (3, 5), axis=(0, 1))                              # the same scene shifted by (dx, dy) = (5, 3)
    rgb1, rgb2 = np.stack([g1] * 3, -1), np.stack([g2] * 3, -1)
    k1, d1 = demo_match.extract_features(args, rgb1, g1, det, hn, DEV)
    k2, d2 = demo_match.extract_features(args, rgb2, g2, det, hn, DEV)
    assert k1.shape[1] == 2 and d1.shape == (k1.shape[0], 128) and 0 < k1.shape[0] <= args.num_features
    # stage: patches and descriptors against the oracle, from the GPU's keypoints
    ref_p = oracle.extract_patches(torch.from_numpy(g1.astype(np.float32) / 255.0), torch.from_numpy(k1).float(),
                                   float(args.s_mult))
    ref_d = oracle.hardnet_forward(synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED), ref_p).numpy()
    assert np.abs(d1 - ref_d).max() < 1e-4
    # stage: matches against the oracle, from the GPU's descriptors
    _, ri = oracle.match_smnn(torch.from_numpy(d1), torch.from_numpy(d2), 0.99)
    p1, p2 = demo_match.extract_matches(args, rgb1, g1, rgb2, g2, det, hn, DEV)
    assert p1.shape == p2.shape == (len(ri), 2)
    assert np.array_equal(p1, k1[ri[:, 0].numpy()]) and np.array_equal(p2, k2[ri[:, 1].numpy()])
    # sanity of the whole chain: an image matched against itself pairs every keypoint with itself
    # (with random detector weights the keypoints do not follow the image content, so a shifted copy proves nothing)
    s1, s2 = demo_match.extract_matches(args, rgb1, g1, rgb1, g1, det, hn, DEV)
    assert len(s1) == k1.shape[0] and np.array_equal(s1, s2) and np.array_equal(s1, k1)


def test_detect_mirror_shape(models):
    det, _ = models
    g = synth.synthetic_gray_u8(120, 160, 2)
    pts = demo_match.detect(demo_match.DEFAULT_ARGS, np.stack([g] * 3, -1), det, DEV)
    assert pts.ndim == 2 and pts.shape[1] == 3 and (pts[:, 2] == 1.0).all()


def test_demo_main_sequence(tmp_path):
    """The statements of the reference's demo ``__main__`` (demo/demo_match.py:120-147) with balf_amd's modules
    swapped in: checkpoints on disk, get_model loaders, HardNet.load_state_dict(checkpoint['state_dict'])."""
    from balf_amd.utils import test_utils
    det_ckpt, hn_ckpt = tmp_path / "balf.pth", tmp_path / "HardNet++.pth"
    torch.save({"epoch": 3, "repeatability": 0.7, "model_state": synth.synthetic_state_dict(cases.WEIGHT_SEED)}, det_ckpt)
    torch.save({"state_dict": synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED)}, hn_ckpt)
    device = torch.device("cuda")
    cfg = {"model": arch.DEFAULT_MODEL_CFG}
    detector = get_model.load_model(cfg["model"])
    epoch, rep = get_model.load_test_pretrained_model(model=detector, filename=str(det_ckpt))
    assert (epoch, rep) == (3, 0.7)
    detector = detector.eval().to(device)
    descriptor = HardNet()
    checkpoint_descriptor = torch.load(str(hn_ckpt), weights_only=True)
    descriptor.load_state_dict(checkpoint_descriptor["state_dict"])
    descriptor = descriptor.eval().to(device)
    g1 = synth.synthetic_gray_u8(200, 264, 21, blur=7)
    g2 = synth.synthetic_gray_u8(200, 264, 22, blur=7)
    m1, m2 = demo_match.extract_matches(demo_match.DEFAULT_ARGS, np.stack([g1] * 3, -1), g1, np.stack([g2] * 3, -1), g2,
                                        detector, descriptor, device)
    assert m1.shape == m2.shape and m1.shape[1] == 2 and m1.dtype == np.float64
    assert (m1[:, 0] >= 0).all() and (m1[:, 0] <= 263).all() and (m1[:, 1] >= 0).all() and (m1[:, 1] <= 199).all()
    # and the reference's own post-processing helpers on a score map, as demo_match.detect uses them
    with torch.inference_mode():
        prob = detector(torch.zeros(1, 3, 256, 320, device=device))["prob"][0].cpu().numpy()
    pts = test_utils.get_points_direct_from_score_map(heatmap=test_utils.remove_borders(prob, borders=15), conf_thresh=0.001,
                                                      nms_size=15, subpixel=True, patch_size=4, order_coord="xysr")
    assert pts.ndim == 2 and pts.shape[1] == 4


def test_batched_detect_and_describe_equals_per_image(models):
    """One pass over a batch (detector, greedy NMS, patches, HardNet in single launches) gives, per image, exactly what
    the per-image demo functions give."""
    det, hn = models
    args = demo_match.DEFAULT_ARGS
    grays = [synth.synthetic_gray_u8(200, 264, 40 + i, blur=5 if i % 2 else 7) for i in range(3)]
    batch = torch.from_numpy(np.stack(grays)).to(DEV)
    xy, desc, count = demo_match.detect_and_describe_batch(args, batch, det, hn)
    assert xy.shape[0] == desc.shape[0] == 3 and desc.shape[2] == 128
    for i, g in enumerate(grays):
        k1, d1 = demo_match.extract_features(args, np.stack([g] * 3, -1), g, det, hn, DEV)
        n = int(count[i])
        assert n == k1.shape[0]
        assert np.array_equal(xy[i, :n].double().cpu().numpy(), k1)
        assert np.array_equal(desc[i, :n].cpu().numpy(), d1)


def test_match_smnn_batch_equals_per_pair():
    rng = np.random.default_rng(8)
    P, K = 5, 300
    d1 = np.stack([_descs(K, 10 + i) for i in range(P)])
    d2 = np.stack([_descs(K, 50 + i) for i in range(P)])
    n1 = np.array([300, 257, 1, 64, 0], np.int32)
    n2 = np.array([300, 300, 200, 2, 100], np.int32)
    for i in range(P):                                    # plant correspondences inside the valid ranges
        m = min(n1[i], n2[i]) // 2
        if m:
            noisy = d1[i, :m] + 0.1 * rng.standard_normal((m, 128)).astype(np.float32)
            d2[i, n2[i] - m:n2[i]] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    t1, t2 = torch.from_numpy(d1).to(DEV), torch.from_numpy(d2).to(DEV)
    dist, idx, count = ops.match_smnn_batch(t1, torch.from_numpy(n1), t2, torch.from_numpy(n2), 0.95)
    for i in range(P):
        c = int(count[i])
        if n1[i] == 0 or n2[i] == 0:
            assert c == 0
            continue
        gd, gi = ops.match_smnn(t1[i, :n1[i]].contiguous(), t2[i, :n2[i]].contiguous(), 0.95)
        assert c == gi.shape[0]
        assert torch.equal(idx[i, :c].long(), gi) and torch.equal(dist[i, :c], gd.view(-1))
        assert bool((idx[i, c:] == -1).all())


def test_rgb_to_gray_matches_pil():
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(2, 67, 91, 3), dtype=np.uint8)
    rgb[0, 0, :8] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [254, 255, 255], [128, 127, 129]]
    ref = np.stack([np.array(Image.fromarray(im).convert("L")) for im in rgb])
    got = ops.rgb_to_gray_u8(torch.from_numpy(rgb).to(DEV)).cpu().numpy()
    assert np.array_equal(got, ref)


def test_batched_rgb_input(models):
    det, hn = models
    rng = np.random.default_rng(4)
    rgb = torch.from_numpy(rng.integers(0, 256, size=(2, 120, 160, 3), dtype=np.uint8)).to(DEV)
    xy, desc, count = demo_match.detect_and_describe_batch(demo_match.DEFAULT_ARGS, rgb, det, hn)
    gray = ops.rgb_to_gray_u8(rgb)
    xy2, desc2, count2 = demo_match.detect_and_describe_batch(demo_match.DEFAULT_ARGS, rgb, det, hn, gray_u8=gray)
    assert torch.equal(xy, xy2) and torch.equal(desc, desc2) and torch.equal(count, count2)


def test_demo_command_line(tmp_path):
    """python -m balf_amd.demo.demo_match with checkpoints and images on disk, as the reference demo is used."""
    from PIL import Image
    det_ckpt, hn_ckpt = tmp_path / "balf.pth", tmp_path / "HardNet++.pth"
    torch.save({"model_state": synth.synthetic_state_dict(cases.WEIGHT_SEED)}, det_ckpt)
    torch.save({"state_dict": synth.synthetic_hardnet_state_dict(cases.HARDNET_SEED)}, hn_ckpt)
    for i in (1, 2):
        g = synth.synthetic_gray_u8(180, 240, 60 + i, blur=7)
        Image.fromarray(np.stack([g, g // 2 + 60, 255 - g], -1).astype(np.uint8)).save(tmp_path / f"im{i}.png")
    out = tmp_path / "matches.png"
    demo_match.main(["--ckpt_file", str(det_ckpt), "--ckpt_descriptor_file", str(hn_ckpt), "--descriptor_precision", "fp16",
                     str(tmp_path / "im1.png"), str(tmp_path / "im2.png"), str(out)])
    assert np.array(Image.open(out)).shape == (180, 480, 3)
