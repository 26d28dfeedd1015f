"""The reference's matching demo with every stage on the GPU.

Same function names and arguments as /root/reference/demo/demo_match.py:21-112 (``detect``, ``extract_features``,
``extract_matches``); ``args`` is any object with the attributes of ``config.parse_test_config``
(/root/reference/balf/configs/config.py:42-59): border_size, nms_size, num_features, s_mult, order_coord,
heatmap_confidence_threshold, sub_pixel, patch_size.  Image decoding (PIL) and drawing (cv2) stay with the caller.

detector -> balf_forward_u8, keypoints -> balf_greedy_nms (+ soft-argmax), patches -> balf_extract_patches,
descriptors -> balf_hardnet_forward, matches -> balf_match_smnn.  The only host round trip is the final result.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .. import arch, ops

DEFAULT_ARGS = SimpleNamespace(border_size=15, nms_size=15, num_features=2048, s_mult=60, order_coord="xysr",
                               heatmap_confidence_threshold=0.001, sub_pixel=True, patch_size=4)
_MAX_POINTS = 16384


def _detect_gpu(args, im: np.ndarray, detector, device) -> torch.Tensor:
    """[n,3] (x, y, 1) on the GPU, strongest first, at most args.num_features rows."""
    if args.order_coord not in ("xysr", "yxsr"):
        raise ValueError(args.order_coord)
    img = torch.as_tensor(np.ascontiguousarray(im), device=device)
    if img.dtype != torch.uint8:
        raise ValueError("expected a uint8 image, as load_im returns it")
    h, w = img.shape[:2]
    hp, wp, top, left = arch.padded_hw(h, w)
    k = min(_MAX_POINTS, h * w)

    def run():
        with torch.inference_mode():
            prob = detector.forward_u8(img.unsqueeze(0), want_logits=False)["prob"]
            return ops.greedy_nms(prob, top, left, h, w, args.border_size, args.heatmap_confidence_threshold, args.nms_size, k,
                                  args.patch_size if args.sub_pixel else 0)
    idx, score, xy, count, total = run()
    n = min(int(count[0].item()), int(args.num_features))       # (a device-to-host read: the stream has passed the forward)
    # one image per call: the split-f16 status block is final here, for free -- a flagged call is repeated on the fp32 kernels
    if getattr(detector, "fp16_guard_check", None) is not None and detector.fp16_guard_check(synchronize=False):
        idx, score, xy, count, total = run()
        n = min(int(count[0].item()), int(args.num_features))
    if args.sub_pixel:
        pts = xy[0, :n]
    else:
        ii = idx[0, :n].long()
        pts = torch.stack([(ii % w).float(), (ii // w).float()], dim=1)
    if args.order_coord == "yxsr":        # rows (y, x, 1): whatever consumes them downstream sees them swapped, as in
        pts = pts.flip(1)                 # the reference (test_utils.py:121-124 feeding demo_match.py:57,62-70)
    return torch.cat([pts, torch.ones((n, 1), device=device)], dim=1)


def detect_and_describe_batch(args, images_u8: torch.Tensor, detector, descriptor, gray_u8: torch.Tensor = None):
    """The feature half of the demo for a whole batch in one pass, nothing on the host: uint8 images on the GPU
    (gray ``[B,H,W]`` or RGB ``[B,H,W,3]`` with their gray versions in ``gray_u8``) -> (xy [B,K,2] keypoints,
    strongest first; descriptors [B,K,128]; count [B] valid rows per image), K = args.num_features.  Per image this
    is what :func:`extract_features` returns (demo_match.py:59-95)."""
    if args.order_coord not in ("xysr", "yxsr"):
        raise ValueError(args.order_coord)
    if gray_u8 is None:       # RGB without its gray version: PIL's convert('L') arithmetic on the GPU (demo_match.py:15-17)
        gray_u8 = images_u8 if images_u8.dim() == 3 else ops.rgb_to_gray_u8(images_u8.contiguous())
    b, h, w = images_u8.shape[:3]
    _, _, top, left = arch.padded_hw(h, w)
    k = min(int(args.num_features), h * w)
    # The selection kernel keeps the raster-first K among the points that reach the K-th score (the window-max path's
    # rule); the demo sorts by score and truncates (demo_match.py:54-55), so a point strictly above the K-th score must
    # never lose to a tie at it: select as many as the kernel returns, truncate to num_features afterwards.
    ksel = min(_MAX_POINTS, h * w)
    with torch.inference_mode():
        prob = detector.forward_u8(images_u8, want_logits=False)["prob"]
        idx, score, xy, count, total = ops.greedy_nms(prob, top, left, h, w, args.border_size,
                                                      args.heatmap_confidence_threshold, args.nms_size, ksel,
                                                      args.patch_size if args.sub_pixel else 0)
        idx, count = idx[:, :k].contiguous(), count.clamp(max=k)
        if not args.sub_pixel:
            ii = idx.long().clamp_(min=0)
            xy = torch.stack([(ii % w).float(), (ii // w).float()], dim=2)
        else:
            xy = xy[:, :k].contiguous()
        if args.order_coord == "yxsr":
            xy = xy.flip(2)
        patches = ops.extract_patches_batch(gray_u8, xy, count, float(args.s_mult))
        descs = descriptor.forward_slots(patches, count)          # unused slots: zero rows, no compute
    return xy, descs, count


def detect(args, im, detector, device):
    """demo_match.py:21-57: image [H,W,3] uint8 -> keypoints [n,3] = (x, y, 1), strongest first."""
    pts = _detect_gpu(args, im, detector, device)
    if pts.shape[0] == 0:
        # the reference returns a PAIR here (demo_match.py:51-52) although every other path returns one array: mirrored,
        # pinned by tests/golden/callers.npz (d_conf_high)
        return np.zeros([0, 3]), np.zeros([0, 1])
    return pts.double().cpu().numpy()


def extract_features(args, im_rgb, im_gray, detector, descriptor, device):
    """demo_match.py:59-95: -> (keypoints [n,2] float64 NumPy, descriptors [n,128] float32 NumPy)."""
    kpts = _detect_gpu(args, im_rgb, detector, device)
    gray = torch.as_tensor(np.ascontiguousarray(im_gray), device=device)
    with torch.inference_mode():
        patches = ops.extract_patches(gray, kpts[:, :2], float(args.s_mult))
        descs = descriptor(patches)
    return kpts[:, :2].double().cpu().numpy(), descs.cpu().numpy()


def extract_matches(args, im_rgb1, im_gray1, im_rgb2, im_gray2, detector, descriptor, device):
    """demo_match.py:97-112: -> (points1 [m,2], points2 [m,2]) of the mutual ratio-test matches (th 0.99)."""
    kpts1, desc1 = extract_features(args, im_rgb1, im_gray1, detector, descriptor, device)
    kpts2, desc2 = extract_features(args, im_rgb2, im_gray2, detector, descriptor, device)
    with torch.inference_mode():
        _, match_ids = ops.match_smnn(torch.from_numpy(desc1).to(device), torch.from_numpy(desc2).to(device), 0.99)
    match_ids = match_ids.cpu().numpy()
    return kpts1[match_ids[:, 0], :2], kpts2[match_ids[:, 1], :2]


def load_im(im_path):
    """demo_match.py:13-19: (RGB uint8 [H,W,3], gray uint8 [H,W]) via PIL."""
    from PIL import Image
    im_rgb = Image.open(im_path).convert('RGB')
    return np.array(im_rgb), np.array(im_rgb.convert('L'))


def draw_matches(im1, kpts1, im2, kpts2):
    """Side-by-side canvas with one line per match (the reference draws with cv2.drawMatches, demo_match.py:114-118;
    cv2 is not a dependency here, PIL is)."""
    from PIL import Image, ImageDraw
    h = max(im1.shape[0], im2.shape[0])
    canvas = Image.new('RGB', (im1.shape[1] + im2.shape[1], h))
    canvas.paste(Image.fromarray(im1), (0, 0))
    canvas.paste(Image.fromarray(im2), (im1.shape[1], 0))
    d = ImageDraw.Draw(canvas)
    for (x1, y1), (x2, y2) in zip(kpts1, kpts2):
        d.line([(float(x1), float(y1)), (float(x2) + im1.shape[1], float(y2))], fill=(0, 255, 0), width=1)
    return np.array(canvas)


def main(argv=None):
    """The reference demo's ``__main__`` (demo_match.py:120-147) as a command:
    python -m balf_amd.demo.demo_match --ckpt_file balf.pth --ckpt_descriptor_file HardNet++.pth im1.jpg im2.jpg out.png"""
    import argparse
    from ..model import get_model
    from ..third_party.hardnet.hardnet_pytorch import HardNet
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument('--ckpt_file', required=True)
    ap.add_argument('--ckpt_descriptor_file', required=True)
    ap.add_argument('--descriptor_precision', default='fp16-split', choices=['fp16-split', 'fp16'])
    ap.add_argument('--detector_precision', default='fp32', choices=['fp32', 'fp16'])
    ap.add_argument('image1'); ap.add_argument('image2'); ap.add_argument('output')
    a = ap.parse_args(argv)
    device = torch.device('cuda')
    detector = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    get_model.load_test_pretrained_model(model=detector, filename=a.ckpt_file)
    detector.precision = a.detector_precision
    detector = detector.eval().to(device)
    descriptor = HardNet()
    descriptor.load_state_dict(torch.load(a.ckpt_descriptor_file, weights_only=True)['state_dict'])
    descriptor.precision = a.descriptor_precision
    descriptor = descriptor.eval().to(device)
    im_rgb1, im_gray1 = load_im(a.image1)
    im_rgb2, im_gray2 = load_im(a.image2)
    m1, m2 = extract_matches(DEFAULT_ARGS, im_rgb1, im_gray1, im_rgb2, im_gray2, detector, descriptor, device)
    from PIL import Image
    Image.fromarray(draw_matches(im_rgb1, m1, im_rgb2, m2)).save(a.output)
    print(f"{len(m1)} matches -> {a.output}")


if __name__ == "__main__":
    main()
