#!/usr/bin/env python3
"""Throughput of the demo's feature half for a batch: uint8 images -> detector -> greedy NMS + sub-pixel -> 32x32
patches -> HardNet descriptors (demo_match.extract_features per image), plus one mutual-NN match of two images.
Usage: python tools/bench_demo.py [batch] [H] [W] [steps] [fp16-split|fp16 (descriptor operands)]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd import arch, ops                                         # noqa: E402
from balf_amd.demo import demo_match                                   # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet       # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    desc_precision = sys.argv[5] if len(sys.argv) > 5 else "fp16-split"
    dev = torch.device("cuda:0")
    det = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    det.load_state_dict(synth.synthetic_state_dict(20240))
    det.precision = "fp16"
    det = det.eval().to(dev)
    hn = HardNet()
    hn.load_state_dict(synth.synthetic_hardnet_state_dict(515))
    hn.precision = desc_precision
    hn = hn.eval().to(dev)
    imgs = torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, i, blur=5 if i % 2 == 0 else 3) for i in range(b)])).to(dev)
    args = demo_match.DEFAULT_ARGS
    for _ in range(2):
        xy, desc, count = demo_match.detect_and_describe_batch(args, imgs, det, hn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        xy, desc, count = demo_match.detect_and_describe_batch(args, imgs, det, hn)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.profile_begin()
    demo_match.detect_and_describe_batch(args, imgs, det, hn)
    torch.cuda.synchronize()
    prof = ops.profile_end()
    groups = {"detector": 0.0, "greedy_nms": 0.0, "patches": 0.0, "hardnet": 0.0}
    for name, (ms, _) in prof.items():
        key = ("detector" if name.startswith("stage") else "hardnet" if name.startswith("hardnet")
               else "patches" if name.startswith("patch") else "greedy_nms")       # greedy_*, topk_select
        groups[key] += ms
    n0, n1 = int(count[0]), int(count[1])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        _, ids = ops.match_smnn(desc[0, :n0], desc[1, :n1], 0.99)
    torch.cuda.synchronize()
    t_match = (time.perf_counter() - t1) / 20
    # all consecutive image pairs of the batch in one batched call (16 pairs at batch 32)
    half = b // 2
    if half:
        d1, d2 = desc[0::2][:half].contiguous(), desc[1::2][:half].contiguous()
        c1, c2 = count[0::2][:half].contiguous(), count[1::2][:half].contiguous()
        ops.match_smnn_batch(d1, c1, d2, c2, 0.99)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(10):
            mdist, midx, mcount = ops.match_smnn_batch(d1, c1, d2, c2, 0.99)
        torch.cuda.synchronize()
        t_batch = (time.perf_counter() - t2) / 10
    kp = float(count.float().mean())
    print(json.dumps({"metric": "demo feature extraction (detect + describe), images/s", "value": b / dt, "unit": "images/s",
                      "batch": b, "image": f"{w}x{h} gray uint8", "ms_per_batch": dt * 1e3, "keypoints_per_image": kp,
                      "descriptors_per_s": b * kp / dt, "device_ms": groups,
                      "match_smnn_ms": t_match * 1e3, "match_pairs": [n0, n1], "matches": int(ids.shape[0]),
                      "match_smnn_batch_ms": (t_batch * 1e3 if half else None), "match_batch_pairs": half,
                      "dtype": f"split-f16 MFMA detector, descriptor {desc_precision}, fp32 everything else"}))


if __name__ == "__main__":
    main()
