#!/usr/bin/env python3
"""Wall time of the reference's canonical caller as mirrored here -- pipeline.extract_detections(image_RGB_norm float64 [H,W,3], ...),
/root/reference/balf/utils/train_utils.py:416-454 -- and of demo_match.detect(uint8 image), per call, with where the host time goes."""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd import arch, pipeline                                    # noqa: E402
from balf_amd.demo import demo_match                                   # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402

m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(3))
m = m.eval().to("cuda:0")
res = {}
for (h, w, k) in ((480, 640, 1000), (1080, 1920, 2000)):
    g = synth.synthetic_gray_u8(h, w, 1)
    img = synth.gray_to_rgb_norm(g).astype(np.float64)
    for _ in range(3):
        pipeline.extract_detections(img, m, "cuda:0", nms_size=15, num_points=k, border_size=15)
    t = []
    for _ in range(10):
        t0 = time.perf_counter()
        pts, sm = pipeline.extract_detections(img, m, "cuda:0", nms_size=15, num_points=k, border_size=15)
        t.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(10):
        x = pipeline.pad_batch(img[None])
    t_pad = (time.perf_counter() - t0) * 100
    t0 = time.perf_counter()
    for _ in range(10):
        xd = x.to("cuda:0")
        torch.cuda.synchronize()
    t_h2d = (time.perf_counter() - t0) * 100
    args = SimpleNamespace(**dict(demo_match.DEFAULT_ARGS.__dict__))
    u8 = np.stack([g, g, g], axis=-1)
    for _ in range(3):
        demo_match.detect(args, u8, m, "cuda:0")
    t0 = time.perf_counter()
    for _ in range(10):
        demo_match.detect(args, u8, m, "cuda:0")
    t_det = (time.perf_counter() - t0) * 100
    res[f"{w}x{h}"] = {"extract_detections_ms": sorted(t)[5], "of_which_pad_batch_numpy_ms": t_pad, "h2d_fp32_padded_ms": t_h2d,
                       "demo_match_detect_ms": t_det}
print(json.dumps(res))
