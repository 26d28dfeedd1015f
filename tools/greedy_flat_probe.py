#!/usr/bin/env python3
"""balf_greedy_nms on score maps of FLAT images (black / white / half black): the detector's softmax is near-uniform there
(1/65 = 0.0154 > the demo's conf_thresh 0.015), every pixel is a candidate and equal values repeat from cell to cell -- the
wavefront case of the parallel form.  python tools/greedy_flat_probe.py [batch] -> ms per call for each kind of image."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd import arch, ops                                         # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h, w = 1080, 1920
dev = torch.device("cuda:0")
det = get_model.load_model(arch.DEFAULT_MODEL_CFG)
det.load_state_dict(synth.synthetic_state_dict(20240))
det = det.eval().to(dev)
hp, wp, top, left = arch.padded_hw(h, w)
res = {}
for name in ("noise", "black", "white", "half"):
    if name == "noise":
        img = np.stack([synth.synthetic_gray_u8(h, w, i, blur=3) for i in range(b)])
    else:
        img = np.zeros((b, h, w), np.uint8)
        if name == "white":
            img[:] = 255
        if name == "half":
            img[:, :, w // 2:] = np.stack([synth.synthetic_gray_u8(h, w, i, blur=3) for i in range(b)])[:, :, w // 2:]
    prob = det.forward_u8(torch.from_numpy(img).to(dev), want_logits=False)["prob"]
    args = (prob, top, left, h, w, 15, 0.015, 15, 2048, 5)
    out = ops.greedy_nms(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out = ops.greedy_nms(*args)
    torch.cuda.synchronize()
    res[name] = {"ms_per_call": (time.perf_counter() - t0) / n * 1e3, "kept_per_image": float(out[4].float().mean()),
                 "candidate_fraction": float((prob[:, top:top + h, left:left + w] >= 0.015).float().mean())}
print(json.dumps({"batch": b, "image": f"{w}x{h}", **res}))
