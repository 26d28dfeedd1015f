#!/usr/bin/env python3
"""Device time of balf_greedy_nms (demo post-processing, row f1) on the detector's own score maps:
python tools/bench_greedy.py [batch] [H] [W] -> one JSON line (per-kernel slots, wall per call, rounds' alive statistics)."""
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


def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    dev = torch.device("cuda:0")
    det = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    det.load_state_dict(synth.synthetic_state_dict(20240))
    det.precision = "fp16"
    det = det.eval().to(dev)
    imgs = torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, i, blur=5 if i % 2 == 0 else 3) for i in range(b)])).to(dev)
    prob = det.forward_u8(imgs, want_logits=False)["prob"]
    hp, wp = prob.shape[1:]
    top, left = (hp - (h + (h & 1))) // 2, (wp - (w + (w & 1))) // 2
    args = (prob, top, left, h, w, 15, 0.015, 15, 2048, 5)
    for _ in range(3):
        out = ops.greedy_nms(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        out = ops.greedy_nms(*args)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ops.profile_begin()
    ops.greedy_nms(*args)
    torch.cuda.synchronize()
    prof = ops.profile_end()
    cand = float((prob[:, top:top + h, left:left + w] >= 0.015).float().mean())
    print(json.dumps({"metric": "balf_greedy_nms ms per batch", "batch": b, "image": f"{w}x{h}", "ms_wall_per_call": wall * 1e3,
                      "device_ms_by_slot": {k: round(v[0], 4) for k, v in prof.items()},
                      "launches_by_slot": {k: v[1] for k, v in prof.items()},
                      "candidate_fraction": cand, "kept_per_image": float(out[4].float().mean()),
                      "rounds_env": os.environ.get("BALF_GREEDY_ROUNDS", "default 12")}))


if __name__ == "__main__":
    main()
