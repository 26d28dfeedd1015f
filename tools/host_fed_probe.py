#!/usr/bin/env python3
"""Where the host-fed leg of bench.py loses against the resident one: the same 32 x 1080p batch through
(a) detect_batch on the resident padded fp32 batch, (b) detect_batch_u8 on resident uint8, (c) the same with logits,
(d) + the slab packing and the D2H, (e) + the double-buffered H2D (bench.host_fed_run).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                           # noqa: E402

bench._import_product()
from balf_amd import arch, ops, pipeline                               # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    b, h, w, k = 32, 1080, 1920, 2000
    dev = torch.device("cuda:0")
    model = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    model.load_state_dict(synth.synthetic_state_dict(20240))
    model = model.eval().to(dev)
    gray = bench.synthetic_batch(h, w, 0, b)
    hp, wp, top, left = arch.padded_hw(h, w)
    g = torch.from_numpy(gray).to(dev)
    x = torch.zeros((b, 3, hp, wp), dtype=torch.float32, device=dev)
    x[:, :, top:top + h, left:left + w] = (g.float() / 255.0)[:, None]
    res = {}
    res["a_f32_resident"] = timed(lambda: pipeline.detect_batch(model, x, h, w, 15, 15, k, precomputed_offsets=(top, left)))
    res["b_u8_resident"] = timed(lambda: pipeline.detect_batch_u8(model, g, 15, 15, k))

    def c():
        out = model.forward_u8(g, want_logits=True)
        return ops.nms_topk(out["prob"], top, left, h, w, 15, 15, k)
    res["c_u8_logits"] = timed(c)
    slab = torch.empty((b, 2 * k + 1), dtype=torch.int32, device=dev)
    host = torch.empty((b, 2 * k + 1), dtype=torch.int32).pin_memory()

    def d():
        idx, score, count = c()
        slab[:, :k] = idx
        slab[:, k:2 * k] = score.view(torch.int32)
        slab[:, 2 * k] = count
        host.copy_(slab, non_blocking=True)
    res["d_plus_slab_d2h"] = timed(d)
    hf = bench.host_fed_run(model, dev, gray, k, 12, lambda: pipeline.detect_batch(model, x, h, w, 15, 15, k, precomputed_offsets=(top, left)))
    res["e_ratio"] = hf["ratio_to_resident"]
    res["e_host_fed"] = hf["ms_per_step"]
    # the H2D alone, and beside the forward
    pin = torch.from_numpy(gray).pin_memory()
    res["h2d_alone_ms"] = timed(lambda: g.copy_(pin, non_blocking=True))
    print(json.dumps({k_: round(v, 3) for k_, v in res.items()}))


if __name__ == "__main__":
    main()
