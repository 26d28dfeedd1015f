#!/usr/bin/env python3
"""Images far beyond 1080p (the kernels address an image with 32-bit byte offsets; the documented limit is 2^25 padded pixels):
one image per size through BOTH kernel families -- split-f16 and exact fp32, which share no stage kernel -- score maps compared;
window NMS + top-K and greedy NMS on the f16 map against the C / NumPy oracles on the same map (identical-input parity).
No CPU forward: the oracle's activations at these sizes would not fit a pool box's share of memory.  One line per size."""
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
from oracle import c_oracle, oracle as O                               # noqa: E402

dev = torch.device("cuda:0")
m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(20240))
m = m.eval().to(dev)
sizes = [(2160, 3840), (4096, 4096), (5760, 5760)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (h, w) in sizes:
    hp, wp, top, left = arch.padded_hw(h, w)
    g = np.random.default_rng(h).integers(0, 256, (h, w), dtype=np.uint8)
    g[h // 3: h // 2, w // 4: w // 2] = 30                       # a flat region as well
    img = torch.from_numpy(g[None]).to(dev)
    out = {"image": f"{w}x{h}", "padded_pixels": hp * wp, "limit": 1 << 25}
    t0 = time.perf_counter()
    m.precision = "fp16"
    p16 = m.forward_u8(img, want_logits=True)
    torch.cuda.synchronize()
    out["f16_ms"] = (time.perf_counter() - t0) * 1e3
    m.precision = "fp32"
    p32 = m.forward_u8(img, want_logits=True)
    torch.cuda.synchronize()
    m.precision = "fp16"
    out["prob_max_abs_f16_vs_f32"] = float((p16["prob"] - p32["prob"]).abs().max())
    out["logits_max_abs_f16_vs_f32"] = float((p16["logits"] - p32["logits"]).abs().max())
    out["finite"] = bool(torch.isfinite(p16["prob"]).all())
    k = 10000
    idx, score, count = ops.nms_topk(p16["prob"], top, left, h, w, 15, 15, k)
    pm = np.ascontiguousarray(p16["prob"][0, top:top + h, left:left + w].cpu().numpy())
    ri, rs, _ = c_oracle.nms_topk(pm, 15, 15, k)
    ri, rs = O.canonical_order(ri.astype(np.int64), rs)
    n = int(count[0])
    out["nms_topk_identical_input"] = bool(n == ri.size and np.array_equal(idx[0, :n].cpu().numpy(), ri.astype(np.int32)) and
                                           np.array_equal(score[0, :n].cpu().numpy().view(np.uint32), rs.view(np.uint32)))
    gi, gs, _, gc, gt = ops.greedy_nms(p16["prob"], top, left, h, w, 15, 0.015, 15, 16384, 0)
    oi, osc = O.greedy_nms(O.remove_borders(pm, 15), 0.015, 15)
    n = int(gc[0])
    out["greedy_total"] = int(gt[0])
    out["greedy_identical_input"] = bool(int(gt[0]) == len(oi) and np.array_equal(gi[0, :n].cpu().numpy(), np.asarray(oi[:n], np.int32)))
    print(json.dumps(out), flush=True)
    del p16, p32
    torch.cuda.empty_cache()
