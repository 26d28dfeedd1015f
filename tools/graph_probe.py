#!/usr/bin/env python3
"""hipGraph capture / replay of pieces of the detection pipeline (development aid for tests/test_graph_capture_gpu.py):
python tools/graph_probe.py <forward|nms|greedy|all> <replays> <eager between replays: 0|1|2>  (2 = an unrelated eager allocation only)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd import arch, ops, pipeline            # noqa: E402
from balf_amd.model import get_model                # noqa: E402
from balf_amd.utils import synth                    # noqa: E402

which, n_rep, eager = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
DEV = "cuda:0"
m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(3))
m = m.eval().to(DEV)
h, w, k = 480, 640, 1000
_, _, top, left = arch.padded_hw(h, w)
imgs = [torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, 10 * j + i) for i in range(2)])).to(DEV) for j in range(3)]
static = imgs[0].clone()
prob0 = m.forward_u8(static, want_logits=False)["prob"]


def run(x):
    if which == "forward":
        return (m.forward_u8(x, want_logits=False)["prob"],)
    if which == "nms":
        return ops.nms_topk(prob0, top, left, h, w, 15, 15, k)
    if which == "greedy":
        return ops.greedy_nms(prob0, top, left, h, w, 15, 0.015, 15, 1024, 5)
    idx, score, count, prob = pipeline.detect_batch_u8(m, x, 15, 15, k)
    return (idx, score, count, prob) + tuple(ops.greedy_nms(prob, top, left, h, w, 15, 0.015, 15, 1024, 5))


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        run(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = run(static)
print(which, "captured", flush=True)
for it in range(n_rep):
    x = imgs[(it + 1) % 3]
    static.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    print(which, "replay", it, "ok", flush=True)
    if eager == 1:
        want = run(x)
        torch.cuda.synchronize()
        print(which, "eager", it, "equal:", all(bool(torch.equal(a, b)) for a, b in zip(outs, want) if a is not None), flush=True)
    elif eager == 2:
        junk = torch.empty(600 << 20, dtype=torch.uint8, device=DEV).fill_(1)
        del junk
        torch.cuda.synchronize()
print("DONE", which, n_rep, eager, flush=True)
