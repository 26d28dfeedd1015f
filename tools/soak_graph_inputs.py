"""A captured detection pipeline replayed on CHANGING inputs (5 different batches, cycled), every replay compared with the eager
result of that batch: stale state inside the graph (the hipMemsetAsync trap of round 6) shows as a mismatch.
Usage: python tools/soak_graph_inputs.py [replays] [batch] [H] [W]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from balf_amd import arch, ops, pipeline          # noqa: E402
from balf_amd.model import get_model              # noqa: E402
from balf_amd.utils import synth                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h = int(sys.argv[3]) if len(sys.argv) > 3 else 480
w = int(sys.argv[4]) if len(sys.argv) > 4 else 640
m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(7))
m = m.eval().cuda()
_, _, top, left = arch.padded_hw(h, w)
batches = []
for j in range(5):
    imgs = np.stack([synth.synthetic_gray_u8(h, w, 31 * j + i, blur=(1, 3, 5)[(i + j) % 3]) for i in range(b)])
    if j == 3:
        imgs[:, : h // 2] = 0                      # half black: flat regions, ties
    if j == 4:
        imgs[:] = 255 * (np.indices((h, w)).sum(0) % 2).astype(np.uint8)    # one-pixel checkerboard
    batches.append(torch.from_numpy(imgs).cuda())


def run(x):
    idx, score, count, prob = pipeline.detect_batch_u8(m, x, 15, 15, 1000)
    return (idx, score, count, prob) + tuple(ops.greedy_nms(prob, top, left, h, w, 15, 0.015, 15, 2048, 5))


want = [[t.clone() for t in run(x)] for x in batches]
static = batches[0].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run(static)
bad = 0
for i in range(reps):
    j = (i * 3 + 1) % 5
    static.copy_(batches[j])
    g.replay()
    if not all(torch.equal(a, r) for a, r in zip(out, want[j])):
        bad += 1
torch.cuda.synchronize()
print(f"graph: {reps} replays of {b} x {w}x{h} over 5 different batches (incl. half-black and checkerboard images): {bad} mismatches")
