"""Determinism soak of the whole pipeline of the final build: detector forward on raw uint8 images -> window NMS + top-K -> greedy NMS
+ sub-pixel step, N repetitions back to back, every output bit-identical to the first; then the same pipeline as a hipGraph
(pipeline.GraphedDetector + a captured greedy NMS) replayed N times.  Usage: python tools/soak_pipeline.py [reps] [batch] [H] [W]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from balf_amd import arch, ops, pipeline          # noqa: E402
from balf_amd.model import get_model              # noqa: E402
from balf_amd.utils import synth                  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(7))
m = m.eval().cuda()
_, _, top, left = arch.padded_hw(h, w)
x = torch.from_numpy(np.stack([synth.synthetic_gray_u8(h, w, i, blur=3) for i in range(b)])).cuda()


def run():
    idx, score, count, prob = pipeline.detect_batch_u8(m, x, 15, 15, 2000)
    return (idx, score, count, prob) + tuple(ops.greedy_nms(prob, top, left, h, w, 15, 0.015, 15, 2048, 5))


ref = [t.clone() for t in run()]
bad = 0
t0 = time.perf_counter()
for i in range(reps):
    out = run()
    if not all(torch.equal(a, r) for a, r in zip(out, ref)):
        bad += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"eager: {reps} repetitions of {b} x {w}x{h} (forward_u8 + nms_topk + greedy_nms + sub-pixel): {bad} mismatches, {b * reps / dt:.0f} img/s incl. the comparisons")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gout = run()
bad = 0
for i in range(reps):
    g.replay()
    if not all(torch.equal(a, r) for a, r in zip(gout, ref)):
        bad += 1
torch.cuda.synchronize()
print(f"graph: {reps} replays of the captured pipeline: {bad} mismatches")
