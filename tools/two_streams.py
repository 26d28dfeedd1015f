"""Experiment (development aid): the 32-image step as two half-batches on two streams vs one stream."""
import sys, time
import torch
sys.path.insert(0, ".")
from balf_amd import arch, ops, pipeline
from balf_amd.model import get_model
from balf_amd.utils import synth
import numpy as np
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m = m.eval().cuda()
h, w, k, b = 1080, 1920, 2000, 32
hp, wp, top, left = arch.padded_hw(h, w)
gray = np.stack([synth.synthetic_gray_u8(h, w, i, blur=5 if i % 2 == 0 else 1) for i in range(8)]).repeat(4, axis=0)
g = torch.from_numpy(gray).cuda().float().div_(255.0)
x = torch.zeros((b, 3, hp, wp), device="cuda"); x[:, :, top:top + h, left:left + w] = g[:, None]
def one():
    return pipeline.detect_batch(m, x, h, w, 15, 15, k, precomputed_offsets=(top, left))
def split(ns):
    outs = []
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams[:ns]):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            lo, hi = i * b // ns, (i + 1) * b // ns
            outs.append(pipeline.detect_batch(m, x[lo:hi], h, w, 15, 15, k, precomputed_offsets=(top, left)))
    for s in streams[:ns]:
        cur.wait_stream(s)
    return outs
streams = [torch.cuda.Stream() for _ in range(4)]
for name, fn in (("1 stream", one), ("2 streams", lambda: split(2)), ("4 streams", lambda: split(4)), ("1 stream", one), ("2 streams", lambda: split(2))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): o = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.2f} ms per 32-image step -> {b/dt:.1f} img/s")
