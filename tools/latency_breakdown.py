"""Where a single-image call spends its host time (development aid): python tools/latency_breakdown.py [H W K]"""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
from balf_amd import arch, pipeline
from balf_amd.model import get_model
from balf_amd.utils import synth
h, w, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (480, 640, 1000)
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m = m.eval().cuda()
img = torch.from_numpy(synth.synthetic_gray_u8(h, w, 0)[None]).cuda()
for _ in range(5):
    pipeline.detect_batch_u8(m, img, 15, 15, k)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    pipeline.detect_batch_u8(m, img, 15, 15, k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host time per call without sync (launch side only): {(t1 - t0) / n * 1e6:.1f} us; drained after {(t2 - t1) * 1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(n):
    pipeline.detect_batch_u8(m, img, 15, 15, k)
    torch.cuda.synchronize()
print(f"wall per call with sync: {(time.perf_counter() - t0) / n * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    pipeline.detect_batch_u8(m, img, 15, 15, k)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
