"""What the per-kernel hipEvent pairs of balf_profile_begin/end cost the step (development aid)."""
import sys, time, torch
sys.path.insert(0, ".")
from balf_amd import arch, ops, pipeline
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m.precision = "fp16"; m = m.eval().cuda()
x = torch.rand((32, 3, 1088, 1920), device="cuda")
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        pipeline.detect_batch(m, x, 1080, 1920, 15, 15, 2000, precomputed_offsets=(4, 0))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(2)
for rep in range(2):
    a = run(5)
    ops.profile_begin(); b = run(5); ops.profile_end()
    print(f"ms per step: plain {a:.3f}, with per-kernel events {b:.3f}")
