"""Forward time of 32 x 1088x1920 (development aid): used to compare micro-batch sizes -- libraries built with
-DBALF_MB_PIXELS=... through BALF_HIP_LIB (8 -> 16 images per launch, +1 %)."""
import sys, time, torch
sys.path.insert(0, ".")
from balf_amd import arch, ops
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m = m.eval().cuda()
x = torch.rand((32, 3, 1088, 1920), device="cuda")
for _ in range(3): m(x, want_logits=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n): m(x, want_logits=False)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n
print(f"32 x 1088x1920 forward: {t:.2f} ms -> {32 / t * 1e3:.1f} img/s")
