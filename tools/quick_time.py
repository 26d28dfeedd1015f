"""Ad-hoc timing of the forward + NMS at a few sizes (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
from balf_amd import arch, ops
from balf_amd.model import get_model
from balf_amd.utils import synth

m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
m.load_state_dict(synth.synthetic_state_dict(1))
m.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
m = m.eval().cuda()
for (b, h, w, k) in [(32, 512, 640, 1000), (8, 1088, 1920, 2000), (32, 1088, 1920, 2000)]:
    x = torch.rand((b, 3, h, w), device="cuda")
    for _ in range(2):
        out = m(x, want_logits=False)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 3
    e0.record()
    for _ in range(n):
        out = m(x, want_logits=False)
    e1.record()
    for _ in range(n):
        r = ops.nms_topk(out["prob"], 4, 0, h - 8, w, 15, 15, k)
    e2.record()
    torch.cuda.synchronize()
    tf, tn = e0.elapsed_time(e1) / n, e1.elapsed_time(e2) / n
    gflop = b * h * w * arch.FLOP_PER_PADDED_PIXEL / 1e9
    print(f"B={b} {h}x{w}: forward {tf:.2f} ms ({b / tf * 1e3:.1f} img/s, {gflop / tf:.1f} TFLOP/s)  nms+topk {tn:.3f} ms "
          f"({b / tn * 1e3:.0f} img/s)")
