"""Determinism soak of the forward: N repetitions at full load, every result must be bit-identical to the first.
Usage: python tools/soak.py [reps] [batch] [H] [W] [fp16|fp32]"""
import sys, torch
sys.path.insert(0, ".")
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1088
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(7)); m.precision = sys.argv[5] if len(sys.argv) > 5 else "fp16"; m = m.eval().cuda()
x = torch.rand((b, 3, h, w), device="cuda")
with torch.inference_mode():
    ref = m(x)
    bad = 0
    for i in range(reps):
        if i and i % 5000 == 0:
            print(f"  ... {i} repetitions, {bad} mismatches so far", flush=True)
        o = m(x)
        if not (torch.equal(o["prob"], ref["prob"]) and torch.equal(o["logits"], ref["logits"])):
            bad += 1
            d = (o["prob"] != ref["prob"])
            w = d.nonzero()
            print("mismatch at repetition", i, int(d.sum()), "differing score-map values; max abs diff %.3e" % float((o["prob"] - ref["prob"]).abs().max()),
                  "first at (image, y, x)", w[0].tolist() if len(w) else None, "last", w[-1].tolist() if len(w) else None, flush=True)
print(f"{m.effective_precision}: {reps} repetitions of {b}x{h}x{w}: {bad} mismatches, finite={bool(torch.isfinite(ref['prob']).all())}")
