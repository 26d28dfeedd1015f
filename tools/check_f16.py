import sys, numpy as np, torch
sys.path.insert(0, ".")
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
def mk(prec):
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(20240)); m.precision = prec
    return m.eval().cuda()
m32, m16 = mk("fp32"), mk("fp16")
for (b, h, w) in [(1, 64, 64), (2, 128, 192), (1, 512, 640), (4, 512, 640), (2, 1088, 1920)]:
    x = torch.rand((b, 3, h, w), device="cuda")
    a = m32(x)["prob"]; c = m16(x)["prob"]; d = m16(x)["prob"]
    err = (a - c).abs()
    print(b, h, w, "max err f16 vs f32: %.3e" % err.max().item(), "run-to-run equal:", torch.equal(c, d),
          "bad px:", int((err > 1e-4).sum()), "nan:", int(torch.isnan(c).sum()))
    if err.max() > 1e-4:
        bad = (err > 1e-4).nonzero()
        print("  first bad:", bad[:5].tolist(), " per-image bad counts:", [(int((err[i] > 1e-4).sum())) for i in range(b)])
