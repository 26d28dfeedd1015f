import sys, torch
sys.path.insert(0, ".")
from balf_amd import arch, ops
from balf_amd.model import get_model
from balf_amd.utils import synth
mode = sys.argv[1]
sd = synth.synthetic_state_dict(1)
if mode == "zero":
    sd = {k: torch.zeros_like(v) if v.is_floating_point() else v for k, v in sd.items()}
    sd["detector_head.norm.running_var"] = torch.ones_like(sd["detector_head.norm.running_var"])
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(sd); m = m.eval().cuda()
x = torch.rand((32, 3, 1088, 1920), device="cuda") if mode != "zero" else torch.zeros((32, 3, 1088, 1920), device="cuda")
for _ in range(2): m(x, want_logits=False)
torch.cuda.synchronize()
ops.profile_begin()
n = 4
for _ in range(n): m(x, want_logits=False)
torch.cuda.synchronize()
prof = ops.profile_end()
tot = sum(v[0] for v in prof.values()) / n
print(mode, "total %.2f ms / 32 img -> %.1f img/s" % (tot, 32 / tot * 1e3), " ".join(f"{k.replace('stage','s').replace('_branch','')}={v[0]/n:.2f}" for k, v in sorted(prof.items())))
