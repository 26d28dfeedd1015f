import sys, os, torch
sys.path.insert(0, ".")
from balf_amd import ops
torch.manual_seed(0)
prob = torch.rand((32, 1088, 1920), device="cuda") ** 8
for _ in range(3): r = ops.nms_topk(prob, 4, 0, 1080, 1920, 15, 15, 2000)
torch.cuda.synchronize()
ops.profile_begin()
for _ in range(10): r = ops.nms_topk(prob, 4, 0, 1080, 1920, 15, 15, 2000)
torch.cuda.synchronize()
p = ops.profile_end()
print(os.environ.get("BALF_NMS_NO_VEC"), {k: round(v[0] / 10, 4) for k, v in p.items()}, "checksum", int(r[0].long().sum()), float(r[1].double().sum()), int(r[2].sum()))
