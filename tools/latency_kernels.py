import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from balf_amd import arch, ops, pipeline
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(3)); m = m.eval().cuda()
for (h, w, k) in ((480, 640, 1000), (1080, 1920, 2000)):
    img = torch.from_numpy(synth.synthetic_gray_u8(h, w, 0)[None]).cuda()
    for _ in range(5): pipeline.detect_batch_u8(m, img, 15, 15, k)
    torch.cuda.synchronize()
    ops.profile_begin()
    for _ in range(20): pipeline.detect_batch_u8(m, img, 15, 15, k)
    torch.cuda.synchronize()
    prof = ops.profile_end()
    tot = sum(v[0] for v in prof.values()) / 20
    print(h, w, "total %.3f ms" % tot, " ".join(f"{n.replace('stage','s').replace('_branch','')}={v[0]/20*1e3:.0f}us" for n, v in sorted(prof.items())))
