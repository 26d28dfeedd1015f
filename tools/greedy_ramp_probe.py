import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from balf_amd import ops
h, w = 1080, 1920
yy, xx = np.mgrid[0:h, 0:w]
res = {}
for name, m in (("ramp_x", (0.1 + 0.8 * xx / w + 1e-5 * yy / h).astype(np.float32)), ("plateau", np.full((h, w), 0.5, np.float32)),
                ("noise", np.random.default_rng(0).random((h, w), dtype=np.float32))):
    t = torch.from_numpy(np.stack([m, m])).cuda()
    out = ops.greedy_nms(t, 0, 0, h, w, 15, 0.015, 15, 2048, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ops.greedy_nms(t, 0, 0, h, w, 15, 0.015, 15, 2048, 0)
    torch.cuda.synchronize()
    res[name] = ((time.perf_counter() - t0) * 1e3, int(out[4][0]))
print(res)
