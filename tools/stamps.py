"""Per-phase cycle breakdown of the f16 stage kernels (diagnostic build libbalf_hip_stamps.so, -DBALF_STAMPS=1)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BALF_HIP_LIB"] = os.path.abspath("balf_amd/libbalf_hip_stamps.so")
from balf_amd import arch, _lib
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m.precision = "fp16"
m = m.eval().cuda()
x = torch.rand((8, 3, 1088, 1920), device="cuda")
raw = C.CDLL(os.environ["BALF_HIP_LIB"])
sums = (C.c_ulonglong * (16 * 24))(); cnt = (C.c_ulonglong * 16)()
m(x, want_logits=False); torch.cuda.synchronize()
raw.balf_debug_stamps(sums, cnt, 1)
for _ in range(2): m(x, want_logits=False)
torch.cuda.synchronize()
raw.balf_debug_stamps(sums, cnt, 0)
names = ["", "prologue", "x0+LN+slot", "dense1+GELU", "LN+slot", "d1a+GELU", "d1b+GELU+LN", "bT write+bar", "mix", "gate+d2+res",
         "u' store | q2", "x0re+R+LN+slot", "conv1+lrelu+slot", "conv2+T+sums", "partials"]
for kid in range(8):
    n = cnt[kid]
    if not n: continue
    tot = sum(sums[kid * 24 + i] for i in range(1, 15))
    print(f"C={[32,64,128,256][kid//2]} {'block' if kid%2 else 'grid'}: {n} WGs, {tot/n:9.0f} cycles/WG:  " +
          "  ".join(f"{names[i]}={sums[kid*24+i]/n:.0f}" for i in range(1, 15) if sums[kid * 24 + i]))
