"""Per-phase cycle breakdown of the head kernel (diagnostic build: tools/build_variant.sh stamps -DBALF_STAMPS=1 ->
balf_amd/libbalf_hip_stamps.so): wave 0 of every workgroup stamps s_memtime between the phases (the STAMPV points of
head_kernel16_ns, kept in a register per lane: no memory traffic between stamps)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BALF_HIP_LIB"] = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "balf_amd/libbalf_hip_stamps.so")
os.environ["BALF_FP16_CHECK"] = "0"
from balf_amd import arch
from balf_amd.model import get_model
from balf_amd.utils import synth
m = get_model.load_model(arch.DEFAULT_MODEL_CFG); m.load_state_dict(synth.synthetic_state_dict(1)); m.precision = "fp16"
m = m.eval().cuda()
x = torch.rand((16, 3, 1088, 1920), device="cuda")
raw = C.CDLL(os.environ["BALF_HIP_LIB"])
sums = (C.c_ulonglong * 40)(); cnt = (C.c_ulonglong * 1)()
m(x, want_logits=False); torch.cuda.synchronize()
raw.balf_debug_head_stamps(sums, cnt, 1)
for _ in range(2): m(x, want_logits=False)
torch.cuda.synchronize()
raw.balf_debug_head_stamps(sums, cnt, 0)
names = ["", "load t, r, scale + split + stage", "barrier", "conv2 (384 MFMA)", "barrier", "relu + split + stage", "barrier",
         "head Linear (120 MFMA)", "BatchNorm + max (+ logits)", "exp x 20 + sums", "normalise + shuffle + prob store"]
n = cnt[0]
v = [sums[i] / n for i in range(40)]
tot = sum(v[1:len(names)])
print(f"head kernel: {n} workgroups, {tot:.0f} cycles per workgroup (wave 0):")
for i in range(1, len(names)):
    print(f"   {names[i]:36s} {v[i]:8.0f}  {100 * v[i] / tot:5.1f} %")
