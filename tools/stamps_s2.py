"""Per-phase cycle breakdown of the persistent stage-2 kernels (diagnostic build: tools/build_variant.sh stamps -DBALF_STAMPS=1):
thread 0 of every workgroup (wave 0 = the first half of pair 0) stamps s_memtime between the phases of every token group."""
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
x = torch.rand((8, 3, 1088, 1920), device="cuda")
raw = C.CDLL(os.environ["BALF_HIP_LIB"])
NS = 40
sums = (C.c_ulonglong * (16 * NS))(); cnt = (C.c_ulonglong * 16)()
m(x, want_logits=False); torch.cuda.synchronize()
raw.balf_debug_stamps(sums, cnt, 1)
for _ in range(2): m(x, want_logits=False)
torch.cuda.synchronize()
raw.balf_debug_stamps(sums, cnt, 0)
common = ["", "in+conv0+LN", "q1+GELU", "LN+split", "d1a+GELU", "d1b+GELU+stats", "wait R0", "gLN+tile0", "wait W0", "mix0+gate", "wait R1",
          "gLN+tile1", "wait W1", "mix1+gate", "split+d2+res"]
names = {2: common + ["split+U store"],
         3: common + ["req+split v'", "wait c0", "q2 v' half", "wait u'+c2", "q2 u' half", "res+x1 store", "LN+r1+lrelu", "sums"]}
for kid in (2, 3):
    n = cnt[kid]
    if not n: continue
    nm = names[kid]
    tot = sum(sums[kid * NS + i] for i in range(1, len(nm)))
    print(f"stage2 {'block' if kid == 3 else 'grid'}: {n} groups stamped, {tot/n:8.0f} cycles/group:  " +
          "  ".join(f"{nm[i]}={sums[kid*NS+i]/n:.0f}" for i in range(1, len(nm))))
