"""Per-phase cycle breakdown of the channel-split stage kernels (diagnostic build: tools/build_variant.sh stamps -DBALF_STAMPS=1
-> balf_amd/libbalf_hip_stamps.so): wave 0 of every workgroup stamps s_memtime between the phases of its token group (the STAMPV(i) points of
stage_cs_f16.h, kept in a register per lane: no memory traffic between stamps), and every wave records the SIMD it runs on."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BALF_HIP_LIB"] = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "balf_amd/libbalf_hip_stamps.so")
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
names = ["", "input staged", "conv0", "relu+LNx+pub", "dense1 half", "GELU+LNx+pub", "d1 a", "GELU(a)+d1 b", "GELU(b)+gLNx+tile", "mix+gate",
         "pub+bar", "dense2", "res+U store | u'+res+bar+pub+bar", "q2+res+x1 store", "LNx+pub", "conv1", "lrelu(+xchg+conv2+T)+sums"]
for kid in range(2, 8):
    n = cnt[kid]
    if not n: continue
    v = [sums[kid * NS + i] / n for i in range(NS)]
    tot = sum(v[1:len(names)])
    print(f"C={[32,64,128,256][kid//2]} {'block' if kid%2 else 'grid'}: {n} groups, {tot:8.0f} cycles/group:  " +
          "  ".join(f"{names[i]}={v[i]:.0f}" for i in range(1, len(names)) if v[i]) +
          f"  || start-up detail: issued={v[17]:.0f} table={v[18]:.0f} arrived={v[19]:.0f} barrier={v[1]:.0f}")
for kid in range(2, 8):
    if cnt[kid]:
        print(f"kid {kid}: SIMD of wave slot w (rows) : " + " | ".join(" ".join(str(sums[(8 + kid) * NS + w * 4 + sd]) for sd in range(4)) for w in range(8)))
