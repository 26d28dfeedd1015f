"""Per-phase cycle breakdown of the HardNet conv kernels (diagnostic build libbalf_hip_hnstamps.so, -DBALF_HN_STAMPS=1)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
os.environ["BALF_HIP_LIB"] = os.path.abspath("balf_amd/libbalf_hip_hnstamps.so")
from balf_amd.third_party.hardnet.hardnet_pytorch import HardNet
from balf_amd.utils import synth
m = HardNet(); m.load_state_dict(synth.synthetic_hardnet_state_dict(515)); m = m.eval().cuda()
x = synth.synthetic_patches(2048, 3).cuda().repeat(16, 1, 1, 1).contiguous()
raw = C.CDLL(os.environ["BALF_HIP_LIB"])
buf = (C.c_ulonglong * 64)()
m(x); torch.cuda.synchronize()
raw.balf_debug_hn_stamps(buf, 1)
m(x); torch.cuda.synchronize()
raw.balf_debug_hn_stamps(buf, 0)
names = ["conv1_2", "conv3", "conv4", "conv5", "conv6"]
ph = ["", "patch load+norm", "conv1->LDS | band load", "K loop", "epilogue"]
for k in range(5):
    n = buf[k * 8]
    if not n: continue
    tot = sum(buf[k * 8 + i] for i in range(1, 5))
    print(f"{names[k]:8s} {n} WGs {tot / n:8.0f} cycles/WG: " + "  ".join(f"{ph[i]}={buf[k*8+i]/n:.0f}" for i in range(1, 5) if buf[k * 8 + i]))
