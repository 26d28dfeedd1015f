"""Stage-3 / stage-4 debugging aid for the wave-team kernels (stage34_f16.h).  Runs the split-f16 forward and compares every
tensor the stage leaves in the workspace -- the grid branch's u', x1 (stage 3) or r = x1 + x0 and t (stage 4), the partial
channel sums, the SE scale -- with the oracle's taps computed FROM THE STAGE INPUT THE GPU ITSELF PRODUCED (X3 / X4 decoded out
of the workspace), so that a wrong kernel is named whatever the earlier stages did.
  stage 4:  BALF_FP16_CHECK=0 python tools/s34_debug.py 4
  stage 3:  tools/build_variant.sh dbg -DBALF_DEBUG_STOP=1
            BALF_HIP_LIB=balf_amd/libbalf_hip_dbg.so BALF_DEBUG_STOP_STAGE=3 BALF_FP16_CHECK=0 python tools/s34_debug.py 3"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balf_amd import _lib, arch, ops                                   # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402
from oracle import oracle as O                                         # noqa: E402
from tools.s2_debug import plan, frag32, frag16                        # noqa: E402


def main():
    stage = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    b, h, w = (2, 128, 192) if len(sys.argv) < 5 else (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    sd = synth.synthetic_state_dict(11)
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(sd)
    m.precision = "fp16"
    m = m.eval().cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.rand((b, 3, h, w), generator=g)
    with torch.inference_mode():
        m(x.cuda())
    torch.cuda.synchronize()
    print("library:", _lib.lib().balf_build_flags().decode()[:40], "stop stage", os.environ.get("BALF_DEBUG_STOP_STAGE"))
    ws = ops._workspace("forward", torch.device("cuda:0"), 0).cpu()
    off, total = plan(b, h, w)
    c, cin = (128, 64) if stage == 3 else (256, 128)
    sh = 4 if stage == 3 else 8
    hs, wsz = h // sh, w // sh
    npix = b * hs * wsz
    d = f"down{stage}"
    xin = frag16(ws[off["X3" if stage == 3 else "X4"]:], npix, cin).reshape(b, hs, wsz, cin)
    taps = {}
    with torch.no_grad():
        out_ref = O.stage_forward(sd, d, xin, last=(stage == 4), taps=taps)
        x0 = F.relu(O._lin(sd, f"{d}.conv.0", xin))

    def rel(name, got, ref):
        ref = ref.reshape(got.shape)
        e = (got - ref).abs()
        print(f"{name:10s} max-abs {float(e.max()):.3e} of {float(ref.abs().max()):.3f}   mean-abs {float(e.mean()):.3e}")
        return e

    def breakdown(name, e):
        if float(e.max()) <= 1e-3:
            return
        ee = e.reshape(b, hs, wsz, c)
        print(f"   {name} error by channel block of 16:", [f"{float(ee[..., k:k + 16].max()):.1e}" for k in range(0, c, 16)])
        print(f"   {name} error by row (first image):", [f"{float(ee[0, y].max()):.1e}" for y in range(hs)])
        print(f"   {name} error by col (first image):", [f"{float(ee[0, :, xx].max()):.1e}" for xx in range(wsz)])

    breakdown("u'", rel("u'", frag32(ws[off["U"]:], npix, c).reshape(b, hs, wsz, c), taps[d + ".u"]))
    rr = ws[off["R"]:off["R"] + npix * c * 4].view(torch.float32).view(b, hs, wsz, c)
    if stage == 3:
        breakdown("x1", rel("x1", rr, taps[d + ".x1"]))
    else:
        breakdown("r", rel("r=x1+x0", rr, taps[d + ".x1"] + x0))
        tt = ws[off["T"]:off["T"] + npix * c * 4].view(torch.float32).view(b, hs, wsz, c)
        breakdown("t", rel("t", tt, taps[d + ".t"]))
    r = f"{d}.residual_channel_attention_block"
    with torch.no_grad():
        hid = F.leaky_relu(O._lin(sd, r + ".conv1", O._ln(sd, r + ".norm", taps[d + ".x1"])), 0.2)
    src = hid if stage == 3 else taps[d + ".t"]
    fh, fw = hs // 8, wsz // 8
    part = ws[off["partial"]:off["partial"] + b * fh * fw * c * 4].view(torch.float32).view(b, fh, fw, c)
    ref_part = src.reshape(b, fh, 8, fw, 8, c).sum(dim=(2, 4))
    e = rel("partial", part, ref_part)
    if float(e.max()) > 1e-2:
        print("   partial error by channel block of 16:", [f"{float(e[..., k:k + 16].max()):.1e}" for k in range(0, c, 16)])
    sc = ws[off["scale"]:off["scale"] + b * c * 4].view(torch.float32).view(b, c)
    rel("SE scale", sc, taps[d + ".s"])
    if stage == 3:
        rel("X4 (out)", frag16(ws[off["X4"]:], b * (hs // 2) * (wsz // 2), c), out_ref.reshape(-1, c))


if __name__ == "__main__":
    main()
