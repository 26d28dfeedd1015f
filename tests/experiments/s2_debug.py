"""Stage-2 debugging aid (diagnostic build: tools/build_variant.sh dbg -DBALF_DEBUG_STOP=1; run with
BALF_HIP_LIB=balf_amd/libbalf_hip_dbg.so BALF_DEBUG_STOP_STAGE=2 BALF_FP16_CHECK=0): run the split-f16 forward up to the end of
stage 2 and compare every tensor that stage leaves in the workspace -- its input X2, the grid branch's u', x1, the squeeze-excite
scale and its output X3 -- with the oracle's taps (oracle.stage_forward), so that a wrong kernel is named."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from balf_amd import _lib, arch, ops                                   # noqa: E402
from balf_amd.model import get_model                                   # noqa: E402
from balf_amd.utils import synth                                       # noqa: E402
from oracle import oracle as O                                         # noqa: E402


def plan(b, hp, wp):
    """make_plan of det_common.h"""
    px = hp * wp
    mb = _lib.lib().balf_forward_micro_batch(b, hp, wp)
    o, off = 0, {}

    def take(name, floats):
        nonlocal o
        off[name] = o
        o = (o + floats * 4 + 255) // 256 * 256
    big = mb * px * 32
    take("U", big); take("T", big); take("R", big)
    take("X2", mb * (px // 4) * 32); take("X3", mb * (px // 16) * 64); take("X4", mb * (px // 64) * 128)
    take("partial", mb * (px // 64) * 32); take("chunk", mb * 128 * 256); take("scale", mb * 256)
    return off, o


def frag32(raw_u8, npix, c):
    """32x32 fragment format -> [npix, c] float32"""
    h = raw_u8[:npix * c * 4].view(torch.float16).view(npix, c // 16, 2, 2, 8).float()      # [pix][K-step][hi|lo][lane half][j]
    v = h[:, :, 0] + h[:, :, 1]                                                              # [pix][s][h][j]
    out = torch.empty((npix, c))
    for s in range(c // 16):
        for hh in range(2):
            for j in range(8):
                out[:, 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)] = v[:, s, hh, j]
    return out


def frag16(raw_u8, npix, c):
    h = raw_u8[:npix * c * 4].view(torch.float16).view(npix, c // 32, 2, 4, 8).float()      # [pix][K-step][hi|lo][quarter][j]
    v = h[:, :, 0] + h[:, :, 1]
    out = torch.empty((npix, c))
    for ks in range(c // 32):
        for q in range(4):
            for j in range(8):
                out[:, 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3)] = v[:, ks, q, j]
    return out


def main():
    b, h, w = 2, 128, 192
    sd = synth.synthetic_state_dict(11)
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(sd)
    m.precision = "fp16"
    m = m.eval().cuda()
    g = torch.Generator().manual_seed(3)
    x = torch.rand((b, 3, h, w), generator=g)
    taps = {}
    with torch.no_grad():
        O.detector_forward(sd, x, taps)
        t = x.permute(0, 2, 3, 1)
        x2_in = O.stage_forward(sd, "down1", t, last=False)
        x3_ref = O.stage_forward(sd, "down2", x2_in, last=False)
    with torch.inference_mode():
        m(x.cuda())
    torch.cuda.synchronize()
    print("library:", _lib.lib().balf_build_flags().decode()[:40], "stop stage", os.environ.get("BALF_DEBUG_STOP_STAGE"))
    ws = ops._workspace("forward", torch.device("cuda:0"), 0).cpu()
    off, total = plan(b, h, w)
    h2, w2 = h // 2, w // 2
    npix = b * h2 * w2

    def rel(name, got, ref):
        ref = ref.reshape(got.shape)
        e = (got - ref).abs()
        print(f"{name:10s} max-abs {float(e.max()):.3e} of {float(ref.abs().max()):.3f}   mean-abs {float(e.mean()):.3e}")
        return e
    rel("X2 (in)", frag32(ws[off["X2"]:], npix, 32), x2_in.reshape(npix, 32))
    e = rel("u'", frag32(ws[off["U"]:], npix, 64), taps["down2.u"].reshape(npix, 64))
    if float(e.max()) > 1e-3:
        ee = e.reshape(b, h2, w2, 64)
        print("   u' error by channel block of 8:", [f"{float(ee[..., c:c + 8].max()):.1e}" for c in range(0, 64, 8)])
        print("   u' error by image row (first image):", [f"{float(ee[0, y].max()):.1e}" for y in range(0, h2, max(1, h2 // 16))])
        print("   u' error by image col (first image):", [f"{float(ee[0, :, xx].max()):.1e}" for xx in range(0, w2, max(1, w2 // 16))])
    # x1 in register order: [item][wave half][tile][quad][lane] x 4 floats
    fh, fw = h2 // 8, w2 // 8
    raw = ws[off["R"]:off["R"] + npix * 64 * 4].view(torch.float32).view(b, fh, fw, 2, 2, 4, 64, 4)
    x1 = torch.empty((b, h2, w2, 64))
    for wv in range(2):
        for lane in range(64):
            n, hh = lane & 31, lane >> 5
            ty, tx = 4 * wv + (n >> 3), n & 7
            for rt in range(2):
                for gq in range(4):
                    c0 = 32 * rt + 8 * gq + 4 * hh
                    x1[:, ty::8, tx::8, c0:c0 + 4] = raw[:, :, :, wv, rt, gq, lane, :]
    e = rel("x1", x1, taps["down2.x1"])
    if float(e.max()) > 1e-3:
        print("   x1 error by channel block of 8:", [f"{float(e[..., c:c + 8].max()):.1e}" for c in range(0, 64, 8)])
        print("   x1 error by token row ty:", [f"{float(e[:, ty::8].max()):.1e}" for ty in range(8)])
        print("   x1 error by token col tx:", [f"{float(e[:, :, tx::8].max()):.1e}" for tx in range(8)])
    # channel sums of the RCAB's hidden layer h = lrelu(conv1(LN(x1))): one partial row per wave half
    import torch.nn.functional as F
    r = "down2.residual_channel_attention_block"
    with torch.no_grad():
        hid = F.leaky_relu(O._lin(sd, r + ".conv1", O._ln(sd, r + ".norm", taps["down2.x1"])), 0.2)
    part = ws[off["partial"]:off["partial"] + b * fh * fw * 2 * 64 * 4].view(torch.float32).view(b, fh, fw, 2, 64)
    hid6 = hid.reshape(b, fh, 8, fw, 8, 64)
    ref_part = torch.stack([hid6[:, :, 0:4].sum(dim=(2, 4)), hid6[:, :, 4:8].sum(dim=(2, 4))], dim=3)     # [b, fh, fw, half, c]
    e = rel("partial", part, ref_part)
    if float(e.max()) > 1e-2:
        print("   partial error by channel:", [f"{float(e[..., c].max()):.1e}" for c in range(0, 64)])
        print("   got[0,0,0,0,:8]", part[0, 0, 0, 0, :8].tolist(), "ref", ref_part[0, 0, 0, 0, :8].tolist())
    sc = ws[off["scale"]:off["scale"] + b * 64 * 4].view(torch.float32).view(b, 64)
    rel("SE scale", sc, taps["down2.s"])
    e = rel("X3 (out)", frag16(ws[off["X3"]:], b * (h2 // 2) * (w2 // 2), 64), x3_ref.reshape(-1, 64))
    if float(e.max()) > 1e-3:
        ee = e.reshape(b, h2 // 2, w2 // 2, 64)
        print("   X3 error by channel block of 8:", [f"{float(ee[..., c:c + 8].max()):.1e}" for c in range(0, 64, 8)])
        print("   X3 error by pooled row parity / col parity:", [f"{float(ee[:, p::2].max()):.1e}" for p in range(2)], [f"{float(ee[:, :, p::2].max()):.1e}" for p in range(2)])


if __name__ == "__main__":
    main()
