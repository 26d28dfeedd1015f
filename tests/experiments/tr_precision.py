"""CPU experiment (test infrastructure: it drives the oracle, so it lives under tests/): score-map error when the
inter-kernel activations T (RCAB body) and R (= x1 + x0) are stored in fewer bytes.  Monkey-patches the oracle's stage
to round t and r before x2 = t*s + r.  Formats: f16 (2 B), bf16 (2 B), f16 hi + f16 lo (4 B, what fp32 storage gives),
'f16s': f16 with a per-channel power-of-two scale (no gain expected), 'e5m10+8' = f16 hi + 8-bit residual (3 B)."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import oracle as O
from balf_amd.utils import synth
from tests.golden import cases

torch.set_num_threads(8)
sd = synth.synthetic_state_dict(cases.WEIGHT_SEED)


def rnd(x, fmt):
    if fmt == "f32":
        return x
    if fmt == "f16":
        return x.half().float()
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "f16+8":
        hi = x.half().float()
        r = x - hi
        # residual in 8 bits relative to hi's ulp: quantise r to 1/256 of the f16 ulp of hi
        ulp = torch.where(hi == 0, torch.full_like(hi, 2.0 ** -24), 2.0 ** (torch.floor(torch.log2(hi.abs().clamp_min(2.0 ** -14))) - 10))
        return hi + torch.round(r / ulp * 128.0) / 128.0 * ulp
    raise ValueError(fmt)


def stage_forward(sd, d, x_nhwc, last, fmt_t, fmt_r):
    c = sd[f"{d}.conv.0.weight"].shape[0]
    x0 = F.relu(O._lin(sd, f"{d}.conv.0", x_nhwc))
    q = f"{d}.{O._RSH}"
    y = F.gelu(O._lin(sd, q + ".dense1", O._ln(sd, q + ".norm", x0)))
    u, v = y[..., :c], y[..., c:]
    u = O._gmlp_branch(sd, q + ".grid_gmlp_layer", "grid_gating_unit", u, True)
    v = O._gmlp_branch(sd, q + ".block_gmlp_layer", "block_gating_unit", v, False)
    x1 = O._lin(sd, q + ".dense2", torch.cat([u, v], dim=-1)) + x0
    r = f"{d}.{O._RCAB}"
    t = O._lin(sd, r + ".conv2", F.leaky_relu(O._lin(sd, r + ".conv1", O._ln(sd, r + ".norm", x1)), 0.2))
    m = t.mean(dim=(1, 2))
    s = torch.sigmoid(O._lin(sd, r + ".calayer.excite.2", F.relu(O._lin(sd, r + ".calayer.excite.0", m))))
    x2 = rnd(t, fmt_t) * s[:, None, None, :] + rnd(x1 + x0, fmt_r)
    if last:
        return O._lin(sd, f"{d}.conv2", x2)
    n, h, w, _ = x2.shape
    return x2.reshape(n, h // 2, 2, w // 2, 2, c).amax(dim=(2, 4))


def forward(x_nchw, fmt_t, fmt_r, stages=(0, 1, 2, 3)):
    x = x_nchw.permute(0, 2, 3, 1)
    for i in range(4):
        ft, fr = (fmt_t, fmt_r) if i in stages else ("f32", "f32")
        x = stage_forward(sd, f"down{i + 1}", x, i == 3, ft, fr)
    z = O._lin(sd, "detector_head.dense", F.relu(x))
    hp = "detector_head.norm."
    z = (z - sd[hp + "running_mean"]) / torch.sqrt(sd[hp + "running_var"] + 1e-5) * sd[hp + "weight"] + sd[hp + "bias"]
    p = torch.softmax(z, dim=-1)[..., :64]
    n, h, w, _ = p.shape
    return p.reshape(n, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(n, h * 8, w * 8)


img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(480, 640, 3, blur=5))
pad = O.mod_padding_symmetric(O.make_shape_even(img), 64)
x = torch.tensor(pad, dtype=torch.float32).permute(2, 0, 1).unsqueeze(0)
with torch.no_grad():
    ref = forward(x, "f32", "f32")
    for ft, fr, st in (("f16", "f16", (0, 1, 2, 3)), ("f16", "f32", (0, 1, 2, 3)), ("f32", "f16", (0, 1, 2, 3)),
                       ("bf16", "bf16", (0, 1, 2, 3)), ("f16", "f16", (0,)), ("f16", "f16", (3,)),
                       ("f16+8", "f16+8", (0, 1, 2, 3)), ("f16+8", "f16", (0, 1, 2, 3)), ("f16", "f16+8", (0, 1, 2, 3))):
        out = forward(x, ft, fr, st)
        print(f"T {ft:6s} R {fr:6s} stages {st}: prob max-abs err {float((out - ref).abs().max()):.3e}")
