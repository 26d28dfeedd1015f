"""CPU experiment (test infrastructure: drives the oracle), round 5: could the two CORRECTION products of the split-f16 scheme --
(weight lo) x (activation hi) + (weight hi) x (activation lo) -- be carried by ONE int8 MFMA per K-step?
Every Linear of the oracle (and the 64x64 token mix) is emulated as
    main   = f16(w) . f16_rtz(x)                                   exact products, fp32 accumulation       (as today)
    corr   = 2^-10 sw[n] sx[t] . ( q8(wl / (2^-10 sw[n])) . q8(xh / sx[t]) + q8(wh / sw[n]) . q8(xl / (2^-10 sx[t])) )
with per-output-row scales sw[n] = max_k |wh[n, k]| / 127 and per-token scales sx[t] = max_k |xh[t, k]| / 127, and compared
with the exact corrections (today's three-product scheme) and with the fp32 forward.  Usage: python tests/experiments/int8_corrections.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import oracle as O
from balf_amd.utils import synth
from balf_amd import pipeline
from tests.golden import cases

torch.set_num_threads(8)
sd = synth.synthetic_state_dict(cases.WEIGHT_SEED)
img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(256, 320, 0))
x0 = pipeline.pad_batch(np.stack([img]))
with torch.no_grad():
    ref = O.detector_forward(sd, x0)["prob"].numpy()


def f16_rtz(x):
    h = x.half().float()
    over = h.abs() > x.abs()
    step = torch.nextafter(h.half(), torch.zeros_like(h).half()).float()
    return torch.where(over, step, h)


def q8(v, scale):
    return torch.clamp(torch.round(v / scale), -127, 127)


MODE = {"name": "exact"}


def linear_emul(x, w, b):
    """x [..., K], w [N, K] -> [..., N] with the operand scheme of MODE."""
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).float()
    wh = w.half().float()
    wl = (w - wh).half().float()
    xh = f16_rtz(x2)
    xl = (x2 - xh).half().float()
    main = (xh.double() @ wh.double().T)
    if MODE["name"] == "exact":
        corr = xh.double() @ wl.double().T + xl.double() @ wh.double().T
    elif MODE["name"] == "none":
        corr = 0.0
    else:
        sw = wh.abs().amax(dim=1).clamp_min(1e-30) / 127.0                      # [N]
        sx = xh.abs().amax(dim=1).clamp_min(1e-30) / 127.0                      # [T]
        shift = 2.0 ** -10
        qwh, qwl = q8(wh, sw[:, None]), q8(wl, sw[:, None] * shift)
        qxh, qxl = q8(xh, sx[:, None]), q8(xl, sx[:, None] * shift)
        ci = qxh.double() @ qwl.double().T + qxl.double() @ qwh.double().T       # exact integers
        corr = ci * (sx[:, None].double() * sw[None, :].double() * shift)
    out = (main + corr).float()
    if b is not None:
        out = out + b
    return out.reshape(*shp[:-1], w.shape[0])


orig_lin, orig_einsum = O._lin, torch.einsum


def lin(sd_, p, v):
    return linear_emul(v, sd_[p + ".weight"], sd_[p + ".bias"])


def einsum(eq, *ops):
    # the token mix: "pqgh,ngihjc->npiqjc" (grid) / "pqgh,nygxhc->nypxqc" (block): a 64x64 matrix applied along the token axes
    if len(ops) == 2 and ops[0].dim() == 4 and ops[1].dim() == 6:
        w4, b6 = ops
        w = w4.reshape(64, 64)
        if eq.startswith("pqgh,ngihjc"):
            n, g, i, h, j, c = b6.shape
            t = b6.permute(0, 2, 4, 5, 1, 3).reshape(-1, 64)                   # [..., (g h)]
            o = linear_emul(t, w, None).reshape(n, i, j, c, 8, 8).permute(0, 4, 1, 5, 2, 3)
            return o
        n, y, g, xx, h, c = b6.shape
        t = b6.permute(0, 1, 3, 5, 2, 4).reshape(-1, 64)
        o = linear_emul(t, w, None).reshape(n, y, xx, c, 8, 8).permute(0, 1, 4, 2, 5, 3)
        return o
    return orig_einsum(eq, *ops)


O._lin = lin
O.torch.einsum = einsum
for name in ("exact", "int8", "none"):
    MODE["name"] = name
    with torch.no_grad():
        p = O.detector_forward(sd, x0)["prob"].numpy()
    print(f"corrections {name:6s}: score-map max-abs error vs the fp32 forward {np.abs(p - ref).max():.2e}", flush=True)
