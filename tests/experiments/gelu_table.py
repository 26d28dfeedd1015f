"""CPU experiment (test infrastructure: it drives the oracle, so it lives under tests/): score-map error when GELU is
evaluated from a table with linear interpolation (value + slope per node, fp32 arithmetic: y = a_i + b_i * x) instead of
erf.  Patches the oracle's F.gelu.  Usage: python tests/experiments/gelu_table.py [H W]"""
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
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(H, W, 0))
from balf_amd import pipeline
x = pipeline.pad_batch(np.stack([img]))


def gelu64(v):
    return (v.double() * 0.5 * (1.0 + torch.erf(v.double() / np.sqrt(2.0))))


def make_table(n, lim):
    """n intervals over [-lim, lim): the line through the interval's end points, shifted by half its maximum
    deviation (equi-oscillating chord), as fp32 (a, b) with y = a + b x."""
    edges = torch.linspace(-lim, lim, n + 1, dtype=torch.float64)
    g = gelu64(edges)
    b = (g[1:] - g[:-1]) / (edges[1:] - edges[:-1])
    a = g[:-1] - b * edges[:-1]
    mid = 0.5 * (edges[1:] + edges[:-1])
    a = a + 0.5 * (gelu64(mid) - (a + b * mid))
    return a.float(), b.float()


def table_gelu(n, lim):
    a, b = make_table(n, lim)
    scale = np.float32(n / (2.0 * lim))

    def f(v):
        v = v.float()
        y = torch.clamp(v * np.float32(1.0 / (2 * lim)) + np.float32(0.5), 0.0, 1.0)      # v_fma ... clamp
        i = torch.clamp((y * np.float32(n)).floor().long(), 0, n - 1)
        out = a[i] + b[i] * v
        out = torch.where(v >= lim, v, out)              # outside: nodes n-1 / 0 are replaced by the exact asymptotes
        out = torch.where(v < -lim, torch.zeros_like(v), out)
        return out
    return f


def run(g):
    orig = F.gelu
    O.F.gelu = g
    try:
        with torch.no_grad():
            return O.detector_forward(sd, x)["prob"].numpy()
    finally:
        O.F.gelu = orig


ref = run(F.gelu)
xs = torch.linspace(-9, 9, 2000001)
for n, lim in [(4096, 6.0), (3072, 6.0), (2048, 6.0), (2048, 5.5), (1024, 6.0)]:
    g = table_gelu(n, lim)
    e1 = float((g(xs).double() - gelu64(xs)).abs().max())
    p = run(g)
    print(f"table {n:5d} nodes on [-{lim}, {lim}): pointwise max err {e1:.2e}; score map max-abs change {np.abs(p - ref).max():.2e}", flush=True)
