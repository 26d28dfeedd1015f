"""CPU experiment (test infrastructure: drives the oracle): which Linears tolerate f16-rounded weights, i.e. could drop the
(weight lo) x (activation hi) product of the split-f16 scheme?  Rounds the weights of one group of Linears at a time to
f16 (round-to-nearest, what the hi plane holds) and reports the score-map change.  Usage: python tests/experiments/wlo_sensitivity.py"""
import sys
import re
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import oracle as O
from balf_amd.utils import synth
from balf_amd import pipeline
from tests.golden import cases

torch.set_num_threads(8)
sd = synth.synthetic_state_dict(cases.WEIGHT_SEED)
img = synth.gray_to_rgb_norm(synth.synthetic_gray_u8(480, 640, 0))
x = pipeline.pad_batch(np.stack([img]))
with torch.no_grad():
    ref = O.detector_forward(sd, x)["prob"].numpy()


def run(pattern):
    sd2 = dict(sd)
    n = 0
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 2 and re.search(pattern, k):
            sd2[k] = v.half().float()
            n += 1
    with torch.no_grad():
        p = O.detector_forward(sd2, x)["prob"].numpy()
    return n, float(np.abs(p - ref).max())


rsh = "residual_split_head_multi_axis_gmlp_layer"
groups = [("all Linears", r"."),
          ("token-mix matrices (all stages)", r"gating_unit\.dense\."),
          ("token-mix, stage 1", r"down1\..*gating_unit\.dense\."),
          ("token-mix, stage 2", r"down2\..*gating_unit\.dense\."),
          ("conv0 (all stages)", r"down\d\.conv\.0\."),
          ("RSHMAG dense1", rsh + r"\.dense1\."),
          ("RSHMAG dense2", rsh + r"\.dense2\."),
          ("branch dense1", r"gmlp_layer\.dense1\."),
          ("branch dense2", r"gmlp_layer\.dense2\."),
          ("RCAB conv1", r"rcab.*conv1\.|residual_channel.*conv1\."),
          ("RCAB conv2", r"rcab.*conv2\.|residual_channel.*conv2\."),
          ("stage 1 (all)", r"down1\."), ("stage 2 (all)", r"down2\."), ("stage 3 (all)", r"down3\."), ("stage 4 (all)", r"down4\."),
          ("head", r"detector_head\.")]
for name, pat in groups:
    n, e = run(pat)
    print(f"{name:36s} {n:3d} tensors  score-map max-abs change {e:.2e}", flush=True)
