"""CPU experiment (test infrastructure: drives the oracle): which Linears tolerate f16-rounded INPUT activations, i.e. could
drop the (weight hi) x (activation lo) product AND the residual half of the operand split?  Rounds the input of one group of
Linears (by weight-key pattern; the token mix by its einsum) to f16 and reports the score-map change."""
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

orig_lin, orig_einsum = O._lin, torch.einsum
state = {"pat": None, "mix": None}


def lin(sd_, p, v):
    if state["pat"] is not None and re.search(state["pat"], p):
        v = v.half().float()
    return orig_lin(sd_, p, v)


def einsum(eq, *ops):
    if state["mix"] and len(ops) == 2:
        ops = tuple(o.half().float() if o.dim() > 2 else o for o in ops)     # the activation operand, not the [64, 64] matrix
    return orig_einsum(eq, *ops)


O._lin = lin
O.torch.einsum = einsum
rsh = "residual_split_head_multi_axis_gmlp_layer"
groups = [("all Linears + token mix", r".", True), ("token mix only", None, True),
          ("conv0", r"down\d\.conv\.0$", False), ("RSHMAG dense1", rsh + r"\.dense1$", False), ("RSHMAG dense2", rsh + r"\.dense2$", False),
          ("branch dense1", r"gmlp_layer\.dense1$", False), ("branch dense2", r"gmlp_layer\.dense2$", False),
          ("RCAB conv1", r"attention_block\.conv1$", False), ("RCAB conv2", r"attention_block\.conv2$", False),
          ("stage-4 conv2 + head", r"down4\.conv2$|detector_head", False), ("stage 1 Linears", r"down1\.", False)]
for name, pat, mix in groups:
    state["pat"], state["mix"] = pat, mix
    with torch.no_grad():
        p = O.detector_forward(sd, x)["prob"].numpy()
    print(f"{name:28s} score-map max-abs change {np.abs(p - ref).max():.2e}", flush=True)
