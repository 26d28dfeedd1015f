"""Debugging aid for the fp32 stage-1 kernels (stage1_f32.h); test infrastructure: the tap mode drives the oracle.
  dump out.npy | cmp a.npy b.npy : stage-1 output (stage view) of the exact-fp32 forward for one library build / two dumps compared
  tap K : with a library built with -DBALF_F32_DBG=K (tools/build_variant.sh) and BALF_DEBUG_STOP_STAGE=1: the grid kernel's
          intermediate tensor K (1 x0, 2 LN(x0), 3 u, 4 gate input a, 5 gated a) against torch on the CPU"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def setup():
    import torch
    from balf_amd import arch
    from balf_amd.model import get_model
    from balf_amd.utils import synth
    sd = synth.synthetic_state_dict(11)
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(sd)
    m.precision = "fp32"
    m = m.eval().cuda()
    torch.manual_seed(3)
    x = torch.rand((2, 3, 128, 192), device="cuda")
    return torch, m, sd, x


if sys.argv[1] == "dump":
    torch, m, sd, x = setup()
    out = m(x, want_logits=False)
    v = m.stage_view(2, 128, 192)[0].cpu().numpy()
    np.save(sys.argv[2], {"view": v, "prob": out["prob"].cpu().numpy()}, allow_pickle=True)
    print("dumped", v.shape)
elif sys.argv[1] == "tap":
    import torch.nn.functional as F
    from balf_amd import ops
    from oracle import oracle as O
    k = int(sys.argv[2])
    torch, m, sd, x = setup()
    m(x, want_logits=False)
    torch.cuda.synchronize()
    ws = [w for kk, w in ops._workspaces.items() if kk[0] != "nms"]
    ws = max(ws, key=lambda t: t.numel())
    got = ws.view(torch.uint8)[: 2 * 128 * 192 * 32 * 4].view(torch.float32).view(2, 128, 192, 32).cpu()
    xc = x.cpu().permute(0, 2, 3, 1)
    d, q = "down1", "down1." + O._RSH
    x0 = F.relu(O._lin(sd, d + ".conv.0", xc))
    ln = lambda t: F.layer_norm(t, (32,), None, None, 1e-5)
    z = F.gelu(O._lin(sd, q + ".dense1", O._ln(sd, q + ".norm", x0)))[..., :32]
    p = q + ".grid_gmlp_layer"
    t = F.gelu(O._lin(sd, p + ".dense1", O._ln(sd, p + ".norm", z)))
    a, b = t[..., :32], t[..., 32:]
    b = O._ln(sd, p + ".grid_gating_unit.norm", b)
    w4 = sd[p + ".grid_gating_unit.dense.weight"].reshape(8, 8, 8, 8)
    b2 = sd[p + ".grid_gating_unit.dense.bias"].reshape(8, 8)
    b6 = b.reshape(2, 8, 16, 8, 24, 32)
    mix = (torch.einsum("pqgh,ngihjc->npiqjc", w4, b6) + b2[None, :, None, :, None, None]).reshape(2, 128, 192, 32)
    want = {1: x0, 2: ln(x0), 3: z, 4: a, 5: a * (mix + 1.0)}[k]
    dd = (got - want).abs()
    print("tap", k, "max abs diff", float(dd.max()), "of", float(want.abs().max()))
    print("per channel:", np.round(dd.reshape(-1, 32).amax(0).numpy(), 4))
    print("per token row (ty):", np.round(dd.reshape(2, 8, 16, 8, 24, 32).amax((0, 2, 4, 5)).numpy(), 4))
else:
    a = np.load(sys.argv[2], allow_pickle=True).item()
    b = np.load(sys.argv[3], allow_pickle=True).item()
    va, vb = a["view"], b["view"]
    d = np.abs(va - vb)
    print("stage-1 view max abs diff", d.max(), "of max", np.abs(vb).max(), "; prob diff", np.abs(a["prob"] - b["prob"]).max())
    print("per channel:", np.round(d.reshape(-1, d.shape[-1]).max(0), 4))
    B, H, W, C = d.shape
    pos = d.max(-1).reshape(B, H // 4, 4, W // 4, 4).max((0, 1, 3))
    print("per position in the pooled 4x4 block:\n", np.round(pos, 4))
