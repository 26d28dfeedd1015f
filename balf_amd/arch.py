"""Static description of the BALF detector architecture (shapes only, no compute).

The reference builds its model from ``cfg['model']['network_architecture']``
(/root/reference/balf/configs/test.yaml:1-14, consumed at
/root/reference/balf/model/mlp_ma_decoder.py:249-256).  This module turns that dict
into the ordered list of state-dict entries (name, shape, dtype) the reference's
``MLP_MA_DECODER.state_dict()`` exposes -- 167 entries for the shipped config -- so
that the parameter container, the weight packer, the synthetic-weight generator and
the oracle all agree on one table.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

DEFAULT_ARCH: Dict[str, object] = {
    "en_embed_dims": [3, 32, 64, 128, 256],
    "grid_size": [8, 8],
    "block_size": [8, 8],
    "grid_gmlp_factor": 2,
    "block_gmlp_factor": 2,
    "input_proj_factor": 2,
    "channels_reduction": 4,
    "out_channels": 65,
    "cell_size": 8,
}

DEFAULT_MODEL_CFG = {"name": "mlp_ma_decoder", "network_architecture": dict(DEFAULT_ARCH)}

RSH = "residual_split_head_multi_axis_gmlp_layer"
RCAB = "residual_channel_attention_block"


def stage_entries(prefix: str, cin: int, c: int, arch: Dict[str, object]) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) pairs of one ``Down`` stage in registration order
    (mlp_ma_decoder.py:201-221 and the sub-module constructors it calls)."""
    gtok = int(arch["grid_size"][0]) * int(arch["grid_size"][1])
    btok = int(arch["block_size"][0]) * int(arch["block_size"][1])
    gf, bf, pf = int(arch["grid_gmlp_factor"]), int(arch["block_gmlp_factor"]), int(arch["input_proj_factor"])
    red = int(arch["channels_reduction"])
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def lin(name, o, i):
        out.append((f"{prefix}.{name}.weight", (o, i)))
        out.append((f"{prefix}.{name}.bias", (o,)))

    def ln(name, n):
        out.append((f"{prefix}.{name}.weight", (n,)))
        out.append((f"{prefix}.{name}.bias", (n,)))

    lin("conv.0", c, cin)
    ln(f"{RSH}.norm", c)
    lin(f"{RSH}.dense1", c * pf, c)
    for br, unit, fac, tok in (("grid_gmlp_layer", "grid_gating_unit", gf, gtok),
                               ("block_gmlp_layer", "block_gating_unit", bf, btok)):
        ln(f"{RSH}.{br}.norm", c)
        lin(f"{RSH}.{br}.dense1", c * fac, c)
        ln(f"{RSH}.{br}.{unit}.norm", c)
        lin(f"{RSH}.{br}.{unit}.dense", tok, tok)
        lin(f"{RSH}.{br}.dense2", c, c)
    lin(f"{RSH}.dense2", c, c * pf)
    ln(f"{RCAB}.norm", c)
    lin(f"{RCAB}.conv1", c, c)
    lin(f"{RCAB}.conv2", c, c)
    lin(f"{RCAB}.calayer.excite.0", c // red, c)
    lin(f"{RCAB}.calayer.excite.2", c, c // red)
    lin("conv2", c, c)
    return out


def state_entries(arch: Dict[str, object] = None) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Ordered (name, shape, dtype) of every state-dict entry (get_model.py:60-67 filters on
    exactly these names and shapes)."""
    arch = dict(DEFAULT_ARCH if arch is None else arch)
    dims = list(arch["en_embed_dims"])
    ents: List[Tuple[str, Tuple[int, ...], str]] = []
    for s in range(4):
        for name, shape in stage_entries(f"down{s + 1}", dims[s], dims[s + 1], arch):
            ents.append((name, shape, "float32"))
    nout = int(arch["cell_size"]) ** 2 + 1
    ents.append(("detector_head.dense.weight", (nout, dims[4]), "float32"))
    ents.append(("detector_head.dense.bias", (nout,), "float32"))
    ents.append(("detector_head.norm.weight", (nout,), "float32"))
    ents.append(("detector_head.norm.bias", (nout,), "float32"))
    ents.append(("detector_head.norm.running_mean", (nout,), "float32"))
    ents.append(("detector_head.norm.running_var", (nout,), "float32"))
    ents.append(("detector_head.norm.num_batches_tracked", (), "int64"))
    return ents


def check_supported(arch: Dict[str, object]) -> None:
    """The HIP path is specialised for the one architecture the reference ships
    (test.yaml:4-12); anything else is rejected loudly rather than run on a fallback."""
    a = dict(arch)
    want = DEFAULT_ARCH
    for k in ("en_embed_dims", "grid_size", "block_size", "grid_gmlp_factor", "block_gmlp_factor",
              "input_proj_factor", "channels_reduction", "cell_size"):
        if k not in a:
            raise KeyError(k)  # same failure mode as mlp_ma_decoder.py:249-256
        if list(a[k]) != list(want[k]) if isinstance(want[k], list) else int(a[k]) != int(want[k]):
            raise NotImplementedError(
                f"balf_amd HIP kernels are built for network_architecture.{k}={want[k]!r}, got {a[k]!r}")


def padded_hw(h: int, w: int, factor: int = 64) -> Tuple[int, int, int, int]:
    """(Hp, Wp, top, left) after make_shape_even + mod_padding_symmetric
    (/root/reference/balf/utils/test_utils.py:16-32) and the crop offsets callers use
    (/root/reference/balf/utils/train_utils.py:437-442)."""
    he, we = h + (h & 1), w + (w & 1)
    hp = he if he % factor == 0 else (he // factor + 1) * factor
    wp = we if we % factor == 0 else (we // factor + 1) * factor
    top = hp // 2 - he // 2
    left = wp // 2 - we // 2
    return hp, wp, top, left


FLOP_PER_PADDED_PIXEL = 2 * 59748  # Linear layers only
