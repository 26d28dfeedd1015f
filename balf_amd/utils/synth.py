"""Seeded synthetic weights and images (no pretrained weights exist offline: SURVEY.md F7).

Both sides of every parity check -- the golden-fixture generator that loads these
weights into the *reference* model in the build container, and the GPU tests / bench on
the GPU box -- call :func:`synthetic_state_dict` with the same seed, so a 5 MB weight
fixture never has to travel.  Values come from a NumPy PCG64 stream keyed by the entry's
position in the state-dict table, so they do not depend on torch's RNG.

The scales are chosen so that the score map is non-degenerate (default torch init gives
prob ~= 1/65 everywhere): LayerNorm/BatchNorm affine terms are randomised and the head
is given enough gain for a peaky 65-way softmax.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from ..arch import DEFAULT_ARCH, state_entries


def synthetic_state_dict(seed: int = 0, arch=None) -> Dict[str, torch.Tensor]:
    arch = dict(DEFAULT_ARCH if arch is None else arch)
    sd: Dict[str, torch.Tensor] = {}
    for i, (name, shape, dtype) in enumerate(state_entries(arch)):
        rng = np.random.Generator(np.random.PCG64([int(seed), i]))
        if dtype == "int64":
            sd[name] = torch.tensor(0, dtype=torch.int64)
            continue
        leaf = name.rsplit(".", 1)[1]
        is_norm = ".norm." in name
        if name.startswith("detector_head.norm"):
            if leaf == "weight":
                v = 1.0 + 0.25 * rng.standard_normal(shape)
            elif leaf == "bias":
                v = 0.2 * rng.standard_normal(shape)
            elif leaf == "running_mean":
                v = 0.3 * rng.standard_normal(shape)
            else:  # running_var
                v = rng.uniform(0.5, 1.5, shape)
        elif is_norm:
            v = 1.0 + 0.2 * rng.standard_normal(shape) if leaf == "weight" else 0.1 * rng.standard_normal(shape)
        elif leaf == "weight":
            fan_in = shape[1]
            gain = 1.0
            if name.endswith("gating_unit.dense.weight"):
                gain = 0.7
            elif name.startswith("detector_head.dense"):
                gain = 0.4
            elif name.endswith("conv.0.weight") and fan_in == 3:
                gain = 2.0
            v = gain * rng.standard_normal(shape) / np.sqrt(fan_in)
        else:  # Linear bias
            v = 0.1 * rng.standard_normal(shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synthetic_gray_u8(h: int, w: int, index: int = 0, blur: int = 5) -> np.ndarray:
    """Deterministic grayscale test image: uniform noise, box-blurred so that
    it has spatial structure, stretched back to the full 0..255 range."""
    rng = np.random.default_rng(1234 + int(index))
    g = rng.integers(0, 256, size=(h, w)).astype(np.float64)
    if blur > 1:
        k = blur
        pad = np.pad(g, ((k // 2, k // 2), (k // 2, k // 2)), mode="reflect")
        c = np.cumsum(np.cumsum(pad, axis=0), axis=1)
        c = np.pad(c, ((1, 0), (1, 0)))
        g = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
        lo, hi = g.min(), g.max()
        g = (g - lo) / max(hi - lo, 1e-12) * 255.0
    return np.clip(np.rint(g), 0, 255).astype(np.uint8)


def gray_to_rgb_norm(gray_u8: np.ndarray) -> np.ndarray:
    """uint8 gray -> [H,W,3] float64 in [0,1], what demo_match.detect feeds in
    (/root/reference/demo/demo_match.py:21-22 with F4's channel replication)."""
    g = gray_u8.astype(np.float64) / 255.0
    return np.stack([g, g, g], axis=-1)


# ----------------------------------------------------------------------------------------
# HardNet descriptor (demo path; /root/reference/third_party/hardnet/hardnet_pytorch.py:31-55)
# ----------------------------------------------------------------------------------------
# (features index of the conv, out channels, in channels, kernel); BatchNorm2d(affine=False) sits at index + 1
HARDNET_CONVS = ((0, 32, 1, 3), (3, 32, 32, 3), (6, 64, 32, 3), (9, 64, 64, 3), (12, 128, 64, 3),
                 (15, 128, 128, 3), (19, 128, 128, 8))


def hardnet_state_entries():
    out = []
    for idx, co, ci, k in HARDNET_CONVS:
        out.append((f"features.{idx}.weight", (co, ci, k, k), "float32"))
        out.append((f"features.{idx + 1}.running_mean", (co,), "float32"))
        out.append((f"features.{idx + 1}.running_var", (co,), "float32"))
        out.append((f"features.{idx + 1}.num_batches_tracked", (), "int64"))
    return out


def synthetic_hardnet_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded HardNet weights (He-scaled convs, randomised BatchNorm running statistics), in the
    reference's state-dict order.  No pretrained HardNet++ checkpoint exists offline."""
    sd: Dict[str, torch.Tensor] = {}
    for i, (name, shape, dtype) in enumerate(hardnet_state_entries()):
        rng = np.random.Generator(np.random.PCG64([int(seed), 7000 + i]))
        if dtype == "int64":
            sd[name] = torch.tensor(0, dtype=torch.int64)
            continue
        if name.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif name.endswith("running_mean"):
            v = 0.2 * rng.standard_normal(shape)
        else:
            v = rng.uniform(0.5, 1.5, shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synthetic_patches(n: int, seed: int = 0) -> torch.Tensor:
    """[n,1,32,32] float32 patches in [0,1] with some spatial structure (smoothed noise + a ramp)."""
    rng = np.random.default_rng(4321 + int(seed))
    g = rng.random((n, 36, 36), dtype=np.float32)
    sm = (g[:, :-4, :-4] + g[:, 2:-2, 2:-2] + g[:, 4:, 4:] + g[:, :-4, 4:] + g[:, 4:, :-4]) / 5.0
    ramp = np.linspace(0.0, 0.3, 32, dtype=np.float32)[None, :, None] * rng.random((n, 1, 1), dtype=np.float32)
    return torch.from_numpy(np.clip(sm + ramp, 0.0, 1.0).astype(np.float32)).unsqueeze(1)
