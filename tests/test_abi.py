"""The C-ABI library loads and exports every symbol include/balf_hip.h declares (no compute: CPU only),
and the host-side entry points that need no GPU behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from balf_amd import _lib, arch
from balf_amd.utils import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def header_functions():
    src = open(os.path.join(ROOT, "include", "balf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(balf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_functions()
    assert len(names) >= 15
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in balf_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "balf_amd/_lib.py prototypes out of step with the header"


def test_version_and_error_strings(lib):
    assert lib.balf_abi_version() == 1
    for code in (0, -1, -2, -3, -4, -5):
        assert lib.balf_error_string(code)
    assert b"workspace" in lib.balf_error_string(-3)


def test_state_table_matches_reference_state_dict(lib):
    ents = [(n, s) for n, s, d in arch.state_entries() if d == "float32"]
    assert lib.balf_num_state_tensors() == len(ents) == 166
    for i, (n, s) in enumerate(ents):
        assert lib.balf_state_tensor_name(i).decode() == n
        assert lib.balf_state_tensor_numel(i) == int(np.prod(s))
    assert lib.balf_state_tensor_name(166) is None


def test_size_queries_and_argument_checks(lib):
    assert lib.balf_forward_workspace_bytes(1, 64, 64) > 0
    assert lib.balf_forward_workspace_bytes(1, 100, 64) == 0          # not a multiple of 64
    # the workspace holds one micro-batch (16 images at 1088x1920), whatever the batch
    assert lib.balf_forward_workspace_bytes(32, 1088, 1920) == lib.balf_forward_workspace_bytes(16, 1088, 1920)
    assert lib.balf_forward_workspace_bytes(8, 1088, 1920) < lib.balf_forward_workspace_bytes(16, 1088, 1920)
    assert lib.balf_forward_micro_batch(32, 1088, 1920) == 16 and lib.balf_forward_micro_batch(5, 1088, 1920) == 5
    assert lib.balf_forward_micro_batch(1, 100, 64) == 0
    # one padded image may hold 2^25 pixels at most (32-bit byte offsets inside an image): refused before anything is launched
    assert lib.balf_forward_workspace_bytes(1, 8192, 4096) > 0 and lib.balf_forward_workspace_bytes(1, 8192, 4160) == 0
    fake = C.c_void_p(4096)
    assert lib.balf_forward(fake, 1, fake, 1, 8192, 4160, None, fake, fake, 1 << 40, None) == -2
    # the stage view validates on the host before anything is launched: stage 1..4, the micro-batch that is resident, the workspace
    ws_bytes = lib.balf_forward_workspace_bytes(2, 64, 64)
    assert lib.balf_forward_stage_view_numel(2, 64, 64, 1) == 2 * 32 * 32 * 32 and lib.balf_forward_stage_view_numel(2, 64, 64, 4) == 2 * 8 * 8 * 256
    assert lib.balf_forward_stage_view_numel(2, 64, 64, 0) == 0 and lib.balf_forward_stage_view_numel(2, 64, 64, 5) == 0
    assert lib.balf_forward_stage_view(1, fake, ws_bytes, 2, 64, 64, 5, fake, None) == -1            # no such stage
    assert lib.balf_forward_stage_view(7, fake, ws_bytes, 2, 64, 64, 1, fake, None) == -1            # no such precision
    assert lib.balf_forward_stage_view(1, fake, ws_bytes - 1, 2, 64, 64, 1, fake, None) == -3        # workspace too small
    assert lib.balf_forward_stage_view(1, fake, 1 << 40, 17, 1088, 1920, 1, fake, None) == -1        # 17 images: two micro-batches
    assert lib.balf_forward_stage_view(1, fake, ws_bytes, 2, 60, 64, 1, fake, None) == -2
    assert lib.balf_nms_topk_workspace_bytes(2, 480, 640, 1000) >= 2 * 480 * 640 * 8
    assert lib.balf_packed_weights_bytes(0) >= 1280728 * 4
    # host-side validation happens before anything touches a device
    assert lib.balf_forward(None, 0, None, 1, 64, 64, None, None, None, 0, None) == -1
    assert lib.balf_nms_topk(None, 1, 64, 64, 0, 0, 64, 64, 15, 15, 10, None, None, None, None, 0, None) == -1
    assert lib.balf_window_nms(None, 1, 8, 8, 0, 3, None, None) == -1


def pack(lib, sd):
    n = lib.balf_num_state_tensors()
    host = [sd[lib.balf_state_tensor_name(i).decode()].contiguous() for i in range(n)]
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in host])
    nbytes = lib.balf_packed_weights_bytes(0)
    blob = np.zeros(nbytes // 4, np.float32)
    assert lib.balf_pack_weights(ptrs, n, 0, blob.ctypes.data, nbytes) == 0
    return blob


def test_pack_weights_fragment_order(lib):
    """Every weight element must appear in the blob where layout.h says (A-fragment order)."""
    sd = synth.synthetic_state_dict(5)
    blob = pack(lib, sd)

    def frags(wt, npad=None):
        w = wt.numpy()
        n, k = w.shape
        npad = npad or n
        wp = np.zeros((npad, k), np.float32); wp[:n] = w
        lane = np.arange(64)
        out = np.empty((npad // 16, k // 16, 64, 4), np.float32)
        for nt in range(npad // 16):
            for kt in range(k // 16):
                for j in range(4):
                    out[nt, kt, :, j] = wp[16 * nt + (lane & 15), 16 * kt + 4 * (lane >> 4) + j]
        return out.ravel()

    def find(sub):
        """offset of `sub` in the blob (weights are random, so the match is unique)"""
        first = np.flatnonzero(blob == sub[0])
        for o in first:
            if o + sub.size <= blob.size and np.array_equal(blob[o:o + sub.size], sub):
                return int(o)
        return -1

    rsh = "residual_split_head_multi_axis_gmlp_layer"
    offs = []
    for name, npad in [("down1.conv.0.weight", None), (f"down2.{rsh}.dense1.weight", None),
                       (f"down3.{rsh}.grid_gmlp_layer.grid_gating_unit.dense.weight", None),
                       (f"down4.{rsh}.dense2.weight", None), ("down4.conv2.weight", None),
                       ("detector_head.dense.weight", 80)]:
        t = sd[name]
        if name.endswith(f"{rsh}.dense1.weight"):      # LayerNorm gamma folded into the consuming Linear
            t = t * sd[name.replace("dense1.weight", "norm.weight")][None, :]
        sub = t.numpy().ravel() if name == "down1.conv.0.weight" else frags(t, npad)
        o = find(sub)
        assert o >= 0 and o % 64 == 0, name
        offs.append(o)
    assert offs == sorted(offs)                                        # blob follows state-dict order
    # BatchNorm folded to alpha/beta at the end of the blob
    g, b = sd["detector_head.norm.weight"].double(), sd["detector_head.norm.bias"].double()
    m, v = sd["detector_head.norm.running_mean"].double(), sd["detector_head.norm.running_var"].double()
    alpha = (g / torch.sqrt(v + 1e-5)).float().numpy()
    beta = (b - m * g / torch.sqrt(v + 1e-5)).float().numpy()
    assert find(alpha) > offs[-1] and find(beta) > find(alpha)


def test_pack_rejects_bad_arguments(lib):
    assert lib.balf_pack_weights(None, 166, 0, None, 0) == -1
    sd = synth.synthetic_state_dict(5)
    n = 166
    host = [sd[lib.balf_state_tensor_name(i).decode()].contiguous() for i in range(n)]
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in host])
    small = np.zeros(16, np.float32)
    assert lib.balf_pack_weights(ptrs, n, 0, small.ctypes.data, small.nbytes) == -3
    assert lib.balf_pack_weights(ptrs, n - 1, 0, small.ctypes.data, small.nbytes) == -1


def test_gelu_chord_tables(lib):
    """The two GELU tables at the end of the blob (layout.h: kGeluLutN / kGeluLogM), evaluated with the kernels' index
    arithmetic restated in float32 NumPy (stage1_f16.h: gelu_lut_off, stage_cs_f16.h: gelu_log_n), against the erf form
    in float64: nn.GELU() of /root/reference/balf/model/mlp_ma_decoder.py:52,99,126."""
    from scipy.special import erf
    blob = pack(lib, synth.synthetic_state_dict(5))
    u8 = (np.arange(256, dtype=np.float64) / 255.0).astype(np.float32)
    o = next(int(i) for i in np.flatnonzero(blob == u8[1]) if np.array_equal(blob[i - 1:i + 255], u8)) - 1
    n_uni, m_log = 3072, 256
    uni = blob[o + 256:o + 256 + 2 * (n_uni + 1)].reshape(-1, 2)
    o_log = o + 256 + (2 * (n_uni + 1) + 63) // 64 * 64
    log = blob[o_log:o_log + 2 * (3 * m_log + 1)].reshape(-1, 2)
    assert o_log + 2 * (3 * m_log + 1) <= blob.size
    x = np.concatenate([np.linspace(-9, 9, 400001), [0.0, -0.0, 6.0, -6.0, 7.0, -7.0, 5.99999, 30.0, -30.0, 1e-30]]).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    f32 = np.float32
    # stage 1: y = clamp01(x / 12 + 1/2); t = y * 8N + 1.5 * 2^23; byte offset = bits(t) & 0x7FF8; gelu = a + b x
    y = np.clip(x * f32(0.5 / 6.0) + f32(0.5), f32(0), f32(1)).astype(np.float32)
    t = (y * f32(8.0 * n_uni) + f32(12582912.0)).astype(np.float32)
    idx = (t.view(np.uint32) & np.uint32(0x7FF8)) >> 3
    assert idx.max() == n_uni and idx.min() == 0
    got = uni[idx, 0].astype(np.float32) + uni[idx, 1].astype(np.float32) * x
    assert np.abs(got - ref).max() < 1.2e-6
    # stages 2-4 (grid kernels): s = clamp01((|x| + 1) / 8); byte offset = (bits(s) >> 12) & 0x1FF8; gelu = x / 2 + (A + B |x|)
    s = np.clip(np.abs(x) * f32(0.125) + f32(0.125), f32(0), f32(1)).astype(np.float32)
    idx = ((s.view(np.uint32) >> 12) & np.uint32(0x1FF8)) >> 3
    assert idx.max() == 3 * m_log and idx.min() == 0
    e = log[idx, 1].astype(np.float32) * np.abs(x) + log[idx, 0].astype(np.float32)
    got = f32(0.5) * x + e.astype(np.float32)
    assert np.abs(got - ref).max() < 1.2e-6
    # stage-2 block kernel (stage2_f16.h): the uniform table with 2048 intervals, same index arithmetic
    n2 = 2048
    o2 = o_log + (2 * (3 * m_log + 1) + 63) // 64 * 64
    uni2 = blob[o2:o2 + 2 * (n2 + 1)].reshape(-1, 2)
    assert o2 + 2 * (n2 + 1) <= blob.size
    t = (y * f32(8.0 * n2) + f32(12582912.0)).astype(np.float32)
    idx = (t.view(np.uint32) & np.uint32(0x7FF8)) >> 3
    assert idx.max() == n2 and idx.min() == 0
    got = uni2[idx, 0].astype(np.float32) + uni2[idx, 1].astype(np.float32) * x
    assert np.abs(got - ref).max() < 2.4e-6
