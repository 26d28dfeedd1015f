"""Guard-band and poison tests of every caller-owned buffer of the detection path.

The library writes through hand-computed offsets into buffers the caller sizes from *_workspace_bytes() and from the
documented output shapes (include/balf_hip.h); the Python wrappers hand it a cached, possibly over-sized workspace, so an
overrun of a few KB -- or a read of a slot nobody wrote -- is invisible to the parity tests.  Here every buffer is carved out
of an allocation with 1 MiB of patterned guard on both sides and passed with EXACTLY the documented size, through the C ABI,
at every BASELINE shape and at batch sizes that end in a partial micro-batch (1, 9, 33: the micro-batch is 16 images at 1088x1920); and the forward is run on a
workspace full of NaN bit patterns against one full of zeros: the outputs must be bit-identical."""
import ctypes as C

import numpy as np
import pytest
import torch

from balf_amd import _lib, arch
from balf_amd.utils import synth
from tests.golden import cases

pytestmark = pytest.mark.gpu
GUARD = 1 << 20
PAT = 0xA5
DEV = "cuda:0"


class Guarded:
    """`nbytes` usable bytes between two guard bands; .ptr is what the library gets."""

    def __init__(self, nbytes, fill=None):
        self.n = int(nbytes)
        self.full = torch.full((2 * GUARD + self.n,), PAT, dtype=torch.uint8, device=DEV)
        if fill is not None:
            self.full[GUARD:GUARD + self.n] = fill
        self.ptr = self.full.data_ptr() + GUARD
        assert self.ptr % 256 == 0

    def view(self, dtype, shape):
        return self.full[GUARD:GUARD + self.n].view(dtype).view(shape)

    def intact(self):
        return bool((self.full[:GUARD] == PAT).all()) and bool((self.full[GUARD + self.n:] == PAT).all())


@pytest.fixture(scope="module")
def blobs():
    from balf_amd.model import get_model
    m = get_model.load_model(arch.DEFAULT_MODEL_CFG)
    m.load_state_dict(synth.synthetic_state_dict(cases.WEIGHT_SEED))
    m = m.eval().to(DEV)
    return {"fp32": (m.packed_weights(DEV, "fp32"), _lib.PREC_FP32), "fp16": (m.packed_weights(DEV, "fp16"), _lib.PREC_FP16)}


def _stream():
    return torch.cuda.current_stream(torch.device(DEV)).cuda_stream


SHAPES = [(480, 640), (720, 1280), (1080, 1920)]                         # BASELINE configs[1..4]


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
@pytest.mark.parametrize("h,w", SHAPES)
@pytest.mark.parametrize("b", [1, 9, 33])
def test_forward_stays_inside_its_buffers(blobs, precision, h, w, b):
    l = _lib.lib()
    blob, prec = blobs[precision]
    hp, wp, _, _ = arch.padded_hw(h, w)
    x = Guarded(b * 3 * hp * wp * 4, fill=0)
    x.view(torch.float32, (b, 3, hp, wp))[:] = torch.rand((1, 3, hp, wp), device=DEV)
    prob, logits = Guarded(b * hp * wp * 4), Guarded(b * 65 * (hp // 8) * (wp // 8) * 4)
    ws = Guarded(l.balf_forward_workspace_bytes(b, hp, wp))
    rc = l.balf_forward(blob.data_ptr(), prec, x.ptr, b, hp, wp, logits.ptr, prob.ptr, ws.ptr, ws.n, _stream())
    torch.cuda.synchronize()
    assert rc == 0
    for name, g in (("x", x), ("prob", prob), ("logits", logits), ("workspace", ws)):
        assert g.intact(), f"balf_forward wrote outside {name} ({precision}, {b} x {hp}x{wp})"
    p = prob.view(torch.float32, (b, hp, wp))
    assert bool(torch.isfinite(p).all()) and float(p.max()) <= 1.0 and float(p.min()) >= 0.0      # every pixel was written
    assert bool(torch.isfinite(logits.view(torch.float32, (b, 65, hp // 8, wp // 8))).all())
    # one byte short of the documented workspace is refused, not overrun
    assert l.balf_forward(blob.data_ptr(), prec, x.ptr, b, hp, wp, logits.ptr, prob.ptr, ws.ptr, ws.n - 1, _stream()) == -3


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
@pytest.mark.parametrize("h,w,ch", [(480, 640, 1), (721, 1279, 3), (1080, 1920, 1)])
@pytest.mark.parametrize("b", [1, 9, 33])
def test_forward_u8_stays_inside_its_buffers(blobs, precision, h, w, ch, b):
    l = _lib.lib()
    blob, prec = blobs[precision]
    hp, wp, _, _ = arch.padded_hw(h, w)
    img = Guarded(b * h * w * ch, fill=0)
    img.view(torch.uint8, (b, h, w, ch))[:] = torch.randint(0, 256, (1, h, w, ch), dtype=torch.uint8, device=DEV)
    prob, logits = Guarded(b * hp * wp * 4), Guarded(b * 65 * (hp // 8) * (wp // 8) * 4)
    ws = Guarded(l.balf_forward_workspace_bytes(b, hp, wp))
    rc = l.balf_forward_u8(blob.data_ptr(), prec, img.ptr, ch, b, h, w, logits.ptr, prob.ptr, ws.ptr, ws.n, _stream())
    torch.cuda.synchronize()
    assert rc == 0
    for name, g in (("image", img), ("prob", prob), ("logits", logits), ("workspace", ws)):
        assert g.intact(), f"balf_forward_u8 touched memory outside {name} ({precision}, {b} x {h}x{w}x{ch})"
    assert bool(torch.isfinite(prob.view(torch.float32, (b, hp, wp))).all())


@pytest.mark.parametrize("h,w,k", [(480, 640, 1000), (720, 1280, 2000), (1080, 1920, 2000), (37, 53, 100)])
@pytest.mark.parametrize("b", [1, 9, 33])
def test_nms_topk_and_greedy_stay_inside_their_buffers(h, w, k, b):
    l = _lib.lib()
    hp, wp, top, left = arch.padded_hw(h, w)
    g = torch.Generator(device="cpu").manual_seed(h + b)
    prob = Guarded(b * hp * wp * 4, fill=0)
    prob.view(torch.float32, (b, hp, wp))[:] = torch.rand((b, hp, wp), generator=g).to(DEV)
    idx, score, count = Guarded(b * k * 4), Guarded(b * k * 4), Guarded(b * 4)
    ws = Guarded(l.balf_nms_topk_workspace_bytes(b, h, w, k))
    rc = l.balf_nms_topk(prob.ptr, b, hp, wp, top, left, h, w, 15, 15, k, idx.ptr, score.ptr, count.ptr, ws.ptr, ws.n, _stream())
    torch.cuda.synchronize()
    assert rc == 0
    for name, gb in (("prob", prob), ("idx", idx), ("score", score), ("count", count), ("workspace", ws)):
        assert gb.intact(), f"balf_nms_topk wrote outside {name} ({b} x {h}x{w}, K = {k})"
    cnt = count.view(torch.int32, (b,))
    assert bool((cnt >= 0).all()) and bool((cnt <= k).all())
    ii = idx.view(torch.int32, (b, k))
    assert bool(((ii >= -1) & (ii < h * w)).all())
    # the demo path's greedy NMS (conf. threshold high enough for a sparse candidate set on a uniform-random map)
    xy, total = Guarded(b * k * 2 * 4), Guarded(b * 4)
    idx2, score2, count2 = Guarded(b * k * 4), Guarded(b * k * 4), Guarded(b * 4)
    ws2 = Guarded(l.balf_greedy_nms_workspace_bytes(b, h, w, k))
    rc = l.balf_greedy_nms(prob.ptr, b, hp, wp, top, left, h, w, 4, C.c_float(0.97), 8, k, 5, idx2.ptr, score2.ptr, xy.ptr,
                           count2.ptr, total.ptr, ws2.ptr, ws2.n, _stream())
    torch.cuda.synchronize()
    assert rc == 0
    for name, gb in (("prob", prob), ("idx", idx2), ("score", score2), ("xy", xy), ("count", count2), ("total", total),
                     ("workspace", ws2)):
        assert gb.intact(), f"balf_greedy_nms wrote outside {name} ({b} x {h}x{w}, K = {k})"


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
@pytest.mark.parametrize("b,hp,wp", [(2, 128, 192), (9, 512, 640), (17, 1088, 1920)])
def test_forward_does_not_read_stale_workspace(blobs, precision, b, hp, wp):
    """Every workspace slot is written before it is read: a workspace full of 0xFF bytes (NaN as fp32 and as f16) gives the
    bits a zeroed one gives.  17 x 1088x1920 is two micro-batches: the second one runs on the first one's leftovers too."""
    l = _lib.lib()
    blob, prec = blobs[precision]
    x = torch.rand((b, 3, hp, wp), device=DEV)
    n = l.balf_forward_workspace_bytes(b, hp, wp)
    outs = []
    for fill in (0x00, 0xFF):
        ws = torch.full((n,), fill, dtype=torch.uint8, device=DEV)
        prob = torch.full((b, hp, wp), float("nan"), device=DEV)
        logits = torch.full((b, 65, hp // 8, wp // 8), float("nan"), device=DEV)
        assert l.balf_forward(blob.data_ptr(), prec, x.data_ptr(), b, hp, wp, logits.data_ptr(), prob.data_ptr(), ws.data_ptr(), n,
                              _stream()) == 0
        torch.cuda.synchronize()
        outs.append((prob, logits))
        del ws
    assert bool(torch.isfinite(outs[1][0]).all()) and bool(torch.isfinite(outs[1][1]).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_nms_topk_does_not_read_stale_workspace():
    l = _lib.lib()
    b, h, w, k = 3, 480, 640, 1000
    hp, wp, top, left = arch.padded_hw(h, w)
    prob = torch.rand((b, hp, wp), device=DEV)
    prob[1] = torch.round(prob[1] * 50) / 50                               # tie-heavy: the raster-first selection runs
    n = l.balf_nms_topk_workspace_bytes(b, h, w, k)
    res = []
    for fill in (0x00, 0xFF):
        ws = torch.full((n,), fill, dtype=torch.uint8, device=DEV)
        idx = torch.full((b, k), -7, dtype=torch.int32, device=DEV)
        score = torch.full((b, k), float("nan"), device=DEV)
        count = torch.full((b,), -7, dtype=torch.int32, device=DEV)
        assert l.balf_nms_topk(prob.data_ptr(), b, hp, wp, top, left, h, w, 15, 15, k, idx.data_ptr(), score.data_ptr(),
                               count.data_ptr(), ws.data_ptr(), n, _stream()) == 0
        torch.cuda.synchronize()
        res.append((idx, score, count))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)
    assert np.all(res[0][2].cpu().numpy() == k)
