"""``MLP_MA_DECODER``: the reference's detector as a parameter container whose forward runs on the
HIP library.

Mirrors /root/reference/balf/model/mlp_ma_decoder.py:246-285 at the interface level only: the module
tree is built so that ``state_dict()`` has exactly the reference's 167 names/shapes/dtypes (the
contract ``get_model.load_test_pretrained_model`` filters on, get_model.py:60-67, :84), and
``forward(x)`` returns ``{'logits', 'prob'}`` like decoder.py:30.  No layer here has a Python
``forward`` of its own: the whole network is one call into ``balf_forward`` (include/balf_hip.h).
There is no CPU path; a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from operator import attrgetter, is_ as _IS

import torch
import torch.nn as nn

from .. import _lib, arch, ops
from .._lib import BalfHipError, check, lib

# The packed-weight cache must notice every way a checkpoint can change under it, and a single-image call (the
# reference's only calling pattern, /root/reference/demo/demo_match.py:29) must not pay for that: walking the 167 state
# tensors through nn.Module.__getattr__ cost 130 us per forward.  So the module keeps a LIST of its state tensors and the
# dict slots they came from, and the per-call key is built from that list alone:
#   (i)   identity of every slot's current content (`m._parameters[name] is cached`): catches what replaces a Parameter or
#         buffer object without telling anyone -- `_apply` on a SUBMODULE (model.down1.float(), model.detector_head.to(..):
#         buffers are re-bound in the dict directly), parameter assignment, load_state_dict(assign=True);
#   (ii)  the sum of the tensors' version counters: in-place updates (load_state_dict, copy_, optimiser steps);
#   (iii) the tuple of their data pointers: `p.data = other` (EMA swaps, torch.nn.utils.vector_to_parameters, `_apply` on
#         parameters), which changes neither the object nor -- reliably -- its version.
# A module-level epoch, bumped by torch's global registration hooks, only says WHEN the list may have gained or lost an
# entry (register_parameter / register_buffer anywhere in the process); a rebuilt list that holds the same objects keeps the
# cache (building another model must not repack this one).
_EPOCH = [0]


def _bump_epoch(*_args, **_kwargs):
    _EPOCH[0] += 1


nn.modules.module.register_module_parameter_registration_hook(_bump_epoch)
nn.modules.module.register_module_buffer_registration_hook(_bump_epoch)
_VERSION = attrgetter("_version")
_DATA_PTR = torch.Tensor.data_ptr
_GETITEM = dict.get                      # (a deleted slot reads None: "changed")


class _Holder(nn.Module):
    """Plain container; exists so that parameter names nest like the reference's modules."""


def _linear(i, o):
    return nn.Linear(i, o)


def _gmlp(c, factor, tokens, unit_name):
    m = _Holder()
    m.norm = nn.LayerNorm(c)
    m.dense1 = _linear(c, c * factor)
    unit = _Holder()
    unit.norm = nn.LayerNorm(c)
    unit.dense = _linear(tokens, tokens)
    setattr(m, unit_name, unit)
    m.dense2 = _linear(c, c)
    return m


def _down(cin, c, a):
    d = _Holder()
    d.conv = nn.Sequential(_linear(cin, c), nn.ReLU(inplace=True))
    r = _Holder()
    r.norm = nn.LayerNorm(c)
    r.dense1 = _linear(c, c * a["input_proj_factor"])
    r.grid_gmlp_layer = _gmlp(c, a["grid_gmlp_factor"], a["grid_size"][0] * a["grid_size"][1], "grid_gating_unit")
    r.block_gmlp_layer = _gmlp(c, a["block_gmlp_factor"], a["block_size"][0] * a["block_size"][1],
                               "block_gating_unit")
    r.dense2 = _linear(c * a["input_proj_factor"], c)
    d.residual_split_head_multi_axis_gmlp_layer = r
    k = _Holder()
    k.norm = nn.LayerNorm(c)
    k.conv1 = _linear(c, c)
    k.conv2 = _linear(c, c)
    ca = _Holder()
    red = a["channels_reduction"]
    ca.excite = nn.Sequential(_linear(c, c // red), nn.ReLU(inplace=True), _linear(c // red, c), nn.Sigmoid())
    k.calayer = ca
    d.residual_channel_attention_block = k
    d.conv2 = _linear(c, c)
    return d


_GUARD_SLOTS = 8


class _PendingCall:
    """A guarded split-f16 forward that has not been looked at yet.  Holds NO tensor: scalars, an event recorded behind the
    call and weak references to its input and outputs, so that a batch the caller has dropped is freed at once (ADVICE r5:
    the closure kept here before pinned ~1.3 GB at 32 x 1080p until the next forward)."""
    __slots__ = ("ev", "wkey", "slot", "seq", "kind", "dims", "refs")

    def __init__(self, ev, wkey, slot, seq, kind, dims, refs):
        self.ev, self.wkey, self.slot, self.seq, self.kind, self.dims, self.refs = ev, wkey, slot, seq, kind, dims, refs


class _Fp16Guard:
    """The status blocks of the split-f16 forwards on one device (include/balf_hip.h: balf_forward_status): a RING of
    _GUARD_SLOTS blocks of four ints in PINNED HOST memory.  Every guarded call gets a block of its own, so that a word raised by
    call A is attributed to A however far the host has run ahead (with one shared block a flag of A was charged to whatever call
    happened to be pending when the host looked); the kernels store into it when a softmax denominator is not finite or
    a stage output reaches the largest f16, the host reads it without a copy.  `pending`: the calls not yet looked at, oldest
    first."""

    def __init__(self, device):
        import collections
        with torch.inference_mode(False):   # (a NORMAL tensor even when the first forward runs under torch.inference_mode(): an
            # inference tensor could not be cleared from outside that mode later)
            self.words = torch.zeros(_GUARD_SLOTS * _lib.STATUS_WORDS, dtype=torch.int32).pin_memory()
        self.device = device
        self.pending = collections.deque()
        self.seq = 0

    def ptr(self, slot: int) -> int:
        return self.words.data_ptr() + slot * _lib.STATUS_WORDS * 4

    def read_and_clear(self, slot: int):
        v = self.words[slot * _lib.STATUS_WORDS:(slot + 1) * _lib.STATUS_WORDS]
        w = v.tolist()
        if any(w):
            v.zero_()
        return w


def _guard_mode() -> str:
    """BALF_FP16_GUARD = lazy (default) | sync | off.  lazy: the status block of a split-f16 forward is looked at when the
    NEXT forward of the module starts (or in fp16_guard_check()), never with a synchronisation on the hot path; sync: before
    the forward returns (one stream synchronisation per call, the flagged batch is re-run before anyone sees it); off: the
    kernels get no status block."""
    m = os.environ.get("BALF_FP16_GUARD", "lazy")
    if m not in ("lazy", "sync", "off"):
        raise ValueError(f"BALF_FP16_GUARD must be lazy, sync or off, got {m!r}")
    return m


class MLP_MA_DECODER(nn.Module):
    def __init__(self, model_cfg, precision: str = "fp16"):
        super().__init__()
        a = {k: model_cfg[k] for k in ("en_embed_dims", "grid_size", "block_size", "grid_gmlp_factor",
                                       "block_gmlp_factor", "input_proj_factor", "channels_reduction",
                                       "cell_size")}          # KeyError on a missing key, like the reference
        arch.check_supported(a)
        dims = a["en_embed_dims"]
        for s in range(4):
            setattr(self, f"down{s + 1}", _down(dims[s], dims[s + 1], a))
        head = _Holder()
        head.dense = _linear(dims[4], a["cell_size"] ** 2 + 1)
        head.norm = nn.BatchNorm2d(a["cell_size"] ** 2 + 1)
        self.detector_head = head
        self.precision = precision     # the REQUESTED precision; see effective_precision for what the last forward ran
        self._tensors = None           # (epoch, [tensors in state_dict order], their dicts, their names), see _state_tensors
        self._gen = 0                  # bumped whenever the tensor list holds a different object
        self._packed = {}              # precision -> (weights key, device blob)
        self._fp16_verdict = None      # (weights key, precision the split-f16 request resolves to for these weights)
        self._effective = None
        self._guard = {}               # device -> _Fp16Guard (status block of the split-f16 forwards, see _guarded_call)

    # ---- weights -> packed device blob -------------------------------------------------------
    @staticmethod
    def _code_of(precision) -> int:
        try:
            return {"fp32": _lib.PREC_FP32, "fp16": _lib.PREC_FP16}[precision]
        except KeyError:
            raise ValueError(f"precision must be 'fp32' or 'fp16', got {precision!r}")

    def _precision_code(self) -> int:
        return self._code_of(self.precision)

    @property
    def effective_precision(self) -> str:
        """What the last forward actually ran: ``precision``, or 'fp32' after the split-f16 range check failed for the
        weights loaded at that time (re-evaluated whenever the weights change)."""
        return self._effective or self.precision

    def _apply(self, fn, *args, **kwargs):
        _bump_epoch()                  # (the slot check below notices this too; kept so that a move is never missed)
        return super()._apply(fn, *args, **kwargs)

    def _state_tensors(self):
        """The 167 state tensors in state_dict order.  The list is re-walked when a parameter or buffer was registered
        somewhere since (module-level epoch) or when a slot no longer holds the cached object; the generation counter --
        part of the cache key -- moves only if the walk found a different object."""
        c = self._tensors
        if c is not None and c[0] == _EPOCH[0] and all(map(_IS, map(_GETITEM, c[2], c[3]), c[1])):
            return c[1]
        dicts, names = [], []
        for m in self.modules():
            for n in m._parameters:
                dicts.append(m._parameters)
                names.append(n)
            for n in m._buffers:
                if n not in m._non_persistent_buffers_set:
                    dicts.append(m._buffers)
                    names.append(n)
        ts = list(map(_GETITEM, dicts, names))
        if c is None or len(ts) != len(c[1]) or not all(map(_IS, ts, c[1])):
            self._gen += 1
        self._tensors = (_EPOCH[0], ts, dicts, names)
        return ts

    def _weights_key(self, device):
        ts = self._state_tensors()
        try:
            ver = sum(map(_VERSION, ts))
        except RuntimeError:           # inference tensors (a model moved under torch.inference_mode()) track no versions
            ver = -1                   # ... and cannot be modified in place; replacing them shows in the pointers
        return (self._gen, str(device), ver, tuple(map(_DATA_PTR, ts)))

    def _state_key(self, device):
        return self._weights_key(device) + (self.precision,)

    def invalidate_packed(self) -> None:
        """Drop the packed blobs and the split-f16 range verdict: the next forward re-packs the weights as they are NOW.

        The cache key (``_weights_key``) notices in-place updates of the parameters themselves, ``load_state_dict``,
        replaced parameters, ``p.data = other`` swaps and dtype / device moves.  It CANNOT see an in-place edit made
        through ``p.data`` -- ``p.data.mul_(2)``, ``p.data.copy_(w)``, ``p.data.clamp_()`` -- because ``.data`` is an
        alias with a version counter of its own (tests/test_host_api.py pins this).  Code that edits weights that way
        (some EMA and weight-clipping helpers do) must call this method afterwards, or use ``with torch.no_grad():
        p.mul_(2)``, which the key does see."""
        self._packed = {}
        self._fp16_verdict = None
        self._effective = None

    def packed_weights(self, device, precision=None, _wkey=None) -> torch.Tensor:
        """The packed blob of the current weights for ``precision`` (default: the requested one) on ``device``."""
        precision = precision or self.precision
        prec = self._code_of(precision)
        wkey = _wkey or self._weights_key(device)
        hit = self._packed.get(precision)
        if hit is not None and hit[0] == wkey:
            return hit[1]
        l = lib()
        sd = self.state_dict()
        n = l.balf_num_state_tensors()
        host = []
        for i in range(n):
            name = l.balf_state_tensor_name(i).decode()
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if t.numel() != l.balf_state_tensor_numel(i):
                raise BalfHipError(f"{name}: {t.numel()} elements, library expects {l.balf_state_tensor_numel(i)}")
            host.append(t)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in host])
        nbytes = l.balf_packed_weights_bytes(prec)
        if nbytes == 0:
            raise BalfHipError(f"precision {precision!r} is not available in this build of libbalf_hip.so")
        blob = torch.empty(nbytes, dtype=torch.uint8)
        check(l.balf_pack_weights(ptrs, n, prec, blob.data_ptr(), nbytes), "balf_pack_weights")
        blob = blob.to(device)
        self._packed[precision] = (wkey, blob)
        return blob

    def _resolve(self, device):
        """-> (precision to run, its blob).  Once per set of weights, when the split-f16 path is requested: the default
        path carries every MFMA operand as two f16 halves, and an activation or weight beyond +-6.5e4 turns into inf/NaN
        without any trap (split16.h).  A checkpoint is therefore tried on three small images (noise, black, white) against
        the exact-fp32 kernels; if the score maps disagree or are not finite THESE WEIGHTS run on the fp32 kernels (still
        the HIP library, ~2.5x slower) and the module says so -- or raises with BALF_FP16_STRICT=1.  The requested
        precision is left alone: the next checkpoint is judged afresh."""
        g = self._guard.get(device)
        if g is not None and g.pending and not getattr(self, "_validating", False) and not torch.cuda.is_current_stream_capturing():
            self._guard_look(g, final=False)                     # lazy look at the finished calls' status blocks (no wait;
            #                                                      not under graph capture: an event query would invalidate it)
        wkey = self._weights_key(device)
        prec = self.precision
        if prec == "fp16" and not getattr(self, "_validating", False) and self._fp16_verdict is not None \
                and self._fp16_verdict[0] == wkey and self._fp16_verdict[1] == "fp32" \
                and os.environ.get("BALF_FP16_CHECK", "1") == "0":
            prec = "fp32"                                        # (a guard verdict holds even where the load-time probes are off)
        elif prec == "fp16" and not getattr(self, "_validating", False) and os.environ.get("BALF_FP16_CHECK", "1") != "0":
            v = self._fp16_verdict
            if v is None or v[0] != wkey:
                self._fp16_verdict = v = (wkey, self._check_fp16_range(device))
            prec = v[1]
        self._code_of(prec)
        self._effective = prec
        self._last_wkey = wkey           # (the guard below tags the call with it: one key computation per forward)
        return prec, self.packed_weights(device, prec, wkey)

    def _check_fp16_range(self, device) -> str:
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.stack([torch.rand((3, 128, 128), generator=g), torch.zeros((3, 128, 128)), torch.ones((3, 128, 128))])
        was_training = self.training
        try:
            self.training = False
            self.validate_fp16(x.to(device))
            return "fp16"
        except BalfHipError as e:
            if os.environ.get("BALF_FP16_STRICT") == "1":
                raise
            warnings.warn(f"balf_amd: this checkpoint is outside the range of the split-f16 path ({e}); "
                          "running it with the exact-fp32 MFMA kernels (effective_precision='fp32')", RuntimeWarning)
            return "fp32"
        finally:
            self.training = was_training

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, want_logits: bool = True):
        if self.training:
            raise BalfHipError("balf_amd implements the inference path only: call .eval() first "
                               "(training, /root/reference/balf/utils/train_utils.py:79-160, is out of scope)")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [B,3,H,W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise BalfHipError("balf_amd has no CPU path: move the model and the input to the GPU")
        b, _, h, w = x.shape
        if h % 64 or w % 64:
            raise ValueError(f"H and W must be multiples of 64 (pad with mod_padding_symmetric), got {h}x{w}")
        x = x.contiguous().float()
        dev = x.device
        _lib.require_mi355x(dev)
        l = lib()
        prec, blob = self._resolve(dev)
        prob = torch.empty((b, h, w), dtype=torch.float32, device=dev)
        logits = torch.empty((b, 65, h // 8, w // 8), dtype=torch.float32, device=dev) if want_logits else None
        nbytes = l.balf_forward_workspace_bytes(b, h, w)

        self._guarded_call(dev, prec, blob, "f32", (b, h, w, nbytes), (x, prob, logits))
        return {"logits": logits, "prob": prob}

    def _launch(self, kind, dims, blob, prec, status, src, prob, logits):
        """One balf_forward_status / balf_forward_u8_status on the current stream of prob's device (``status``: pointer or None)."""
        l = lib()
        dev = prob.device
        ws = ops._workspace("forward", dev, dims[-1])
        lg = logits.data_ptr() if logits is not None else None
        with torch.cuda.device(dev):
            if kind == "f32":
                b, h, w, _ = dims
                check(l.balf_forward_status(blob.data_ptr(), self._code_of(prec), src.data_ptr(), b, h, w, lg, prob.data_ptr(),
                                            ws.data_ptr(), ws.numel(), status, _lib.current_stream_ptr(dev)), "balf_forward")
            else:
                ch, b, h, w, _ = dims
                check(l.balf_forward_u8_status(blob.data_ptr(), self._code_of(prec), src.data_ptr(), ch, b, h, w, lg,
                                               prob.data_ptr(), ws.data_ptr(), ws.numel(), status,
                                               _lib.current_stream_ptr(dev)), "balf_forward_u8")

    # ---- the split-f16 path on inputs nobody has seen ----------------------
    def _guarded_call(self, dev, prec, blob, kind, dims, tensors):
        """Run one forward (``_launch(kind, dims, ...)`` on ``tensors`` = (input, prob, logits or None)).  On the split-f16
        path the kernels get a status block of the call's own (a ring of _GUARD_SLOTS); what it says is acted on according to
        BALF_FP16_GUARD (see _guard_mode): a flagged batch is re-run on the exact-fp32 kernels INTO THE SAME OUTPUT TENSORS, the
        module warns (BALF_FP16_STRICT=1: raises) and these weights run on the fp32 kernels from then on (effective_precision
        says so).  ``sync``: that happens before the forward returns -- nobody sees a flagged value.  ``lazy`` (default): the
        look happens when a later forward starts (or in fp16_guard_check()), never with a wait on the hot path; the repair is
        stream-ordered, so GPU work enqueued AFTER it sees the repaired tensors, but what was enqueued between the flagged call
        and the look (the NMS of pipeline.detect_batch, a copy to the host) has used the split path's values: the warning names
        the call and says how many later ones had been enqueued -- repeat those, or run with BALF_FP16_GUARD=sync where that
        matters.  Only weak references are kept: outputs the caller has dropped are not repaired (and not kept alive)."""
        mode = _guard_mode()
        if prec != "fp16" or mode == "off" or getattr(self, "_validating", False) or torch.cuda.is_current_stream_capturing():
            # (under hipGraph capture -- torch.cuda.graph -- the call is recorded without a status block: events of a capture cannot
            # be queried and a replay has no host side that could look; validate such a pipeline once with validate_fp16)
            self._launch(kind, dims, blob, prec, None, *tensors)
            return
        g = self._guard.get(dev)
        if g is None:
            g = self._guard[dev] = _Fp16Guard(dev)
        if len(g.pending) >= _GUARD_SLOTS:                       # the host is a whole ring ahead: the oldest call's block is
            g.pending[0].ev.synchronize()                        # needed again -- wait for that call (only) and look at it
            self._guard_look(g, final=False)
        slot = g.seq % _GUARD_SLOTS
        g.seq += 1
        self._launch(kind, dims, blob, prec, g.ptr(slot), *tensors)
        import weakref
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        refs = tuple(weakref.ref(t) if t is not None else None for t in tensors)
        g.pending.append(_PendingCall(ev, self._last_wkey, slot, g.seq, kind, dims, refs))
        if mode == "sync":
            ev.synchronize()
            self._guard_look(g, final=True)

    def _guard_look(self, g, final: bool):
        """Look at the status blocks of the pending calls the stream has passed, oldest first (``final``: it has passed all)."""
        while g.pending:
            p = g.pending[0]
            if not final and not p.ev.query():
                return                  # still running: its block is its own, a later look sees it
            g.pending.popleft()
            words = g.read_and_clear(p.slot)
            if any(words):
                self._guard_act(g, p, words, later=len(g.pending))

    def _guard_act(self, g, p, words, later: int):
        # Any word is a reason to leave the split path: beyond 65504 the high half saturates and the low half alone (11 bits)
        # carries the excess -- ~3e-4 relative instead of 2^-20 --, beyond ~1.3e5 the products turn into inf / NaN.
        msg = (f"an operand of the split-f16 path left the range of its f16 halves on a caller's input (guarded call #{p.seq}, "
               f"status words score={words[0]} range={words[1]} se={words[2]}: non-finite score map / stage output >= 65504 / "
               "non-finite squeeze-excite)")
        if os.environ.get("BALF_FP16_STRICT") == "1":
            raise BalfHipError(msg + "; use precision='fp32' for this checkpoint")
        # these weights run on the exact-fp32 kernels from now on ...
        dev = g.device
        same_weights = self._weights_key(dev) == p.wkey
        if same_weights:
            self._fp16_verdict = (p.wkey, "fp32")
        src, prob, logits = (r() if r is not None else None for r in p.refs)
        repaired = False
        if src is not None and prob is not None and same_weights:
            # ... and the flagged batch is computed again into the tensors the caller still holds
            self._launch(p.kind, p.dims, self.packed_weights(dev, "fp32", p.wkey), "fp32", None, src, prob, logits)
            repaired = True
        del src, prob, logits
        self._effective = "fp32"
        how = ("; that batch was re-run on the exact-fp32 kernels into the same output tensors" if repaired
               else "; its outputs could not be repaired (input or outputs released)")
        if later:
            how += (f" after {later} later forward(s) had been enqueued -- GPU work enqueued between call #{p.seq} and now used "
                    "the split path's values: repeat it, or run with BALF_FP16_GUARD=sync")
        warnings.warn("balf_amd: " + msg + how + ", and this checkpoint runs on the fp32 kernels from now on "
                      "(effective_precision='fp32')", RuntimeWarning)

    def fp16_guard_check(self, synchronize: bool = True) -> bool:
        """Act on the status blocks of the split-f16 forwards so far now (see _guarded_call); ``synchronize`` waits for them.
        Returns True when a flag was found and acted on -- a caller that has already used that call's results (copied them to
        the host like pipeline.extract_detections and demo_match.detect, the reference's one-image-per-call pattern, or fed them
        to further kernels like pipeline.detect_batch) repeats its call then."""
        hit = False
        for g in self._guard.values():
            if g.pending and synchronize:
                g.pending[-1].ev.synchronize()
            before = self._fp16_verdict
            self._guard_look(g, final=synchronize)
            hit = hit or (self._fp16_verdict is not before and self._fp16_verdict is not None and self._fp16_verdict[1] == "fp32")
        return hit

    def stage_view(self, b: int, h: int, w: int, device=None):
        """Validation aid: after ``forward`` / ``forward_u8`` of a batch whose PADDED shape is ``[b,3,h,w]`` on the current
        stream, the activations that crossed the stage boundaries of that call, as fp32 NHWC tensors:
        ``[down1 out, down2 out, down3 out, x2 of down4 before its conv2]`` (balf_forward_stage_view, include/balf_hip.h).
        The reference exposes the same tensors to forward hooks on ``down1..down4``
        (/root/reference/balf/model/mlp_ma_decoder.py:278-285); apply ``down4.conv2`` to the last one to get ``down4``'s."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        l = lib()
        nbytes = l.balf_forward_workspace_bytes(b, h, w)
        ws = ops._workspace("forward", dev, nbytes)
        prec = self._code_of(self.effective_precision)
        outs = []
        with torch.cuda.device(dev):
            for s in range(1, 5):
                n = l.balf_forward_stage_view_numel(b, h, w, s)
                if n == 0:
                    raise ValueError(f"bad shape for stage_view: {(b, h, w)}")
                sh = min(s, 3)
                o = torch.empty((b, h >> sh, w >> sh, n // (b * (h >> sh) * (w >> sh))), dtype=torch.float32, device=dev)
                check(l.balf_forward_stage_view(prec, ws.data_ptr(), ws.numel(), b, h, w, s, o.data_ptr(),
                                                _lib.current_stream_ptr(dev)), "balf_forward_stage_view")
                outs.append(o)
        return outs

    def validate_fp16(self, x: torch.Tensor, tol: float = 1e-4) -> float:
        """Check the split-f16 path against the exact-fp32 path on ``x`` (a padded [B,3,H,W] batch): returns the
        max-abs score-map difference and raises if it exceeds ``tol`` or if the f16 path produced a non-finite value.
        Worth one call per new checkpoint: every MFMA operand is carried as two f16 halves, so an activation or
        weight beyond +-6.5e4 saturates (split16.h) -- LayerNorm keeps most operands O(1), but the stage inputs, the
        gated branch, the RCAB hidden layer and the head input scale with the checkpoint's weights."""
        keep = self.precision
        nested = getattr(self, "_validating", False)
        try:
            self._validating = True
            self.precision = "fp32"
            ref = self.forward(x, want_logits=False)["prob"]
            self.precision = "fp16"
            out = self.forward(x, want_logits=False)["prob"]
        finally:
            self.precision = keep
            self._validating = nested
        if not bool(torch.isfinite(out).all()):
            raise BalfHipError("the split-f16 path produced non-finite values on this input: an operand left the f16 "
                               "range (|v| < 6.5e4); use precision='fp32' for this checkpoint")
        err = float((out - ref).abs().max())
        if err > tol:
            raise BalfHipError(f"split-f16 and fp32 score maps differ by {err:.3e} (> {tol:g}) on this input")
        return err

    def forward_u8(self, images: torch.Tensor, want_logits: bool = True):
        """Raw uint8 images on the GPU -- gray ``[B,H,W]`` or RGB ``[B,H,W,3]`` -- straight into the network:
        ``/255``, ``make_shape_even`` and ``mod_padding_symmetric(64)`` (what ``demo_match.detect`` does on the
        host with NumPy, /root/reference/demo/demo_match.py:21-29) happen inside the first kernels.  Returns the
        same dict as ``forward`` at the padded size, bit-identical to ``forward`` on the host-prepared input."""
        if self.training:
            raise BalfHipError("balf_amd implements the inference path only: call .eval() first")
        if images.dtype != torch.uint8 or images.dim() not in (3, 4) or (images.dim() == 4 and images.shape[-1] != 3):
            raise ValueError("expected uint8 [B,H,W] (gray) or [B,H,W,3] (RGB)")
        if not images.is_cuda:
            raise BalfHipError("balf_amd has no CPU path: move the model and the input to the GPU")
        images = images.contiguous()
        b, h, w = images.shape[:3]
        ch = 1 if images.dim() == 3 else 3
        hp, wp, _, _ = arch.padded_hw(h, w)
        dev = images.device
        _lib.require_mi355x(dev)
        l = lib()
        prec, blob = self._resolve(dev)
        prob = torch.empty((b, hp, wp), dtype=torch.float32, device=dev)
        logits = torch.empty((b, 65, hp // 8, wp // 8), dtype=torch.float32, device=dev) if want_logits else None
        nbytes = l.balf_forward_workspace_bytes(b, hp, wp)

        self._guarded_call(dev, prec, blob, "u8", (ch, b, h, w, nbytes), (images, prob, logits))
        return {"logits": logits, "prob": prob}
