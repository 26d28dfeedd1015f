"""ctypes binding of libbalf_hip.so (include/balf_hip.h).

There is no CPU fallback anywhere in ``balf_amd``: if the shared library is missing or the
device is not an MI355X, the calls below raise.  The library is built in-tree by
``balf_amd/csrc/build.sh`` (plain ``hipcc --offload-arch=gfx950``, see ``__graft_entry__.build``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BALF_HIP_LIB", os.path.join(_HERE, "libbalf_hip.so"))   # override: tuning builds only

OK = 0
PREC_FP32, PREC_FP16 = 0, 1
MAX_NMS_SIZE, MAX_TOPK = 32, 16384
STATUS_SCORE, STATUS_RANGE, STATUS_SE, STATUS_WORDS = 0, 1, 2, 4      # include/balf_hip.h: balf_forward_status

# name -> (restype, argtypes); kept in step with include/balf_hip.h (tests/test_abi.py checks)
_vp, _i, _sz, _fp = C.c_void_p, C.c_int, C.c_size_t, C.c_void_p
PROTOTYPES = {
    "balf_abi_version": (_i, []),
    "balf_error_string": (C.c_char_p, [_i]),
    "balf_device_check": (_i, []),
    "balf_build_flags": (C.c_char_p, []),
    "balf_num_state_tensors": (_i, []),
    "balf_state_tensor_name": (C.c_char_p, [_i]),
    "balf_state_tensor_numel": (_sz, [_i]),
    "balf_packed_weights_bytes": (_sz, [_i]),
    "balf_pack_weights": (_i, [C.POINTER(_vp), _i, _i, _vp, _sz]),
    "balf_forward_workspace_bytes": (_sz, [_i, _i, _i]),
    "balf_forward_micro_batch": (_i, [_i, _i, _i]),
    "balf_forward": (_i, [_vp, _i, _fp, _i, _i, _i, _fp, _fp, _vp, _sz, _vp]),
    "balf_forward_u8": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _fp, _fp, _vp, _sz, _vp]),
    "balf_forward_status": (_i, [_vp, _i, _fp, _i, _i, _i, _fp, _fp, _vp, _sz, _vp, _vp]),
    "balf_forward_u8_status": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _fp, _fp, _vp, _sz, _vp, _vp]),
    "balf_forward_stage_view_numel": (_sz, [_i, _i, _i, _i]),
    "balf_forward_stage_view": (_i, [_i, _vp, _sz, _i, _i, _i, _i, _fp, _vp]),
    "balf_window_nms": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _vp]),
    "balf_nms_topk_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "balf_nms_topk": (_i, [_fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _fp, _vp, _vp, _sz, _vp]),
    "balf_nms_threshold": (_i, [_fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp, _fp, _vp, _vp, _sz, _vp]),
    "balf_greedy_nms_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "balf_greedy_nms": (_i, [_fp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _i, _i, _i, _vp, _fp, _fp, _vp, _vp, _vp,
                            _sz, _vp]),
    "balf_hardnet_num_state_tensors": (_i, []),
    "balf_hardnet_state_tensor_name": (C.c_char_p, [_i]),
    "balf_hardnet_state_tensor_numel": (_sz, [_i]),
    "balf_hardnet_packed_weights_bytes": (_sz, []),
    "balf_hardnet_pack_weights": (_i, [C.POINTER(_vp), _i, _vp, _sz]),
    "balf_hardnet_workspace_bytes": (_sz, [_i]),
    "balf_hardnet_forward": (_i, [_vp, _fp, _i, _fp, _vp, _sz, _vp]),
    "balf_hardnet_forward_masked": (_i, [_vp, _fp, _i, _i, _vp, _fp, _vp, _sz, _vp]),
    "balf_hardnet_forward_ex": (_i, [_vp, _fp, _i, _i, _vp, _i, _fp, _vp, _sz, _vp]),
    "balf_extract_patches_workspace_bytes": (_sz, [_i, _i, C.c_float]),
    "balf_extract_patches": (_i, [_vp, _i, _i, _fp, _i, C.c_float, _fp, _vp, _sz, _vp]),
    "balf_extract_patches_batch_workspace_bytes": (_sz, [_i, _i, _i, C.c_float]),
    "balf_extract_patches_batch": (_i, [_vp, _i, _i, _i, _fp, _vp, _i, C.c_float, _fp, _vp, _sz, _vp]),
    "balf_rgb_to_gray": (_i, [_vp, C.c_long, _vp, _vp]),
    "balf_match_smnn_workspace_bytes": (_sz, [_i, _i]),
    "balf_match_smnn": (_i, [_fp, _i, _fp, _i, C.c_float, _vp, _fp, _vp, _vp, _sz, _vp]),
    "balf_match_smnn_batch_workspace_bytes": (_sz, [_i, _i, _i]),
    "balf_match_smnn_batch": (_i, [_fp, _i, _vp, _fp, _i, _vp, _i, C.c_float, _vp, _fp, _vp, _vp, _sz, _vp]),
    "balf_repeatability_workspace_bytes": (_sz, [_i, _i, _i]),
    "balf_repeatability": (_i, [_vp, _i, _vp, _i, C.c_double, C.c_double, C.c_double, C.c_double, _i, _vp, _vp, _vp, _vp,
                               _vp, _sz, _vp]),
    "balf_apply_homography": (_i, [_vp, _i, _vp, _vp, _vp]),
    "balf_common_region_masks": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "balf_profile_num_slots": (_i, []),
    "balf_profile_slot_name": (C.c_char_p, [_i]),
    "balf_profile_begin": (_i, []),
    "balf_profile_end": (_i, [_vp, _vp]),
}


class BalfHipError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load the library once; fail loudly when it is absent (never fall back to a CPU path)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise BalfHipError(
                f"{LIB_PATH} not found: build it with balf_amd/csrc/build.sh (or __graft_entry__.build()); "
                "balf_amd has no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if l.balf_abi_version() != 1:
            raise BalfHipError("libbalf_hip.so ABI version mismatch")
        flags = l.balf_build_flags().decode()
        if not flags.startswith("release"):
            # a timing-ablation / instrumented build (csrc/diag.h) computes wrong results: only ever loaded on purpose
            if "BALF_HIP_LIB" not in os.environ:
                raise BalfHipError(f"{LIB_PATH} is a diagnostic build ({flags}): rebuild it with balf_amd/csrc/build.sh")
            import warnings
            warnings.warn(f"balf_amd: {LIB_PATH} is a DIAGNOSTIC build ({flags}); its results are not valid", RuntimeWarning)
        _lib = l
    return _lib


_checked_devices = set()


def require_mi355x(device) -> None:
    """Raise unless ``device`` is a gfx950 GPU (balf_device_check: the kernels are built for MI355X only).  Checked
    once per device, by every op wrapper before its first launch there."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _checked_devices:
        return
    with torch.cuda.device(idx):
        rc = lib().balf_device_check()
    if rc != OK:
        raise BalfHipError(f"cuda:{idx} is not an MI355X (gfx950): {lib().balf_error_string(rc).decode()} ({rc})")
    _checked_devices.add(idx)


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise BalfHipError(f"{what} failed: {lib().balf_error_string(rc).decode()} ({rc})")


def require_gpu_tensor(t, name: str) -> None:
    if not t.is_cuda:
        raise BalfHipError(f"{name} must live on the GPU: balf_amd has no CPU path (got device {t.device})")
    if not t.is_contiguous():
        raise BalfHipError(f"{name} must be contiguous")
    require_mi355x(t.device)


def current_stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def tensor_key(t):
    """(storage address, version) of a parameter/buffer, for the packed-weight caches.  Inference tensors
    (a model moved under torch.inference_mode()) do not track versions and cannot be modified in place."""
    try:
        return (t.data_ptr(), t._version)
    except RuntimeError:
        return (t.data_ptr(), -1)
