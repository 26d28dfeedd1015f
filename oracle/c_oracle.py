"""ctypes binding of oracle/liboracle_nms.so (TEST INFRASTRUCTURE ONLY; see oracle/oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle_nms.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.oracle_nms_topk.restype = C.c_int
        _lib.oracle_nms_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    return _lib


def nms_topk(score: np.ndarray, border: int, size: int, k: int):
    """-> (idx int32 raster order, score fp32, dense nms map)"""
    s = np.ascontiguousarray(score, dtype=np.float32)
    h, w = s.shape
    nms = np.empty_like(s)
    idx = np.empty(k, np.int32)
    sc = np.empty(k, np.float32)
    n = lib().oracle_nms_topk(s.ctypes.data, h, w, border, size, k, nms.ctypes.data, idx.ctypes.data, sc.ctypes.data)
    if n < 0:
        raise IndexError("num_points exceeds the number of pixels (or bad arguments)")
    return idx[:n], sc[:n], nms
