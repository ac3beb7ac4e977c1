"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper over oracle/map_oracle.cpp
(restatement of /root/reference/utils/calc_utils.py:8-39 with the libstdc++ tie order)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcmh_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "map_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        p, i64 = ctypes.c_void_p, ctypes.c_int64
        lib.oracle_hamming_row.argtypes = [p, p, i64, i64, p]
        lib.oracle_hamming_row.restype = None
        lib.oracle_sort_perm.argtypes = [p, i64, p]
        lib.oracle_sort_perm.restype = None
        lib.oracle_sort_perm_depth.argtypes = [p, i64, i64, p]
        lib.oracle_sort_perm_depth.restype = None
        lib.oracle_map_k.argtypes = [p, p, p, p, i64, i64, i64, i64, i64, ctypes.c_int, p, p]
        lib.oracle_map_k.restype = ctypes.c_float
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def hamming_row(q, rB):
    q, rB = _f32(q), _f32(rB)
    out = np.empty(rB.shape[0], np.float32)
    _load().oracle_hamming_row(q.ctypes.data, rB.ctypes.data, rB.shape[0], rB.shape[1], out.ctypes.data)
    return out


def sort_perm(keys):
    keys = _f32(keys)
    out = np.empty(keys.shape[0], np.int64)
    _load().oracle_sort_perm(keys.ctypes.data, keys.shape[0], out.ctypes.data)
    return out


def sort_perm_depth(keys, depth_limit):
    """std::sort's permutation with the introsort depth budget forced to `depth_limit`."""
    keys = _f32(keys)
    out = np.empty(keys.shape[0], np.int64)
    _load().oracle_sort_perm_depth(keys.ctypes.data, keys.shape[0], int(depth_limit), out.ctypes.data)
    return out


def map_k(qB, rB, qL, rL, k=None, stable=False, want_ind=False):
    """-> (mAP float32, ap[Q] float32, ind[Q,N] int64 or None)."""
    qB, rB, qL, rL = _f32(qB), _f32(rB), _f32(qL), _f32(rL)
    Q, K = qB.shape
    N, C = rL.shape
    ap = np.empty(Q, np.float32)
    ind = np.empty((Q, N), np.int64) if want_ind else None
    m = _load().oracle_map_k(qB.ctypes.data, rB.ctypes.data, qL.ctypes.data, rL.ctypes.data,
                             Q, N, K, C, 0 if k is None else int(k), int(bool(stable)),
                             ap.ctypes.data, None if ind is None else ind.ctypes.data)
    return np.float32(m), ap, ind
