"""TEST INFRASTRUCTURE ONLY - ctypes wrapper of oracle/rvq_ref.c (built by oracle/Makefile)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librvq_ref.so")
_lib = None


def build() -> str:
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "rvq_ref.c")):
        subprocess.run(["make", "-C", _HERE, "_build/librvq_ref.so"], check=True, capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def search_f32(x: np.ndarray, cb: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    Q, K, D = cb.shape
    idx = np.empty((x.shape[0], Q), np.int64)
    _load().rvq_search_f32(_p(x), C.c_int64(x.shape[0]), _p(cb), Q, K, D, _p(idx))
    return idx


def check_f64(x: np.ndarray, cb: np.ndarray, idx: np.ndarray):
    x = np.ascontiguousarray(x, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    Q, K, D = cb.shape
    n = x.shape[0]
    excess = np.empty((n, Q), np.float64)
    best = np.empty((n, Q), np.int64)
    gap = np.empty((n, Q), np.float64)
    _load().rvq_check_f64(_p(x), C.c_int64(n), _p(cb), Q, K, D, _p(idx), _p(excess), _p(best), _p(gap))
    return excess, best, gap


def lookup_f32(idx: np.ndarray, cb: np.ndarray) -> np.ndarray:
    cb = np.ascontiguousarray(cb, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    Q, K, D = cb.shape
    out = np.empty((idx.shape[0], D), np.float32)
    _load().rvq_lookup_f32(_p(idx), C.c_int64(idx.shape[0]), _p(cb), Q, K, D, _p(out))
    return out
