"""ORACLE (test infrastructure only) -- ctypes loader for oracle/wpt_oracle.c."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libwpt_oracle.so")


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.wpt_oracle_nodes.restype = ctypes.c_long
    lib.wpt_oracle_nodes.argtypes = [dp, ctypes.c_long, ctypes.c_long, dp, ctypes.c_long,
                                     ctypes.c_int, dp]
    lib.wpt_oracle_features.restype = ctypes.c_long
    lib.wpt_oracle_features.argtypes = [dp, ctypes.c_long, ctypes.c_long, dp, ctypes.c_long,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, dp]
    return lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def wpt_nodes_c(x: np.ndarray, dec_lo, level: int) -> np.ndarray:
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    lo = np.ascontiguousarray(dec_lo, dtype=np.float64)
    b, n = x.shape
    t = lib.wpt_oracle_nodes(_dp(x), b, n, _dp(lo), len(lo), level, None)
    out = np.empty((b, 1 << level, t), dtype=np.float64)
    lib.wpt_oracle_nodes(_dp(x), b, n, _dp(lo), len(lo), level, _dp(out))
    return out


def packet_features_c(x, dec_lo, level, log_scale=False, loss_less=False, power=2.0):
    lib = _load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    lo = np.ascontiguousarray(dec_lo, dtype=np.float64)
    b, n = x.shape
    t = lib.wpt_oracle_nodes(_dp(x), b, n, _dp(lo), len(lo), level, None)
    c = 2 if (log_scale and loss_less) else 1
    out = np.empty((b, c, 1 << level, t), dtype=np.float64)
    lib.wpt_oracle_features(_dp(x), b, n, _dp(lo), len(lo), level, int(log_scale),
                            int(loss_less), float(power), _dp(out))
    return out
