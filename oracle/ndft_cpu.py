"""ctypes wrapper around oracle/ndft_c.c (exact NDFT, OpenMP) -- TEST INFRASTRUCTURE ONLY.

Used by tests to cross-check ``oracle/ndft.py`` and by ``bench.py``'s
``cpu_baseline`` leg (kind "port": it is this repo's restatement of
``/root/reference/torch_nfft/ndft.py:5-44``, not the reference's own code).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libndft_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ndft_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        sig = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
               ctypes.c_int64, ctypes.c_void_p, ctypes.c_int]
        for f in (lib.ndft_oracle_adjoint, lib.ndft_oracle_forward):
            f.argtypes = sig
            f.restype = ctypes.c_int
        lib.ndft_oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def max_threads():
    return int(_load().ndft_oracle_max_threads())


def _ranges(batch, n):
    if batch is None:
        return [(0, np.arange(n))]
    batch = np.asarray(batch)
    return [(b, np.nonzero(batch == b)[0]) for b in range(int(batch.max()) + 1)]


def ndft_adjoint(x, pos, batch=None, N=16, nthreads=0):
    lib = _load()
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    x = np.asarray(x)
    n, d = pos.shape
    cols = x.shape[1:]
    xc = np.ascontiguousarray(x.reshape(n, -1).astype(np.complex128))
    C = xc.shape[1]
    parts = _ranges(batch, n)
    y = np.zeros((len(parts),) + (N,) * d + (C,), dtype=np.complex128)
    for b, sel in parts:
        pb = np.ascontiguousarray(pos[sel])
        xb = np.ascontiguousarray(xc[sel])
        rc = lib.ndft_oracle_adjoint(pb.ctypes.data, xb.ctypes.data, pb.shape[0], d, C, N,
                                     y[b].ctypes.data, nthreads)
        if rc:
            raise RuntimeError("ndft_oracle_adjoint failed: %d" % rc)
    return y.reshape((len(parts),) + (N,) * d + cols)


def ndft_forward(x, pos, batch=None, nthreads=0):
    lib = _load()
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    x = np.asarray(x)
    n, d = pos.shape
    B, N = x.shape[0], x.shape[1]
    cols = x.shape[1 + d:]
    xc = np.ascontiguousarray(x.reshape((B,) + (N,) * d + (-1,)).astype(np.complex128))
    C = xc.shape[-1]
    y = np.zeros((n, C), dtype=np.complex128)
    for b, sel in _ranges(batch, n):
        pb = np.ascontiguousarray(pos[sel])
        yb = np.zeros((pb.shape[0], C), dtype=np.complex128)
        rc = lib.ndft_oracle_forward(pb.ctypes.data, xc[b].ctypes.data, pb.shape[0], d, C, N,
                                     yb.ctypes.data, nthreads)
        if rc:
            raise RuntimeError("ndft_oracle_forward failed: %d" % rc)
        y[sel] = yb
    return y.reshape((n,) + cols)
