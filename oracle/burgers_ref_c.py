"""ctypes front end of the C oracle (``oracle/burgers_ref_c.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``burgers_ref_c.c``.  Used for
parity cases too large for the NumPy oracle and as the ``cpu_baseline`` leg of
``bench.py``.  ``build()`` compiles it with ``make -C oracle``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libburgers_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "burgers_ref_c.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        L.bo_fom_run.restype = ctypes.c_int
        L.bo_fom_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                 ctypes.c_int, dp, ip, ctypes.c_int]
        L.bo_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def fom_run(X, u0, mu1, mu2, dt, nsteps, E=0.0, tol=1e-6, max_it=20, supg=True, nthreads=0):
    """Batched FOM.  Returns ``hist (B, nsteps+1, N)`` (time-major) and ``iters (B, nsteps)``."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    mu1 = np.atleast_1d(np.asarray(mu1, dtype=np.float64)).copy()
    mu2 = np.atleast_1d(np.asarray(mu2, dtype=np.float64)).copy()
    B, n = len(mu1), len(X)
    u0 = np.ascontiguousarray(np.broadcast_to(np.asarray(u0, dtype=np.float64), (B, n)))
    hist = np.empty((B, nsteps + 1, n))
    iters = np.zeros((B, nsteps), dtype=np.int32)
    rc = lib().bo_fom_run(n, B, nsteps, _dp(X), _dp(u0), _dp(mu1), _dp(mu2), dt, E, tol, max_it,
                          1 if supg else 0, _dp(hist),
                          iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), nthreads)
    if rc != 0:
        raise RuntimeError(f"bo_fom_run failed: {rc}")
    return hist, iters


def max_threads():
    return lib().bo_max_threads()
