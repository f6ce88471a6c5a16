"""Drop-in ``fd_burgers`` module: ``from fd_burgers import FDBurgers`` (reference FD/fd_burgers.py).

Mirrors ``FDBurgers(a, b, N).fom_burgers_newton(dt, n_steps, U0, mu1, mu2, max_iter=30, tol=1e-8,
use_fd_jacobian=False)`` (:59) on the MI355X through ``bg_fd_run``; ``mu1`` / ``mu2`` may be arrays
of B samples (result ``(B, N, n_steps+1)``).  ``use_fd_jacobian=True`` (the reference's debugging
aid, :46-57) makes the Jacobian dense and runs as batched library calls on the device.
"""
from __future__ import annotations

import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from burgers_hip import fom as _fom          # noqa: E402
from burgers_hip import lib as _lib          # noqa: E402

__all__ = ["FDBurgers"]


class FDBurgers:
    def __init__(self, a, b, N):
        self.a, self.b, self.N = a, b, N
        self.dx = (b - a) / (N - 1)
        self.x = np.linspace(a, b, N)
        self.last_iters = None
        self.last_flags = None

    def compute_source_term(self, mu2):
        return 0.02 * np.exp(mu2 * self.x)

    def fom_burgers_newton(self, dt, n_steps, U0, mu1, mu2, max_iter=30, tol=1e-8, use_fd_jacobian=False):
        batched = np.ndim(mu1) > 0 or np.ndim(mu2) > 0 or np.ndim(U0) > 1
        res = _fom.fd_run(self.a, self.b, self.N, np.asarray(U0, dtype=np.float64), mu1, mu2, dt, int(n_steps),
                          max_iter=max_iter, tol=tol, use_fd_jacobian=bool(use_fd_jacobian))
        U = _lib.to_host(res.snapshots())
        self.last_iters = res.iters.cpu().numpy()
        self.last_flags = res.flags.cpu().numpy()
        if not batched:
            self.last_iters, self.last_flags = self.last_iters[0], int(self.last_flags[0])
            return np.ascontiguousarray(U[0])
        return np.ascontiguousarray(U)
