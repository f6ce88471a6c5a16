"""Drop-in ``fem_burgers`` module: ``from fem_burgers import FEMBurgers``.

Mirrors the call signatures of the reference's ``FEM/fem_burgers.py`` for the hot-path
methods (``fom_burgers`` :646, ``pod_prom_burgers`` :709, ``pod_quadratic_manifold``
:1081, ``pod_ann_prom`` :1177) so that the reference's driver scripts
(``FEM/paper_training_stage.py:48-49``, ``POD/Results_thesis/prom_pod.py:42,58``, ...) run
unmodified when this directory is what ``sys.path`` finds.  The work is done by the HIP
kernels in ``libburgers_hip.so`` through a ctypes C ABI; there is no CPU fallback.

Differences from the reference, all additive:
  * ``mu1`` / ``mu2`` may be arrays of B samples; the result is then ``(B, N, nT+1)``;
  * per-iteration console prints are off unless ``FEMBurgers.verbose`` is set;
  * ``last_iters`` / ``last_flags`` hold the Picard iteration counts and status bits of
    the latest call (the reference exposes them only through its prints).
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
from burgers_hip import rom as _rom          # noqa: E402

__all__ = ["FEMBurgers"]


class FEMBurgers:
    """P1 FEM Burgers solver on a uniform 1-D mesh, MI355X back end.

    ``X``: node coordinates ``(N,)``; ``T``: 1-based element connectivity ``(N-1, 2)`` as
    the reference builds it (``FEM/paper_training_stage.py:35-36``).
    """

    verbose = False

    def __init__(self, X, T):
        self.X = np.asarray(X, dtype=np.float64)
        self.T = np.asarray(T)
        n = len(self.X)
        if self.T.shape != (n - 1, 2) or not (
                np.array_equal(self.T[:, 0], np.arange(1, n)) and np.array_equal(self.T[:, 1], np.arange(2, n + 1))):
            raise NotImplementedError("connectivity must be the chain [[e, e+1]] (1-based) of 2-node elements")
        # tables kept for attribute compatibility with the reference (:317-322)
        self.ngaus = 2
        self.zgp = np.array([-np.sqrt(3) / 3, np.sqrt(3) / 3])
        self.wgp = np.array([1, 1])
        self.N = np.array([(1 - self.zgp) / 2, (1 + self.zgp) / 2]).T
        self.Nxi = np.array([[-1 / 2, 1 / 2], [-1 / 2, 1 / 2]])
        self.last_iters = None
        self.last_flags = None

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _batched(mu1, mu2, u0=None):
        return np.ndim(mu1) > 0 or np.ndim(mu2) > 0 or np.ndim(u0) > 1

    def _finish(self, res, batched):
        snaps = res.snapshots()                       # (B, N, nT+1) on device
        self.last_iters = res.iters.cpu().numpy()
        self.last_flags = res.flags.cpu().numpy()
        U = _lib.to_host(snaps)
        if self.verbose:
            errs = getattr(res, "errs", None)
            errs = None if errs is None else errs.cpu().numpy()
            At = getattr(self, "_last_At", None)
            for b in range(U.shape[0]):
                for n, k in enumerate(self.last_iters[b]):
                    if errs is not None and At is not None:
                        # the reference's own console lines (:659, :664): the error shown at iteration k is the one
                        # of iteration k - 1 (1 before the first)
                        print(f"Time Step: {n}. Time: {n * At}")
                        for j in range(int(k)):
                            print(f"Iteration: {j}, Error: {1 if j == 0 else float(errs[b, n, j - 1])}")
                    else:
                        print(f"Time Step: {n}. Iterations: {int(k)}")
        if not batched:
            self.last_iters = self.last_iters[0]
            self.last_flags = int(self.last_flags[0])
            return np.ascontiguousarray(U[0])
        return np.ascontiguousarray(U)

    # ---------------------------------------------------------------------- FOM
    def fom_burgers(self, At, nTimeSteps, u0, mu1, E, mu2):
        """Implicit-Euler / Picard FOM (reference :646-707).  Returns ``(N, nTimeSteps+1)``."""
        batched = self._batched(mu1, mu2, u0)
        self._last_At = At
        res = _fom.fom_run(self.X, np.asarray(u0, dtype=np.float64), mu1, mu2, At, int(nTimeSteps), E=E,
                           tol=1e-6, max_it=20, supg=True, trace=bool(self.verbose))
        return self._finish(res, batched)

    # --------------------------------------------------------------- POD-Galerkin / LSPG
    def pod_prom_burgers(self, At, nTimeSteps, u0, mu1, E, mu2, Phi, projection="Galerkin"):
        """POD projection ROM (reference :709-785).  ``projection`` is "Galerkin" or "LSPG",
        case-sensitive as in the reference; anything else raises ValueError."""
        batched = self._batched(mu1, mu2, u0)
        res = _rom.pod_prom_run(self.X, np.asarray(u0, dtype=np.float64), mu1, mu2, At, int(nTimeSteps),
                                np.asarray(Phi, dtype=np.float64), projection=projection, E=E)
        return self._finish(res, batched)

    # ---------------------------------------------------------------- quadratic manifold
    def pod_quadratic_manifold(self, At, nTimeSteps, u0, uxa, E, mu2, Phi, H, projection="LSPG",
                               newton_tol=1e-6, newton_itmax=25):
        """Quadratic-manifold PROM (reference :1081-1175); ``uxa`` is the left Dirichlet value."""
        batched = self._batched(uxa, mu2, u0)
        res = _rom.quadratic_run(self.X, np.asarray(u0, dtype=np.float64), uxa, mu2, At, int(nTimeSteps),
                                 np.asarray(Phi, dtype=np.float64), np.asarray(H, dtype=np.float64),
                                 projection=projection, E=E, newton_tol=newton_tol, newton_itmax=newton_itmax)
        if self.verbose and (res.flags != 0).any():
            print("  Warning: Newton did not converge")
        return self._finish(res, batched)

    # --------------------------------------------------------------------------- POD-ANN
    def pod_ann_prom(self, At, nTimeSteps, u0, mu1, E, mu2, U_p, U_s, model, projection="LSPG"):
        """POD-ANN PROM (reference :1177-1251).  ``model`` is any torch.nn.Module mapping
        (., n) -> (., nbar); it is borrowed and evaluated in float32 like the reference."""
        import copy
        batched = self._batched(mu1, mu2, u0)
        res = _rom.pod_ann_run(self.X, np.asarray(u0, dtype=np.float64), mu1, mu2, At, int(nTimeSteps),
                               np.asarray(U_p, dtype=np.float64), np.asarray(U_s, dtype=np.float64),
                               copy.deepcopy(model), projection=projection, E=E)
        return self._finish(res, batched)

    # --------------------------------------------------------------------------- POD-RBF
    def pod_rbf_prom(self, At, nTimeSteps, u0, mu1, E, mu2, U_p, U_s, X_train, W, epsilon,
                     x_min, x_max, y_min, y_max, projection="LSPG", kernel="gaussian",
                     tol_newton=1e-6, max_newton=30):
        """POD-RBF PROM with the scaled Gaussian / IMQ closure (reference :1278-1398)."""
        batched = self._batched(mu1, mu2, u0)
        res = _rom.pod_rbf_run(self.X, np.asarray(u0, dtype=np.float64), mu1, mu2, At, int(nTimeSteps), U_p, U_s,
                               X_train, W, epsilon, x_min, x_max, y_min, y_max, projection=projection,
                               kernel=kernel, E=E, tol_newton=tol_newton, max_newton=max_newton)
        return self._finish(res, batched)

    # ------------------------------------------------------------------------- local POD
    def local_prom_burgers(self, At, nTimeSteps, u0, mu1, E, mu2, kmeans, local_bases, U_global,
                           num_global_modes, projection="Galerkin"):
        """Local (clustered) POD PROM (reference :979-1079).  ``kmeans`` is the fitted
        scikit-learn KMeans of the reference (only ``cluster_centers_`` is used: ``predict`` is the
        nearest centre), ``local_bases`` a dict cluster id -> (N, r_c) basis."""
        batched = self._batched(mu1, mu2, u0)
        res = _rom.local_prom_run(self.X, np.asarray(u0, dtype=np.float64), mu1, mu2, At, int(nTimeSteps),
                                  np.asarray(kmeans.cluster_centers_, dtype=np.float64), local_bases, U_global,
                                  int(num_global_modes), projection=projection, E=E)
        return self._finish(res, batched)
