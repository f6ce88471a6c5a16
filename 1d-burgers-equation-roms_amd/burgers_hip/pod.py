"""Offline basis builders on torch tensors (device or CPU) and the reference's .npy contracts.

  POD basis           POD/pod.py:8-14 (energy rule), :68-90 (thin SVD, file names)
  quadratic manifold  Quadratic_manifold/build_quadratic_manifold.py:25-48, quad_utils.py:63-81
  snapshot files      FEM/paper_training_stage.py:52-53
"""
from __future__ import annotations

import os

import numpy as np
import torch


def snapshot_matrix(hist):
    """(B, nT+1, N) time-major histories -> (N, B*(nT+1)) snapshot matrix, i.e. np.hstack of the
    per-sample (N, nT+1) arrays the reference stacks (POD/pod.py:80-82)."""
    B, T, N = hist.shape
    return hist.reshape(B * T, N).t()


def n_modes_for_tolerance(s, epsilon_squared):
    """K = argmax(1 - cumsum(s^2)/sum(s^2) <= eps^2) + 1   (POD/pod.py:8-14)."""
    s = torch.as_tensor(s, dtype=torch.float64)
    s_sorted = torch.sort(s, descending=True).values
    c = torch.cumsum(s_sorted ** 2, 0)
    loss = 1.0 - c / c[-1]
    hit = torch.nonzero(loss <= epsilon_squared)
    return int(hit[0]) + 1 if len(hit) else 1


def pod_basis(S, epsilon_squared=None, n_modes=None):
    """Thin SVD of the snapshot matrix and truncation.  Returns (U[:, :K], s[:K], s_all)."""
    U, s, _ = torch.linalg.svd(S, full_matrices=False)
    K = n_modes if n_modes is not None else n_modes_for_tolerance(s, epsilon_squared)
    return U[:, :K].contiguous(), s[:K].contiguous(), s


def align_signs(U, U_ref):
    """Singular vectors are defined up to sign; flip columns of U to match U_ref."""
    sgn = torch.sign((U * U_ref).sum(0))
    sgn[sgn == 0] = 1
    return U * sgn


def build_Q(q):
    """(n, Ns) reduced coordinates -> (k, Ns) unique monomials q_i q_j, j >= i (quad_utils.py:21-31)."""
    n = q.shape[0]
    I, J = np.triu_indices(n)
    I = torch.as_tensor(I, device=q.device); J = torch.as_tensor(J, device=q.device)
    return q[I] * q[J]


def compute_H(Q, E, alpha):
    """Ridge fit min ||E - H Q||_F^2 + alpha^2 ||H||_F^2 through the thin SVD of Q (quad_utils.py:63-81)."""
    Uq, s, VqT = torch.linalg.svd(Q, full_matrices=False)
    s2 = s ** 2
    f = s2 / (s2 + alpha ** 2)
    Gamma = (VqT @ E.t()) / s[:, None]
    return ((Uq * f) @ Gamma).t().contiguous()


def build_quadratic_manifold(S, n, alpha=1e-2):
    """Phi (N, n), H (N, n(n+1)/2), q (n, Ns) from snapshots S (build_quadratic_manifold.py:25-48)."""
    U, _, _ = torch.linalg.svd(S, full_matrices=False)
    Phi = U[:, :n].contiguous()
    q = Phi.t() @ S
    Q = build_Q(q)
    Em = S - Phi @ q
    return Phi, compute_H(Q, Em, alpha), q


# ---- .npy contracts --------------------------------------------------------------------------
def snapshot_filename(mu1, mu2):
    return f"fem_simulation_mu1_{mu1:.3f}_mu2_{mu2:.4f}.npy"          # paper_training_stage.py:52


def save_snapshots(directory, snaps, mu1, mu2):
    """Write one C-ordered (N, nT+1) float64 .npy per sample, named like the reference."""
    os.makedirs(directory, exist_ok=True)
    snaps = snaps.detach().cpu().numpy() if isinstance(snaps, torch.Tensor) else np.asarray(snaps)
    paths = []
    for U, a, b in zip(snaps, np.atleast_1d(mu1), np.atleast_1d(mu2)):
        p = os.path.join(directory, snapshot_filename(float(a), float(b)))
        np.save(p, np.ascontiguousarray(U, dtype=np.float64))
        paths.append(p)
    return paths


def save_modes(directory, U, s, eps2):
    """U_modes_tol_{eps2:.0e}.npy and Singular_values_modes_tol_{eps2:.0e}.npy (POD/pod.py:73-76)."""
    os.makedirs(directory, exist_ok=True)
    pu = os.path.join(directory, f"U_modes_tol_{eps2:.0e}.npy")
    ps = os.path.join(directory, f"Singular_values_modes_tol_{eps2:.0e}.npy")
    np.save(pu, np.ascontiguousarray(U.detach().cpu().numpy(), dtype=np.float64))
    np.save(ps, np.ascontiguousarray(s.detach().cpu().numpy(), dtype=np.float64))
    return pu, ps
