"""Offline basis builders on torch tensors (device or CPU) and the reference's .npy contracts.

  POD basis           POD/pod.py:8-14 (energy rule), :68-90 (thin SVD, file names)
  quadratic manifold  Quadratic_manifold/build_quadratic_manifold.py:25-48, quad_utils.py:63-81
  snapshot files      FEM/paper_training_stage.py:52-53
"""
from __future__ import annotations

import functools
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


@functools.lru_cache(maxsize=8)
def _round_robin(m):
    """(m' - 1, m'/2, 2) int32 pairs of a round-robin tournament over m rows (m' = m rounded up to even;
    the dummy player shows as -1)."""
    n = m + (m & 1)
    idx = np.arange(n)
    steps = np.empty((n - 1, n // 2, 2), dtype=np.int32)
    for r in range(n - 1):
        steps[r, :, 0] = idx[:n // 2]
        steps[r, :, 1] = idx[::-1][:n // 2]
        idx = np.concatenate([idx[:1], idx[-1:], idx[1:-1]])
    steps[steps >= m] = -1
    return torch.from_numpy(steps)


def jacobi_svd(R, tol=1e-15, max_sweeps=40):
    """SVD of a square device matrix by one-sided Jacobi on the HIP kernel bg_jacobi_sweep.
    Returns U, s, Vh with R = U diag(s) Vh, s descending.  The rotations act on the ROWS of R, i.e.
    on the columns of R^T: R^T J = W with orthogonal columns  =>  R = J (W^T): left vectors J, right W/|W|."""
    from . import lib as _lib
    L = _lib.load()
    m = R.shape[0]
    G = R.contiguous().clone()
    Jt = torch.eye(m, dtype=torch.float64, device=R.device)
    pairs = _round_robin(m).to(R.device)
    rot = torch.zeros((1,), dtype=torch.int32, device=R.device)
    with torch.cuda.device(R.device):
        for _ in range(max_sweeps):
            rot.zero_()
            _lib.check(L.bg_jacobi_sweep(m, m, _lib.ptr(G), _lib.ptr(Jt), _lib.ptr(pairs), pairs.shape[0], pairs.shape[1],
                                         float(tol), _lib.ptr(rot), _lib.stream_ptr(R.device)), "bg_jacobi_sweep")
            if int(rot.item()) == 0:
                break
    s = torch.linalg.vector_norm(G, dim=1)
    order = torch.argsort(s, descending=True)
    s, G, Jt = s[order], G[order], Jt[order]
    Vh = G / torch.clamp(s, min=torch.finfo(torch.float64).tiny)[:, None]
    return Jt.t().contiguous(), s, Vh


def thin_svd(A):
    """U, s, Vh of a wide matrix A (m x M), M >> m (snapshot matrices are N x B (nT+1)), on A's device.

    A^T = Q R by Householder QR (O(M m^2), rocSOLVER), then the SVD of the m x m core by one-sided Jacobi
    (bg_jacobi_sweep), and A = R^T Q^T = Vr s (Q Ur)^T.  rocSOLVER's own SVD is a Jacobi eigensolver on the
    Gram matrix: measured absolute accuracy 1e-9 sigma_max (tools/time_pod.py), which loses the singular
    triplets a 1e-6 energy tolerance still keeps.  On CPU tensors (tests, fixtures) LAPACK does the lot."""
    m, M = A.shape
    if not A.is_cuda:
        return torch.linalg.svd(A, full_matrices=False)
    if m > M:                                                     # tall: work on the transpose
        V, s, Uh = thin_svd(A.t())
        return Uh.t().contiguous(), s, V.t().contiguous()
    Q, R = torch.linalg.qr(A.t(), mode="reduced")                # (M, m), (m, m)
    Ur, s, VrT = jacobi_svd(R)
    return VrT.t().contiguous(), s, (Q @ Ur).t()


def pod_basis(S, epsilon_squared=None, n_modes=None):
    """Thin SVD of the snapshot matrix and truncation.  Returns (U[:, :K], s[:K], s_all)."""
    U, s, _ = thin_svd(S)
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
    """Ridge fit min ||E - H Q||_F^2 + alpha^2 ||H||_F^2  (quad_utils.py:63-81).

    The reference writes the minimiser through the thin SVD of Q, H = (Uq diag(s^2/(s^2+alpha^2))) (Vq^T E^T / s)^T.
    The same minimiser is the least-squares solution of [Q^T; alpha I] H^T = [E^T; 0], solved here by Householder
    QR and a triangular solve on the device: backward stable at condition sigma_max/alpha, where the device SVD
    (see thin_svd) returned H with 2e-3 relative error.  Agrees with the SVD formula to 1e-11 and with the
    committed H.npy to 1e-10 (tests)."""
    k = Q.shape[0]
    A = torch.cat([Q.t(), alpha * torch.eye(k, dtype=Q.dtype, device=Q.device)], 0)      # (Ns + k, k)
    B = torch.cat([E.t(), torch.zeros((k, E.shape[0]), dtype=Q.dtype, device=Q.device)], 0)
    Qa, Ra = torch.linalg.qr(A, mode="reduced")
    return torch.linalg.solve_triangular(Ra, Qa.t() @ B, upper=True).t().contiguous()


def build_quadratic_manifold(S, n, alpha=1e-2):
    """Phi (N, n), H (N, n(n+1)/2), q (n, Ns) from snapshots S (build_quadratic_manifold.py:25-48)."""
    U, _, _ = thin_svd(S)
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
