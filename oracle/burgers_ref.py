"""CPU oracle for the 1-D Burgers FOM/ROM Picard ("Newton") hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product path
(``1d-burgers-equation-roms_amd/``) never imports this module and raises when
its HIP library is missing.

What it is: a NumPy/SciPy restatement of the algorithm in the reference's
``FEM/fem_burgers.py`` (read as text; no source copied).  Element loops of the
reference are restated as whole-array expressions over the ``N-1`` two-node
elements while keeping the reference's order of floating-point operations
where that is cheap (Gauss-point accumulation order, element-then-element
assembly order, left-to-right ``M + At*C + At*E*K``).

Parity: PINNED.  ``tests/test_oracle_golden.py`` checks this module against
(i) slices of the ``.npy`` outputs the reference repository itself commits
(FOM snapshots, POD modes, POD-PROM / quadratic-PROM solutions) and (ii)
vectors produced by importing the reference in the build container
(``tests/golden/make_golden.py``, committed next to its outputs).

Reference citations are ``FEM/fem_burgers.py:<line>`` unless another file is
named.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

__all__ = [
    "gauss_tables", "mass_tridiag", "diffusion_tridiag", "convection_tridiag",
    "forcing_vector", "supg_term", "tridiag_matvec", "tridiag_solve",
    "system_tridiag", "fom_burgers", "pod_prom_burgers", "get_sym", "get_dQ_dq",
    "pod_quadratic_manifold", "mlp_forward", "mlp_jacobian", "pod_ann_prom",
    "pod_basis", "n_modes_for_tolerance", "compute_H", "build_Q", "predict_on_fom_grid", "fd_newton", "rbf_value", "rbf_jacobian", "pod_rbf_prom", "local_prom_burgers",
]


# --------------------------------------------------------------------------
# Gauss tables                                   (fem_burgers.py:315-322)
# --------------------------------------------------------------------------
def gauss_tables():
    """2-point Gauss tables: ``zgp=±1/√3``, ``wgp=[1,1]``, ``N[gp,node]``, ``Nxi``."""
    zgp = np.array([-np.sqrt(3) / 3, np.sqrt(3) / 3])
    wgp = np.array([1.0, 1.0])
    N = np.array([(1 - zgp) / 2, (1 + zgp) / 2]).T      # N[gp, local node]
    Nxi = np.array([[-0.5, 0.5], [-0.5, 0.5]])
    return zgp, wgp, N, Nxi


def _elem_geometry(X):
    """Per-element Jacobian ``J = Nxi @ x_e`` and ``dV = w*|J|`` (:339-343)."""
    X = np.asarray(X, dtype=np.float64)
    xl, xr = X[:-1], X[1:]
    J = -0.5 * xl + 0.5 * xr                 # dN_dxi_gp @ x_element, same for both gp
    dV = 1.0 * np.abs(J)
    return xl, xr, J, dV


def _assemble_tridiag(Ell, Elr, Erl, Err):
    """Scatter 2x2 element matrices into (lower, diag, upper) in element order."""
    ne = Ell.shape[-1]
    n = ne + 1
    shp = Ell.shape[:-1]
    lo = np.zeros(shp + (n,))
    di = np.zeros(shp + (n,))
    up = np.zeros(shp + (n,))
    # node i receives [r,r] of element i-1 first, then [l,l] of element i
    di[..., 1:] += Err
    di[..., :-1] += Ell
    up[..., :-1] = Elr          # A[i, i+1]
    lo[..., 1:] = Erl           # A[i, i-1]
    return lo, di, up


def mass_tridiag(X):
    """Consistent P1 mass matrix as three diagonals (:324-353)."""
    _, _, N, _ = gauss_tables()
    _, _, _, dV = _elem_geometry(X)
    E = [[None, None], [None, None]]
    for i in range(2):
        for j in range(2):
            acc = np.zeros_like(dV)
            for gp in range(2):
                acc = acc + (N[gp, i] * N[gp, j]) * dV
            E[i][j] = acc
    return _assemble_tridiag(E[0][0], E[0][1], E[1][0], E[1][1])


def diffusion_tridiag(X):
    """P1 stiffness matrix as three diagonals (:355-387)."""
    _, _, _, Nxi = gauss_tables()
    _, _, J, dV = _elem_geometry(X)
    E = [[None, None], [None, None]]
    for i in range(2):
        for j in range(2):
            acc = np.zeros_like(dV)
            for gp in range(2):
                dNi = Nxi[gp, i] / J
                dNj = Nxi[gp, j] / J
                acc = acc + (dNi * dNj) * dV
            E[i][j] = acc
    return _assemble_tridiag(E[0][0], E[0][1], E[1][0], E[1][1])


def convection_tridiag(X, U):
    """Convection matrix ``C(U)`` as three diagonals (:389-425).

    ``U`` may be ``(N,)`` or ``(B, N)``; the diagonals carry the same leading axis.
    """
    _, _, N, Nxi = gauss_tables()
    _, _, J, dV = _elem_geometry(X)
    U = np.asarray(U, dtype=np.float64)
    ul, ur = U[..., :-1], U[..., 1:]
    E = [[None, None], [None, None]]
    for i in range(2):
        for j in range(2):
            acc = np.zeros_like(ul)
            for gp in range(2):
                u_gp = N[gp, 0] * ul + N[gp, 1] * ur
                dNj = Nxi[gp, j] / J
                acc = acc + (N[gp, i] * (u_gp * dNj)) * dV
            E[i][j] = acc
    return _assemble_tridiag(E[0][0], E[0][1], E[1][0], E[1][1])


def forcing_vector(X, mu2):
    """Load vector of ``0.02*exp(mu2*x)`` by 2-pt Gauss (:427-461).

    ``mu2`` scalar -> ``(N,)``; ``mu2`` of shape ``(B,)`` -> ``(B, N)``.
    """
    _, _, N, _ = gauss_tables()
    xl, xr, _, dV = _elem_geometry(X)
    mu2 = np.asarray(mu2, dtype=np.float64)[..., None]
    Fl = np.zeros(mu2.shape[:-1] + xl.shape)
    Fr = np.zeros_like(Fl)
    for gp in range(2):
        x_gp = N[gp, 0] * xl + N[gp, 1] * xr
        f_gp = 0.02 * np.exp(mu2 * x_gp)
        Fl = Fl + f_gp * N[gp, 0] * dV
        Fr = Fr + f_gp * N[gp, 1] * dV
    F = np.zeros(Fl.shape[:-1] + (len(xl) + 1,))
    F[..., 1:] += Fr
    F[..., :-1] += Fl
    return F


def supg_term(X, U, mu2):
    """SUPG stabilisation vector (:500-581): ``tau_e * R(u) * dN/dx`` integrated."""
    _, _, N, Nxi = gauss_tables()
    xl, xr, J, dV = _elem_geometry(X)
    U = np.asarray(U, dtype=np.float64)
    mu2 = np.asarray(mu2, dtype=np.float64)
    if U.ndim > 1:
        mu2 = mu2[..., None]
    ul, ur = U[..., :-1], U[..., 1:]
    h_e = xr - xl
    u_e = (ul + ur) / 2.0                                  # np.mean of two
    vel = np.where(np.abs(u_e) > 1.0e-10, np.abs(u_e), 1.0e-10)
    tau = 0.5 * h_e / (2.0 * vel)
    du_dx = (ur - ul) / h_e
    Sl = np.zeros_like(ul)
    Sr = np.zeros_like(ul)
    for gp in range(2):
        x_gp = N[gp, 0] * xl + N[gp, 1] * xr
        u_gp = N[gp, 0] * ul + N[gp, 1] * ur
        f_gp = 0.02 * np.exp(mu2 * x_gp)
        R_gp = (u_gp * du_dx) - f_gp
        Sl = Sl + tau * R_gp * (Nxi[gp, 0] / J) * dV
        Sr = Sr + tau * R_gp * (Nxi[gp, 1] / J) * dV
    S = np.zeros(Sl.shape[:-1] + (len(xl) + 1,))
    S[..., 1:] += Sr
    S[..., :-1] += Sl
    return S


# --------------------------------------------------------------------------
# Tridiagonal helpers (stand in for scipy.sparse csc arithmetic / SuperLU)
# --------------------------------------------------------------------------
def tridiag_matvec(lo, di, up, x):
    """``A @ x`` accumulated in column order like a CSC mat-vec (:683,689)."""
    y = np.zeros(np.broadcast(di, x).shape)
    y[..., 1:] += lo[..., 1:] * x[..., :-1]
    y += di * x
    y[..., :-1] += up[..., :-1] * x[..., 1:]
    return y


def tridiag_solve(lo, di, up, rhs):
    """Pivoted banded solve (LAPACK ``gbsv``) standing in for ``spsolve`` (:692)."""
    n = di.shape[-1]
    ab = np.zeros((3, n))
    ab[0, 1:] = up[:-1]
    ab[1, :] = di
    ab[2, :-1] = lo[1:]
    return sla.solve_banded((1, 1), ab, rhs, check_finite=False)


def tridiag_dense(lo, di, up):
    n = di.shape[-1]
    A = np.zeros((n, n))
    i = np.arange(n)
    A[i, i] = di
    A[i[1:], i[:-1]] = lo[1:]
    A[i[:-1], i[1:]] = up[:-1]
    return A


def tridiag_matmat(lo, di, up, W):
    """``A @ W`` for a dense ``(N, r)`` block."""
    Y = di[:, None] * W
    Y[1:] += lo[1:, None] * W[:-1]
    Y[:-1] += up[:-1, None] * W[1:]
    return Y


def system_tridiag(M3, K3, C3, At, E):
    """``A = M + At*C + At*E*K`` then Dirichlet row 0 (:676-680)."""
    out = []
    for m, c, k in zip(M3, C3, K3):
        a = (m + At * c) + (At * E) * k
        out.append(a)
    lo, di, up = out
    lo = lo.copy(); di = di.copy(); up = up.copy()
    lo[..., 0] = 0.0
    di[..., 0] = 1.0
    up[..., 0] = 0.0
    return lo, di, up


# --------------------------------------------------------------------------
# FOM                                               (fem_burgers.py:646-707)
# --------------------------------------------------------------------------
def fom_burgers(X, At, nTimeSteps, u0, mu1, E, mu2, tol=1e-6, max_it=20,
                return_iters=False, return_errs=False):
    """Implicit-Euler / Picard FOM; returns ``U (N, nTimeSteps+1)`` (and the iteration counts; and ``errs
    (nTimeSteps, max_it)``, error_U of every iteration as the reference prints it at :664, NaN where none ran)."""
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = u0
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)          # constant; the reference recomputes it (:673)
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    errs = np.full((nTimeSteps, max_it), np.nan)
    for nstep in range(nTimeSteps):
        Un = U[:, nstep]
        U0 = Un
        err = 1.0
        k = 0
        U1 = U0
        Mun = tridiag_matvec(*M3, Un)
        while err > tol and k < max_it:
            C3 = convection_tridiag(X, U0)
            S = supg_term(X, U0, mu2)
            lo, di, up = system_tridiag(M3, K3, C3, At, E)
            b = Mun + At * F - At * S
            b[0] = mu1
            R = tridiag_matvec(lo, di, up, U0) - b
            dU = tridiag_solve(lo, di, up, -R)
            U1 = U0 + dU
            err = np.linalg.norm(dU) / np.linalg.norm(U1)
            errs[nstep, k] = err
            U0 = U1
            k += 1
        iters[nstep] = k
        U[:, nstep + 1] = U1
    if return_errs:
        return U, iters, errs
    return (U, iters) if return_iters else U


# --------------------------------------------------------------------------
# POD-Galerkin / LSPG PROM                          (fem_burgers.py:709-785)
# --------------------------------------------------------------------------
def _reduce(lo, di, up, R, W, projection):
    """``Ar, br`` for Galerkin ``Wᵀ A W`` or LSPG ``(AW)ᵀ(AW)`` (:754-762)."""
    Y = tridiag_matmat(lo, di, up, W)
    if projection == "galerkin":
        return W.T @ Y, W.T @ R
    return Y.T @ Y, Y.T @ R


def pod_prom_burgers(X, At, nTimeSteps, u0, mu1, E, mu2, Phi, projection="Galerkin",
                     tol=1e-6, max_it=20, return_iters=False):
    if projection not in ("Galerkin", "LSPG"):       # case-sensitive (:754-764)
        raise ValueError(f"Projection method '{projection}' is not available. "
                         "Please use 'Galerkin' or 'LSPG'.")
    proj = projection.lower()
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = u0
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    for nstep in range(nTimeSteps):
        Un = U[:, nstep]
        U0 = Un
        err, k, U1 = 1.0, 0, Un
        Mun = tridiag_matvec(*M3, Un)
        while err > tol and k < max_it:
            C3 = convection_tridiag(X, U0)
            S = supg_term(X, U0, mu2)
            lo, di, up = system_tridiag(M3, K3, C3, At, E)
            b = Mun + At * F - At * S
            b[0] = mu1
            R = tridiag_matvec(lo, di, up, U0) - b
            Ar, br = _reduce(lo, di, up, R, Phi, proj)
            dq = np.linalg.solve(Ar, -br)
            q = Phi.T @ U0 + dq
            U1 = Phi @ q
            err = np.linalg.norm(dq) / np.linalg.norm(q)
            U0 = U1
            k += 1
        iters[nstep] = k
        U[:, nstep + 1] = U1
    return (U, iters) if return_iters else U


# --------------------------------------------------------------------------
# Quadratic manifold                       (fem_burgers.py:263-312, 1081-1175)
# --------------------------------------------------------------------------
def get_sym(q):
    """Unique monomials ``q_i q_j, j>=i`` in row-major upper-triangle order (:263-273)."""
    i, j = np.triu_indices(len(q))
    return q[i] * q[j]


def get_dQ_dq(q):
    """``d get_sym / dq`` of shape ``(k, n)`` (:292-312)."""
    n = len(q)
    i, j = np.triu_indices(n)
    k = len(i)
    dQ = np.zeros((k, n))
    rows = np.arange(k)
    diag = i == j
    dQ[rows[diag], i[diag]] = 2.0 * q[i[diag]]
    off = ~diag
    dQ[rows[off], i[off]] = q[j[off]]
    dQ[rows[off], j[off]] = q[i[off]]
    return dQ


def pod_quadratic_manifold(X, At, nTimeSteps, u0, uxa, E, mu2, Phi, H, projection="LSPG",
                           newton_tol=1e-6, newton_itmax=25, return_iters=False):
    proj = projection.lower()
    if proj not in ("galerkin", "lspg"):
        raise ValueError("projection must be 'Galerkin' or 'LSPG'")
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = np.asarray(u0).copy()
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    for m in range(nTimeSteps):
        q = Phi.T @ U[:, m]
        u = Phi @ q + H @ get_sym(q)
        Mun = tridiag_matvec(*M3, U[:, m])
        it_done = 0
        for it in range(newton_itmax):
            C3 = convection_tridiag(X, u)
            lo, di, up = system_tridiag(M3, K3, C3, At, E)
            b = Mun + At * F                          # no SUPG in this variant (:1142)
            b[0] = uxa
            R = tridiag_matvec(lo, di, up, u) - b
            T = Phi + H @ get_dQ_dq(q)
            Ar, br = _reduce(lo, di, up, R, T, proj)
            dq = np.linalg.solve(Ar, -br)
            q = q + dq
            u = Phi @ q + H @ get_sym(q)
            rel = np.linalg.norm(dq) / max(1e-14, np.linalg.norm(q))
            it_done = it + 1
            if rel < newton_tol:
                break
        iters[m] = it_done
        U[:, m + 1] = u
    return (U, iters) if return_iters else U


# --------------------------------------------------------------------------
# POD-ANN                                     (fem_burgers.py:1177-1275)
# --------------------------------------------------------------------------
def _elu32(x):
    x = x.astype(np.float32, copy=False)
    return np.where(x > 0, x, np.expm1(np.minimum(x, np.float32(0))).astype(np.float32))


def mlp_forward(weights, biases, q_p):
    """fp32 MLP with ELU between layers (``POD-ANN/pod_ann.py:38-56``)."""
    x = np.asarray(q_p, dtype=np.float32)
    L = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        x = (x @ W.T.astype(np.float32) + b.astype(np.float32)).astype(np.float32)
        if i < L - 1:
            x = _elu32(x)
    return x


def mlp_jacobian(weights, biases, q_p):
    """fp32 input-Jacobian ``(nbar, n)``; analytic stand-in for
    ``torch.autograd.functional.jacobian`` (:1254-1275)."""
    x = np.asarray(q_p, dtype=np.float32)
    J = np.eye(len(x), dtype=np.float32)
    L = len(weights)
    for i, (W, b) in enumerate(zip(weights, biases)):
        W = W.astype(np.float32)
        z = (W @ x + b.astype(np.float32)).astype(np.float32)
        J = (W @ J).astype(np.float32)
        if i < L - 1:
            d = np.where(z > 0, np.float32(1), np.exp(np.minimum(z, np.float32(0)))).astype(np.float32)
            J = (d[:, None] * J).astype(np.float32)
            x = _elu32(z)
        else:
            x = z
    return J


def pod_ann_prom(X, At, nTimeSteps, u0, mu1, E, mu2, U_p, U_s, weights, biases,
                 projection="LSPG", tol=1e-6, max_it=50, return_iters=False):
    proj = projection.lower()
    if proj not in ("galerkin", "lspg"):
        raise ValueError("projection must be 'Galerkin' or 'LSPG'")
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = u0
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    for nt in range(nTimeSteps):
        U0 = U[:, nt].copy()
        q_p = U_p.T @ U0
        Mun = tridiag_matvec(*M3, U[:, nt])
        err, it, U1 = 1.0, 0, U0
        while err > tol and it < max_it:
            C3 = convection_tridiag(X, U0)
            S = supg_term(X, U0, mu2)
            lo, di, up = system_tridiag(M3, K3, C3, At, E)
            b = Mun + At * F - At * S
            b[0] = mu1
            R = tridiag_matvec(lo, di, up, U0) - b
            dN = mlp_jacobian(weights, biases, q_p.astype(np.float32)).astype(np.float64)
            dD = U_p + U_s @ dN
            Ar, br = _reduce(lo, di, up, R, dD, proj)
            dq = np.linalg.solve(Ar, -br)
            q_p = q_p + dq
            q_s = mlp_forward(weights, biases, q_p.astype(np.float32)).astype(np.float64)
            U1 = U_p @ q_p + U_s @ q_s
            err = np.linalg.norm(dq) / (np.linalg.norm(q_p) + 1e-14)
            U0 = U1
            it += 1
        iters[nt] = it
        U[:, nt + 1] = U1
    return (U, iters) if return_iters else U


# --------------------------------------------------------------------------
# Offline bases                     (POD/pod.py:8-14,68-90; quad_utils.py)
# --------------------------------------------------------------------------
def n_modes_for_tolerance(s, epsilon_squared):
    """``K = argmax(1 - cumsum(σ²)/Σσ² <= ε²) + 1`` (``POD/pod.py:8-14``)."""
    s_sorted = np.sort(s)[::-1]
    c = np.cumsum(s_sorted ** 2)
    loss = 1.0 - c / c[-1]
    return int(np.argmax(loss <= epsilon_squared) + 1)


def pod_basis(snapshots, epsilon_squared):
    """Thin SVD + energy truncation (``POD/pod.py:80-90``): ``(U[:, :K], s[:K])``."""
    Um, s, _ = np.linalg.svd(snapshots, full_matrices=False)
    K = n_modes_for_tolerance(s, epsilon_squared)
    return Um[:, :K], s[:K]


def build_Q(q):
    """``Q (k, Ns)`` of unique monomials (``Quadratic_manifold/quad_utils.py:21-31``)."""
    i, j = np.triu_indices(q.shape[0])
    return q[i] * q[j]


def compute_H(Q, Em, alpha):
    """Ridge fit ``min ||E - H Q||² + α²||H||²`` by thin SVD of Q
    (``Quadratic_manifold/quad_utils.py:63-81``)."""
    Uq, s, VqT = np.linalg.svd(Q, full_matrices=False)
    s2 = s ** 2
    f = s2 / (s2 + alpha ** 2)
    Gamma = (VqT @ Em.T) / s[:, None]
    return ((Uq * f) @ Gamma).T


# --------------------------------------------------------------------------
# Non-intrusive decoder          (Non-Instrusive/predict_pod_ann.py:73-80)
# --------------------------------------------------------------------------
def predict_on_fom_grid(mu1, mu2, Nt, U_modes, weights, biases, mean, std):
    """``Uhat = U_modes @ MLP(standardize([mu1, mu2, tau]))ᵀ`` with a float32 MLP."""
    tau = np.linspace(0.0, 1.0, Nt)
    Z = np.column_stack([np.full(Nt, mu1), np.full(Nt, mu2), tau])
    std = np.array(std, dtype=np.float64).copy()
    std[std == 0] = 1.0
    Zs = ((Z - mean) / std).astype(np.float32)
    Q = np.stack([mlp_forward(weights, biases, z) for z in Zs])
    return U_modes @ Q.T.astype(np.float64)


# --------------------------------------------------------------------------
# Finite-difference true-Newton stepper          (FD/fd_burgers.py:14-107)
# --------------------------------------------------------------------------
def _fd_residual(U, Up, dt, dx, s_src):
    """compute_residual (FD/fd_burgers.py:28-35) on the last axis of U (any leading batch axes)."""
    nu = 0.25 * dx * np.max(np.abs(U), axis=-1, keepdims=True)
    R = np.zeros_like(U)
    conv = (0.5 * U[..., 2:] ** 2 - 0.5 * U[..., :-2] ** 2) / (2 * dx)
    diff = nu * (U[..., 2:] - 2 * U[..., 1:-1] + U[..., :-2]) / dx ** 2
    R[..., 1:-1] = (U[..., 1:-1] - Up[1:-1]) / dt + conv - s_src[1:-1] - diff
    return R


def fd_jacobian_fd(U, Up, dt, dx, s_src, epsilon=1e-8):
    """compute_jacobian_fd (FD/fd_burgers.py:46-57): one-sided differences of the residual, column by column; the
    artificial viscosity is re-evaluated for every perturbed state, so the matrix is dense in the row of max|U|."""
    N = len(U)
    Rb = _fd_residual(U, Up, dt, dx, s_src)
    Upert = U[None, :] + epsilon * np.eye(N)[1:N - 1]                 # row j-1: U with U[j] += eps
    Rp = _fd_residual(Upert, Up, dt, dx, s_src)
    J = np.zeros((N, N))
    J[1:N - 1, 1:N - 1] = ((Rp[:, 1:N - 1] - Rb[None, 1:N - 1]) / epsilon).T
    return J


def fd_newton(a, b, N, dt, n_steps, U0, mu1, mu2, max_iter=30, tol=1e-8, return_iters=False, use_fd_jacobian=False):
    """``FDBurgers.fom_burgers_newton``: central differences, lagged artificial viscosity
    ``nu = 0.25 dx max|U|``, backward Euler, Newton with the tridiagonal analytical Jacobian of
    FD/fd_burgers.py:36-43 (or, ``use_fd_jacobian``, the dense finite-difference one of :46-57 and a dense solve),
    stop on max residual or max relative update < tol."""
    dx = (b - a) / (N - 1)
    x = np.linspace(a, b, N)
    s_src = 0.02 * np.exp(mu2 * x)
    Uall = np.zeros((N, n_steps + 1))
    Uc = np.array(U0, dtype=np.float64).copy()
    Uc[0] = mu1; Uc[-1] = Uc[-2]
    Uall[:, 0] = Uc
    iters = np.zeros(n_steps, dtype=np.int32)
    for step in range(n_steps):
        Up = Uc.copy()
        Ug = Up.copy()
        k = 0
        for it in range(max_iter):
            Ug[0] = mu1; Ug[-1] = Ug[-2]
            nu = 0.25 * dx * np.max(np.abs(Ug))
            R = np.zeros(N)
            conv = (0.5 * Ug[2:] ** 2 - 0.5 * Ug[:-2] ** 2) / (2 * dx)
            diff = nu * (Ug[2:] - 2 * Ug[1:-1] + Ug[:-2]) / dx ** 2
            R[1:-1] = (Ug[1:-1] - Up[1:-1]) / dt + conv - s_src[1:-1] - diff
            if np.max(np.abs(R[1:-1])) < tol:
                break
            if use_fd_jacobian:
                J = fd_jacobian_fd(Ug, Up, dt, dx, s_src)
                dU = np.zeros(N)
                dU[1:-1] = np.linalg.solve(J[1:-1, 1:-1], -R[1:-1])
                rel = np.max(np.abs(dU[1:-1])) / max(np.max(np.abs(Ug[1:-1])), 1e-15)
                Ug = Ug + dU
                k += 1
                if rel < tol:
                    break
                continue
            lo = np.zeros(N); di = np.ones(N); up = np.zeros(N)
            lo[1:-1] = -Ug[:-2] / (2 * dx) - nu / dx ** 2
            up[1:-1] = Ug[2:] / (2 * dx) - nu / dx ** 2
            di[1:-1] = 1 / dt + 2 * nu / dx ** 2
            lo[1] = 0.0; up[-2] = 0.0              # interior block J[1:-1, 1:-1]
            dU = tridiag_solve(lo, di, up, -R)
            dU[0] = 0.0; dU[-1] = 0.0
            rel = np.max(np.abs(dU[1:-1])) / max(np.max(np.abs(Ug[1:-1])), 1e-15)
            Ug = Ug + dU
            k += 1
            if rel < tol:
                break
        iters[step] = k
        Uc = Ug.copy()
        Uc[0] = mu1; Uc[-1] = Uc[-2]
        Uall[:, step + 1] = Uc
    return (Uall, iters) if return_iters else Uall


# --------------------------------------------------------------------------
# POD-RBF PROM                       (FEM/fem_burgers.py:160-260, 1278-1398)
# --------------------------------------------------------------------------
def _rbf_scale(q_p, x_min, x_max):
    dx = (x_max - x_min).copy()
    dx[dx < 1e-15] = 1.0
    return 2.0 * ((q_p - x_min) / dx) - 1.0, dx


def rbf_value(q_p, X_train, W, eps, kernel, x_min, x_max, y_min, y_max):
    """``interpolate_with_rbf_scaled`` (:225-236): scale, kernel sum, unscale."""
    xs, _ = _rbf_scale(q_p, x_min, x_max)
    r = np.linalg.norm(xs[None, :] - X_train, axis=1)
    k = np.exp(-(eps * r) ** 2) if kernel == "gaussian" else 1.0 / np.sqrt(1.0 + (eps * r) ** 2)
    dy = (y_max - y_min).copy()
    dy[dy < 1e-15] = 1.0
    return 0.5 * (k @ W + 1.0) * dy + y_min


def rbf_jacobian(q_p, X_train, W, eps, kernel, x_min, x_max, y_min, y_max):
    """``compute_rbf_jacobian_full`` (:238-260): ``diag(dy/2) Wᵀ G diag(2/dx)``."""
    xs, dx = _rbf_scale(q_p, x_min, x_max)
    diff = xs[None, :] - X_train
    r = np.linalg.norm(diff, axis=1)
    if kernel == "gaussian":
        k = np.exp(-(eps * r) ** 2)
        G = (-2.0 * eps ** 2) * (k[:, None] * diff)
    else:
        k = (1.0 + (eps ** 2) * (r ** 2)) ** (-0.5)
        G = (-(eps ** 2)) * ((k ** 3)[:, None] * diff)
    dy = (y_max - y_min).copy()
    dy[dy < 1e-15] = 1.0
    J = (W.T @ G) * (2.0 / dx)[None, :]
    return (0.5 * dy)[:, None] * J


def pod_rbf_prom(X, At, nTimeSteps, u0, mu1, E, mu2, U_p, U_s, X_train, W, epsilon, x_min, x_max, y_min, y_max,
                 projection="LSPG", kernel="gaussian", tol_newton=1e-6, max_newton=30, return_iters=False):
    if kernel not in ("gaussian", "imq"):
        raise ValueError("kernel must be 'gaussian' or 'imq'.")
    proj = projection.lower()
    if proj not in ("galerkin", "lspg"):
        raise ValueError("projection must be 'LSPG' or 'Galerkin'.")
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = u0
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    rb = (X_train, W, epsilon, kernel, x_min, x_max, y_min, y_max)
    for nstep in range(nTimeSteps):
        U0 = U[:, nstep].copy()
        Mun = tridiag_matvec(*M3, U[:, nstep])
        err, it, U1 = 1.0, 0, U0
        while err > tol_newton and it < max_newton:
            C3 = convection_tridiag(X, U0)
            S = supg_term(X, U0, mu2)
            A3 = [m + At * (c + E * k) for m, c, k in zip(M3, C3, K3)]        # A = M + At*(C + E*K)   (:1338)
            lo, di, up = [a.copy() for a in A3]
            lo[0] = 0.0; di[0] = 1.0; up[0] = 0.0
            b = Mun + At * (F - S)
            b[0] = mu1
            R = tridiag_matvec(lo, di, up, U0) - b
            q_p = U_p.T @ U0
            dD = U_p + U_s @ rbf_jacobian(q_p, *rb)
            Ar, br = _reduce(lo, di, up, R, dD, proj)
            dq = np.linalg.solve(Ar, -br)
            q_new = q_p + dq
            U1 = U_p @ q_new + U_s @ rbf_value(q_new, *rb)
            denom = np.linalg.norm(q_new)
            err = (np.linalg.norm(dq) / denom) if denom > 0 else np.linalg.norm(dq)
            U0 = U1
            it += 1
        iters[nstep] = it
        U[:, nstep + 1] = U1
    return (U, iters) if return_iters else U


# --------------------------------------------------------------------------
# Local (clustered) POD PROM                     (FEM/fem_burgers.py:979-1079)
# --------------------------------------------------------------------------
def local_prom_burgers(X, At, nTimeSteps, u0, mu1, E, mu2, centers, local_bases, U_global, num_global_modes,
                       projection="Galerkin", tol=1e-6, max_it=20, return_iters=False):
    """``pod_prom_burgers`` with ONE local basis per time step: the cluster whose centre is nearest to
    ``U_global[:, :m]ᵀ u^n`` (what ``kmeans.predict`` returns, :1011-1013)."""
    if projection not in ("Galerkin", "LSPG"):
        raise ValueError(f"Projection method '{projection}' is not available. Please use 'Galerkin' or 'LSPG'.")
    proj = projection.lower()
    X = np.asarray(X, dtype=np.float64)
    n = len(X)
    U = np.zeros((n, nTimeSteps + 1))
    U[:, 0] = u0
    M3 = mass_tridiag(X)
    K3 = diffusion_tridiag(X)
    F = forcing_vector(X, mu2)
    Ug = U_global[:, :num_global_modes]
    iters = np.zeros(nTimeSteps, dtype=np.int32)
    clusters = np.zeros(nTimeSteps, dtype=np.int32)
    for nstep in range(nTimeSteps):
        Un = U[:, nstep]
        U0 = Un
        qg = Ug.T @ U0
        cid = int(np.argmin(((centers - qg[None, :]) ** 2).sum(1)))
        Phi = local_bases[cid]
        clusters[nstep] = cid
        err, k, U1 = 1.0, 0, Un
        Mun = tridiag_matvec(*M3, Un)
        while err > tol and k < max_it:
            C3 = convection_tridiag(X, U0)
            S = supg_term(X, U0, mu2)
            lo, di, up = system_tridiag(M3, K3, C3, At, E)
            b = Mun + At * F - At * S
            b[0] = mu1
            R = tridiag_matvec(lo, di, up, U0) - b
            Ar, br = _reduce(lo, di, up, R, Phi, proj)
            dq = np.linalg.solve(Ar, -br)
            q = Phi.T @ U0 + dq
            U1 = Phi @ q
            err = np.linalg.norm(dq) / np.linalg.norm(q)
            U0 = U1
            k += 1
        iters[nstep] = k
        U[:, nstep + 1] = U1
    return (U, iters, clusters) if return_iters else U
