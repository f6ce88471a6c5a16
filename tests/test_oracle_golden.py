"""Pins the oracle (oracle/burgers_ref.py and the C restatement) to the reference:
golden vectors produced by importing the reference (tests/golden/make_golden.py) and
column slices of the .npy results the reference repository commits."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from oracle import burgers_ref as br
from oracle import burgers_ref_c as bc


def test_assemblers_general_mesh():
    g = load_golden("fom_general.npz")
    X, u, mu2 = g["X"], g["u"], float(g["mu2"])
    assert np.abs(br.tridiag_dense(*br.mass_tridiag(X)) - g["M"]).max() < 1e-15
    assert np.abs(br.tridiag_dense(*br.diffusion_tridiag(X)) - g["K"]).max() < 1e-14
    assert np.abs(br.tridiag_dense(*br.convection_tridiag(X, u)) - g["C"]).max() < 1e-14
    assert np.abs(br.forcing_vector(X, mu2) - g["F"]).max() < 1e-15
    assert np.abs(br.supg_term(X, u, mu2) - g["S"]).max() < 1e-14
    # eps_vel branch of tau_e (|u_e| <= 1e-10): huge tau, compare relatively
    S_small = br.supg_term(X, g["u_small"], mu2)
    assert np.abs(S_small - g["S_small"]).max() <= 1e-13 * np.abs(g["S_small"]).max()


def test_fom_general_mesh_live_reference():
    g = load_golden("fom_general.npz")
    U, it = br.fom_burgers(g["X"], float(g["At"]), int(g["nT"]), g["u0"], float(g["mu1"]), float(g["E"]),
                           float(g["mu2"]), return_iters=True)
    assert rel_l2(U, g["U"]) < 1e-13
    assert np.array_equal(it, g["iters"])
    h, itc = bc.fom_run(g["X"], g["u0"], float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]),
                        E=float(g["E"]))
    assert rel_l2(h[0].T, g["U"]) < 1e-13
    assert np.array_equal(itc[0], g["iters"])


def test_fom_config1_n256_live_reference():
    g = load_golden("fom_n256.npz")
    U, it = br.fom_burgers(g["X"], float(g["At"]), int(g["nT"]), np.ones(256), float(g["mu1"]), 0.0,
                           float(g["mu2"]), return_iters=True)
    assert rel_l2(U, g["U"]) < 1e-13
    assert np.array_equal(it, g["iters"])
    assert int(it.sum()) == 840          # 840 Newton-steps
    h, itc = bc.fom_run(g["X"], np.ones(256), float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]))
    assert rel_l2(h[0].T, g["U"]) < 1e-13 and np.array_equal(itc[0], g["iters"])


def test_fom_config2_n1024_live_reference():
    g = load_golden("fom_n1024.npz")
    mus = g["mus"]
    h, itc = bc.fom_run(g["X"], np.ones(1024), mus[:, 0], mus[:, 1], float(g["At"]), int(g["nT"]))
    for b in range(len(mus)):
        assert rel_l2(h[b].T, g["U"][b]) < 1e-13
        assert np.array_equal(itc[b], g["iters"][b])
    U, it = br.fom_burgers(g["X"], float(g["At"]), int(g["nT"]), np.ones(1024), mus[0, 0], 0.0, mus[0, 1],
                           return_iters=True)
    assert rel_l2(U, g["U"][0]) < 1e-13 and np.array_equal(it, g["iters"][0])


@pytest.mark.parametrize("key,mu1,mu2", [("4.250_0.0150", 4.25, 0.015), ("5.500_0.0300", 5.5, 0.03),
                                         ("4.750_0.0200", 4.75, 0.02), ("6.200_0.0400", 6.2, 0.04)])
def test_fom_committed_snapshots_n512(key, mu1, mu2):
    """The reference's committed (512, 501) snapshots, first 21 columns and 9 column slices."""
    g = load_golden("committed_fom_n512.npz")
    X = np.linspace(0, 100, 512)
    h, _ = bc.fom_run(X, np.ones(512), mu1, mu2, 0.05, 500)
    U = h[0].T
    assert rel_l2(U[:, :21], g["first21_" + key]) < 1e-12
    assert rel_l2(U[:, g["cols"]], g["U_" + key]) < 1e-10
    Un = br.fom_burgers(X, 0.05, 20, np.ones(512), mu1, 0.0, mu2)
    assert rel_l2(Un, g["first21_" + key]) < 1e-13


def test_pod_truncation_rule_and_committed_prom():
    g = load_golden("committed_pod_r40.npz")
    Ks = [br.n_modes_for_tolerance(g["s_all"], e) for e in g["eps2"]]
    assert Ks == list(g["K_expected"])           # 9, 40, 96, 160, 227
    X = np.linspace(0, 100, 512)
    for tag, proj in (("galerkin", "Galerkin"), ("lspg", "LSPG")):
        U = br.pod_prom_burgers(X, 0.05, 12, np.ones(512), 4.75, 0.0, 0.02, g["Phi"], projection=proj)
        assert rel_l2(U, g["first13_" + tag]) < 1e-11


def test_pod_live_reference():
    g = load_golden("pod_live_r40.npz")
    Phi = load_golden("committed_pod_r40.npz")["Phi"]
    X = np.linspace(0, 100, 512)
    for proj in ("Galerkin", "LSPG"):
        U, it = br.pod_prom_burgers(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0,
                                    float(g["mu2"]), Phi, projection=proj, return_iters=True)
        assert rel_l2(U, g["U_" + proj]) < 1e-11
        assert np.array_equal(it, g["iters_" + proj])
    with pytest.raises(ValueError):
        br.pod_prom_burgers(X, 0.05, 1, np.ones(512), 5.0, 0.0, 0.02, Phi, projection="lspg")


def test_quadratic_helpers_and_committed():
    g = load_golden("quadratic_live_n21.npz")
    assert np.array_equal(br.get_sym(g["q"]), g["sym"])
    assert np.array_equal(br.get_dQ_dq(g["q"]), g["dQ"])
    c = load_golden("committed_quadratic_n21.npz")
    X = np.linspace(0, 100, 512)
    for proj in ("Galerkin", "LSPG"):
        U, it = br.pod_quadratic_manifold(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0,
                                          float(g["mu2"]), c["Phi"], c["H"], projection=proj, return_iters=True)
        assert rel_l2(U, g["U_" + proj]) < 1e-10
        assert np.array_equal(it, g["iters_" + proj])
    U = br.pod_quadratic_manifold(X, 0.05, 6, np.ones(512), float(c["mu1"]), 0.0, float(c["mu2"]), c["Phi"], c["H"])
    assert rel_l2(U, c["first7"]) < 1e-10


def test_ann_forward_jacobian_and_prom():
    g = load_golden("ann_n5.npz")
    Ws = [g[f"W{i}"] for i in range(6)]
    bs = [g[f"b{i}"] for i in range(6)]
    for i in range(len(g["qp"])):
        f = br.mlp_forward(Ws, bs, g["qp"][i])
        J = br.mlp_jacobian(Ws, bs, g["qp"][i])
        assert np.abs(f - g["fwd"][i]).max() <= 2e-5 * max(1.0, np.abs(g["fwd"][i]).max())
        assert np.abs(J - g["jac"][i]).max() <= 2e-4 * max(1.0, np.abs(g["jac"][i]).max())
    X = np.linspace(0, 100, 512)
    U, it = br.pod_ann_prom(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0, float(g["mu2"]),
                            g["U_p"], g["U_s"], Ws, bs, return_iters=True)
    # the reference evaluates the MLP and its Jacobian in fp32: parity is fp32-limited
    assert rel_l2(U, g["U"]) < 5e-6


def test_nonintrusive_decoder():
    g = load_golden("nonintrusive_decoder.npz")
    Ws = [g[f"{i}_weight"] for i in (0, 2, 4, 6)]
    bs = [g[f"{i}_bias"] for i in (0, 2, 4, 6)]
    U = br.predict_on_fom_grid(float(g["mu1"]), float(g["mu2"]), int(g["Nt"]), g["U_modes"], Ws, bs, g["mean"], g["std"])
    assert U.shape == (512, 501)
    assert rel_l2(U[:, g["cols"]], g["Uhat_cols"]) < 1e-5          # fp32 MLP


def test_fd_newton_live_and_committed():
    g = load_golden("fd_newton.npz")
    for tag in ("n128", "n512"):
        N, dt, nT, mu1, mu2 = g["par_" + tag]
        U, it = br.fd_newton(0.0, 100.0, int(N), dt, int(nT), np.ones(int(N)), mu1, mu2, return_iters=True)
        assert rel_l2(U, g["U_" + tag]) < 1e-14 and np.array_equal(it, g["iters_" + tag])
    U, it = br.fd_newton(0.0, 100.0, 64, 0.1, 5, np.ones(64), 4.7, 0.02, return_iters=True, use_fd_jacobian=True)
    assert rel_l2(U, g["U_fdjac_n64"]) < 1e-13 and np.array_equal(it, g["iters_fdjac_n64"])     # dense FD Jacobian (:46-57)
    U = br.fd_newton(0.0, 100.0, 512, 0.05, 10, np.ones(512), 4.25, 0.015)
    assert rel_l2(U, g["committed_first11"]) < 1e-14        # FD/fd_training_data (committed by the reference)


def test_pod_rbf_prom_live_reference():
    g = load_golden("rbf_n17.npz")
    X = np.linspace(0, 100, 512)
    for kernel, proj in (("gaussian", "LSPG"), ("imq", "Galerkin")):
        args = (g["X_train"], g["W_" + kernel], float(g["eps_" + kernel]), kernel, g["x_min"], g["x_max"], g["y_min"], g["y_max"])
        assert np.abs(br.rbf_value(g["qp_" + kernel], *args) - g["val_" + kernel]).max() < 1e-12
        assert np.abs(br.rbf_jacobian(g["qp_" + kernel], *args) - g["jac_" + kernel]).max() < 1e-10 * np.abs(g["jac_" + kernel]).max()
        U, it = br.pod_rbf_prom(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0, float(g["mu2"]),
                                g["U_p"], g["U_s"], g["X_train"], g["W_" + kernel], float(g["eps_" + kernel]),
                                g["x_min"], g["x_max"], g["y_min"], g["y_max"], projection=proj, kernel=kernel,
                                max_newton=20, return_iters=True)
        assert rel_l2(U, g["U_" + kernel]) < 1e-11 and np.array_equal(it, g["iters_" + kernel])
    with pytest.raises(ValueError):
        br.pod_rbf_prom(X, 0.05, 1, np.ones(512), 4.75, 0.0, 0.02, g["U_p"], g["U_s"], g["X_train"], g["W_imq"], 1.0,
                        g["x_min"], g["x_max"], g["y_min"], g["y_max"], kernel="multiquadric")


def test_local_prom_live_reference():
    g = load_golden("local_pod.npz")
    X = np.linspace(0, 100, 512)
    bases = {c: g[f"basis{c}"] for c in range(4)}
    for proj in ("Galerkin", "LSPG"):
        U, it, cl = br.local_prom_burgers(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0,
                                          float(g["mu2"]), g["centers"], bases, g["U_global"], 12, projection=proj,
                                          return_iters=True)
        assert rel_l2(U[:, ::int(g["stride"])], g["U_" + proj]) < 1e-12 and np.array_equal(it, g["iters_" + proj])
        assert len(np.unique(cl)) >= 2                     # the run really switches bases


def test_ann_reduced_system_needs_the_pivot_search():
    """Why bg_ann_rom_run always eliminates with a pivot search (and bg_rom_run does not): on the reference's committed
    closure the first-iteration reduced systems of pod_ann_prom have sub-diagonal multipliers above 1 in the natural
    order (LAPACK's getrf leaves the diagonal), while a POD basis gives multipliers far below 1."""
    g = load_golden("ann_n5.npz")
    X = np.linspace(0.0, 100.0, 512)
    Ws = [g[f"W{i}"] for i in range(6)]; bs = [g[f"b{i}"] for i in range(6)]
    U0 = np.ones(512)
    M3 = br.mass_tridiag(X); K3 = br.diffusion_tridiag(X)
    mu1, mu2, At = 4.75, 0.02, 0.05
    F = br.forcing_vector(X, mu2)
    C3 = br.convection_tridiag(X, U0)
    S = br.supg_term(X, U0, mu2)
    lo, di, up = br.system_tridiag(M3, K3, C3, At, 0.0)
    b = br.tridiag_matvec(*M3, U0) + At * F - At * S
    b[0] = mu1
    R = br.tridiag_matvec(lo, di, up, U0) - b

    def worst_multiplier(A):
        A = A.copy(); n = len(A); worst = 0.0
        for k in range(n - 1):
            m = A[k + 1:, k] / A[k, k]
            worst = max(worst, np.abs(m).max())
            A[k + 1:] -= np.outer(m, A[k])
        return worst

    q_p = g["U_p"].T @ U0
    dN = br.mlp_jacobian(Ws, bs, q_p.astype(np.float32)).astype(np.float64)
    W = g["U_p"] + g["U_s"] @ dN
    for proj in ("lspg", "galerkin"):
        Ar, _ = br._reduce(lo, di, up, R, W, proj)
        assert worst_multiplier(Ar) > 1.0, proj
    Phi = load_golden("committed_pod_r40.npz")["Phi"]
    for proj in ("lspg", "galerkin"):
        Ar, _ = br._reduce(lo, di, up, R, Phi, proj)
        assert worst_multiplier(Ar) < 1.0, proj
