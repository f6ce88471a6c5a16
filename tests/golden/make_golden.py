#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

Runs ONLY in the build container, where the reference checkout is mounted at
/root/reference.  It (a) imports the reference's own ``FEM/fem_burgers.py`` and
records input/output vectors of the hot-path methods at small sizes, and (b)
slices a few columns out of the ``.npy`` result files the reference repository
commits.  Only data (inputs, expected outputs) is written; no reference source
travels.  The fixtures are what pins ``oracle/burgers_ref.py`` (see
tests/test_oracle_golden.py) and, through it, the HIP path.

Usage:  python tests/golden/make_golden.py [--only NAME ...]
"""
import argparse
import contextlib
import io
import os
import sys
import zipfile

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
COLS = np.array([0, 1, 2, 5, 10, 50, 100, 250, 500])


def ref_solver():
    sys.path.insert(0, os.path.join(REF, "FEM"))
    import fem_burgers  # the reference module
    return fem_burgers


def mesh(n_nodes, a=0.0, b=100.0):
    m = n_nodes - 1
    X = np.linspace(a, b, m + 1)
    T = np.array([np.arange(1, m + 1), np.arange(2, m + 2)]).T
    return X, T


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def iters_from_log(log, step_marker="Time Step:", iter_marker="Iteration:"):
    counts, cur = [], None
    for line in log.splitlines():
        s = line.strip()
        if s.startswith(step_marker):
            if cur is not None:
                counts.append(cur)
            cur = 0
        elif s.startswith(iter_marker) and cur is not None:
            cur += 1
    if cur is not None:
        counts.append(cur)
    return np.array(counts, dtype=np.int32)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ---------------------------------------------------------------- fixtures
def fx_fom_n256():
    """BASELINE config 1: N=256, dt=0.05, 100 steps, mu=(4.75, 0.02)."""
    fb = ref_solver()
    X, T = mesh(256)
    mu1, mu2, At, nT = 4.75, 0.02, 0.05, 100
    U, log = quiet(fb.FEMBurgers(X, T).fom_burgers, At, nT, np.ones_like(X), mu1, 0.0, mu2)
    save("fom_n256.npz", X=X, At=At, nT=nT, mu1=mu1, mu2=mu2, E=0.0, U=U, iters=iters_from_log(log))


def fx_fom_n1024():
    """BASELINE config 2 slice: N=1024, dt=0.025, 2 mu x 20 steps."""
    fb = ref_solver()
    X, T = mesh(1024)
    At, nT = 0.025, 20
    mus = np.array([[5.5, 0.03], [4.31, 0.0171]])
    Us, its = [], []
    for mu1, mu2 in mus:
        U, log = quiet(fb.FEMBurgers(X, T).fom_burgers, At, nT, np.ones_like(X), mu1, 0.0, mu2)
        Us.append(U); its.append(iters_from_log(log))
    save("fom_n1024.npz", X=X, At=At, nT=nT, mus=mus, E=0.0, U=np.stack(Us), iters=np.stack(its))


def fx_fom_general():
    """Non-uniform mesh, E != 0, non-constant u0: pins the assemblers in general position."""
    fb = ref_solver()
    rng = np.random.default_rng(20251121)
    n = 96
    X = np.sort(np.concatenate([[0.0, 100.0], rng.uniform(0, 100, n - 2)]))
    # keep elements from degenerating
    X = np.linspace(0, 100, n) + rng.uniform(-0.3, 0.3, n) * (100 / (n - 1))
    X[0], X[-1] = 0.0, 100.0
    T = np.array([np.arange(1, n), np.arange(2, n + 1)]).T
    fem = fb.FEMBurgers(X, T)
    u = 1.0 + 3.0 * rng.random(n)
    mu2 = 0.0237
    M = fem.compute_mass_matrix().toarray()
    K = fem.compute_diffusion_matrix().toarray()
    C = fem.compute_convection_matrix(u).toarray()
    F = fem.compute_forcing_vector(mu2)
    S = fem.compute_supg_term(u, mu2)
    # one slow-flow vector that exercises the eps_vel branch of tau_e
    u_small = np.where(np.arange(n) % 7 == 0, 1e-12, u)
    u_small[1::7] = -1e-12
    S_small = fem.compute_supg_term(u_small, mu2)
    At, nT, mu1, E = 0.02, 12, 3.3, 0.05
    u0 = 1.0 + 0.5 * np.sin(X / 100 * np.pi)
    U, log = quiet(fem.fom_burgers, At, nT, u0, mu1, E, mu2)
    save("fom_general.npz", X=X, u=u, mu2=mu2, M=M, K=K, C=C, F=F, S=S, u_small=u_small,
         S_small=S_small, At=At, nT=nT, mu1=mu1, E=E, u0=u0, U=U, iters=iters_from_log(log))


def fx_committed_fom():
    """Column slices of committed (512, 501) FOM snapshots (dt=0.05, 500 steps, E=0)."""
    files = {
        "4.250_0.0150": "FEM/fem_training_data/fem_simulation_mu1_4.250_mu2_0.0150.npy",
        "5.500_0.0300": "FEM/fem_training_data/fem_simulation_mu1_5.500_mu2_0.0300.npy",
        "4.750_0.0200": "FEM/fem_testing_data/fem_simulation_mu1_4.750_mu2_0.0200.npy",
        "6.200_0.0400": "FEM/fem_testing_data/fem_simulation_mu1_6.200_mu2_0.0400.npy",
    }
    out = {"cols": COLS}
    for k, f in files.items():
        a = np.load(os.path.join(REF, f))
        assert a.shape == (512, 501)
        out["U_" + k] = a[:, COLS]
        out["first21_" + k] = a[:, :21]
    save("committed_fom_n512.npz", **out)


def fx_committed_pod():
    """Committed POD basis (tol 1e-03 -> 40 modes), singular values and PROM slices."""
    Phi = np.load(os.path.join(REF, "POD/modes/U_modes_tol_1e-03.npy"))
    s_all = np.load(os.path.join(REF, "POD/modes/Singular_values_modes_tol_0e+00.npy"))
    out = {"Phi": Phi, "s_all": s_all, "cols": COLS,
           "K_expected": np.array([9, 40, 96, 160, 227]),
           "eps2": np.array([1e-2, 1e-3, 1e-4, 1e-5, 1e-6])}
    for tag in ("galerkin", "lspg"):
        a = np.load(os.path.join(REF, f"POD/Results_thesis/rom_solutions/U_PROM_tol_1e-03_mu1_4.750_mu2_0.0200_{tag}.npy"))
        out["U_" + tag] = a[:, COLS]
        out["first13_" + tag] = a[:, :13]
    save("committed_pod_r40.npz", **out)


def fx_pod_live():
    """Live reference run of pod_prom_burgers, r=40, 6 steps, both projections (iteration counts)."""
    fb = ref_solver()
    X, T = mesh(512)
    Phi = np.load(os.path.join(REF, "POD/modes/U_modes_tol_1e-03.npy"))
    out = {}
    for proj in ("Galerkin", "LSPG"):
        U, log = quiet(fb.FEMBurgers(X, T).pod_prom_burgers, 0.05, 6, np.ones_like(X), 5.19, 0.0, 0.026,
                       Phi, projection=proj)
        out["U_" + proj] = U
        out["iters_" + proj] = iters_from_log(log)
    save("pod_live_r40.npz", At=0.05, nT=6, mu1=5.19, mu2=0.026, **out)


def fx_committed_quadratic():
    Phi = np.load(os.path.join(REF, "Quadratic_manifold/Phi.npy"))
    H = np.ascontiguousarray(np.load(os.path.join(REF, "Quadratic_manifold/H.npy")))
    a = np.load(os.path.join(REF, "Quadratic_manifold/quadratic_rom_solutions/"
                                  "quadratic_PROM_U_PROM_21_modes_mu1_5.190_mu2_0.0260.npy"))
    save("committed_quadratic_n21.npz", Phi=Phi, H=H, cols=COLS, U=a[:, COLS], first7=a[:, :7],
         mu1=5.19, mu2=0.026)


def fx_quadratic_live():
    """Live run, both projections, 3 steps; plus get_sym / get_dQ_dq vectors."""
    fb = ref_solver()
    X, T = mesh(512)
    Phi = np.load(os.path.join(REF, "Quadratic_manifold/Phi.npy"))
    H = np.load(os.path.join(REF, "Quadratic_manifold/H.npy"))
    out = {}
    for proj in ("Galerkin", "LSPG"):
        U, log = quiet(fb.FEMBurgers(X, T).pod_quadratic_manifold, 0.05, 3, np.ones_like(X), 4.56, 0.0, 0.019,
                       Phi, H, projection=proj)
        out["U_" + proj] = U
        out["iters_" + proj] = np.array(
            [blk.count("Newton ") - blk.count("Warning") for blk in log.split("=== time step")[1:]], dtype=np.int32)
    rng = np.random.default_rng(7)
    q = rng.standard_normal(6)
    save("quadratic_live_n21.npz", At=0.05, nT=3, mu1=4.56, mu2=0.019, q=q, sym=fb.get_sym(q),
         dQ=fb.get_dQ_dq(6, q), **out)


def _ann_weights():
    """Raw float32 storages out of the checkpoint zip (no unpickling)."""
    z = zipfile.ZipFile(os.path.join(REF, "POD-ANN/pod_ann_model.pth"))
    dims = [5, 32, 64, 128, 256, 256, 91]
    Ws, bs = [], []
    for layer in range(6):
        w = np.frombuffer(z.read(f"pod_ann_model/data/{2 * layer}"), dtype="<f4")
        b = np.frombuffer(z.read(f"pod_ann_model/data/{2 * layer + 1}"), dtype="<f4")
        Ws.append(w.reshape(dims[layer + 1], dims[layer]).copy())
        bs.append(b.copy())
    return Ws, bs


def fx_ann():
    """MLP weights (raw storages), U_p/U_s, live reference pod_ann_prom (4 steps), torch fwd/jacobian vectors."""
    import torch
    import torch.nn as nn
    fb = ref_solver()
    Ws, bs = _ann_weights()

    class POD_ANN(nn.Module):           # 5-32-64-128-256-256-91 ELU MLP, declared fresh
        def __init__(self):
            super().__init__()
            dims = [5, 32, 64, 128, 256, 256, 91]
            self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(6)])
            self.act = nn.ELU()

        def forward(self, x):
            for i, l in enumerate(self.layers):
                x = l(x)
                if i < 5:
                    x = self.act(x)
            return x

    model = POD_ANN()
    with torch.no_grad():
        for l, W, b in zip(model.layers, Ws, bs):
            l.weight.copy_(torch.from_numpy(W)); l.bias.copy_(torch.from_numpy(b))
    model.eval()
    U_p = np.load(os.path.join(REF, "POD-ANN/U_p.npy"))
    U_s = np.load(os.path.join(REF, "POD-ANN/U_s.npy"))
    X, T = mesh(512)
    fem = fb.FEMBurgers(X, T)
    U, log = quiet(fem.pod_ann_prom, 0.05, 4, np.ones_like(X), 4.56, 0.0, 0.019, U_p, U_s, model)
    iters = np.array([blk.count("Newton") for blk in log.split("Time step")[1:]], dtype=np.int32)
    # forward / jacobian vectors on a snapshot-like reduced state
    snap = np.load(os.path.join(REF, "FEM/fem_testing_data/fem_simulation_mu1_4.750_mu2_0.0200.npy"))[:, [0, 3, 77, 400]]
    qp = (U_p.T @ snap).T.astype(np.float32)                    # (4, 5)
    with torch.no_grad():
        fwd = model(torch.from_numpy(qp)).numpy()
    jac = np.stack([fem.compute_ann_jacobian(model, torch.from_numpy(qp[i:i + 1])).numpy() for i in range(4)])
    committed = np.load(os.path.join(REF, "POD-ANN/pod_ann_prom_solutions/POD_ANN_PROM_U_n5_nb91_mu1_4.560_mu2_0.0190.npy"))
    arrs = {f"W{i}": W for i, W in enumerate(Ws)}
    arrs.update({f"b{i}": b for i, b in enumerate(bs)})
    save("ann_n5.npz", U_p=U_p, U_s=U_s, At=0.05, nT=4, mu1=4.56, mu2=0.019, U=U, iters=iters,
         qp=qp, fwd=fwd, jac=jac, committed_shape=np.array(committed.shape),
         committed_first5=committed[:, :5], **arrs)


def fx_nonintrusive():
    """Decoder-only POD-ANN (Non-Instrusive/predict_pod_ann.py:73-80): state_dict weights (safe
    loader), standardiser, modes, and the reference's own prediction for one test point."""
    import torch
    sys.path.insert(0, os.path.join(REF, "Non-Instrusive"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "Non-Instrusive"))
    try:
        import predict_pod_ann as ppa
        U_modes = np.load(ppa.UMODES_PATH)
        sc = np.load(ppa.SCALER_Z_NPZ)
        mean, std = sc["mean"], sc["std"]
        state = torch.load(ppa.MODEL_PT, map_location="cpu", weights_only=True)
        model = ppa.make_mlp(3, U_modes.shape[1], [32, 64, 128], "elu", 0.0)
        model.load_state_dict(state)
        model.eval()
        U_FOM = np.load(os.path.join(REF, "FEM/fem_testing_data/fem_simulation_mu1_4.750_mu2_0.0200.npy"))
        Uhat = ppa.predict_on_fom_grid(4.75, 0.02, U_modes, model, mean, std, U_FOM)
    finally:
        os.chdir(cwd)
    arrs = {k.replace(".", "_"): v.numpy() for k, v in state.items()}
    save("nonintrusive_decoder.npz", U_modes=U_modes, mean=mean, std=std, mu1=4.75, mu2=0.02, Nt=U_FOM.shape[1],
         Uhat_cols=Uhat[:, COLS], cols=COLS, rel_err_vs_fom=np.linalg.norm(U_FOM - Uhat) / np.linalg.norm(U_FOM),
         **arrs)


def fx_fd():
    """FD true-Newton stepper (FD/fd_burgers.py): live runs + slices of the committed training data."""
    sys.path.insert(0, os.path.join(REF, "FD"))
    import fd_burgers
    out = {}
    for tag, (N, dt, nT, mu1, mu2) in {"n128": (128, 0.1, 12, 4.8, 0.021), "n512": (512, 0.05, 8, 5.5, 0.03)}.items():
        fd = fd_burgers.FDBurgers(0.0, 100.0, N)
        U, log = quiet(fd.fom_burgers_newton, dt, nT, np.ones(N), mu1, mu2)
        its = np.array([blk.count("relative update") for blk in log.split("Time step")[1:]], dtype=np.int32)
        out.update({f"U_{tag}": U, f"iters_{tag}": its, f"par_{tag}": np.array([N, dt, nT, mu1, mu2])})
    # the finite-difference Jacobian option (FD/fd_burgers.py:46-57), a small case: the matrix is dense
    fd = fd_burgers.FDBurgers(0.0, 100.0, 64)
    U, log = quiet(fd.fom_burgers_newton, 0.1, 5, np.ones(64), 4.7, 0.02, use_fd_jacobian=True)
    out["U_fdjac_n64"] = U
    out["iters_fdjac_n64"] = np.array([blk.count("relative update") for blk in log.split("Time step")[1:]], dtype=np.int32)
    c = np.load(os.path.join(REF, "FD/fd_training_data/fd_simulation_mu1_4.250_mu2_0.0150.npy"))
    out["committed_first11"] = c[:, :11]
    out["committed_cols"] = c[:, COLS]
    save("fd_newton.npz", cols=COLS, **out)


def fx_rbf():
    """POD-RBF PROM (pod_rbf_prom :1278-1398).  The committed closure has 4501 centres (3.5 MB of text
    artefacts), so the fixture carries a 300-centre Gaussian/IMQ closure fitted HERE on a subsample of the
    reference's own training coordinates (Q_train / Qbar_train) and the reference's outputs for it."""
    fb = ref_solver()
    d = os.path.join(REF, "POD-RBF/rbf_training_simple")
    U_p = np.load(os.path.join(d, "Phi_primary.npy")); U_s = np.load(os.path.join(d, "Phi_secondary.npy"))
    Q = np.load(os.path.join(d, "Q_train.npy")); Qb = np.load(os.path.join(d, "Qbar_train.npy"))
    if Q.shape[0] != 4501:
        Q, Qb = Q.T, Qb.T
    idx = np.linspace(0, Q.shape[0] - 1, 300).astype(int)
    x_min, x_max = Q.min(0), Q.max(0); y_min, y_max = Qb.min(0), Qb.max(0)
    Xs = 2.0 * (Q[idx] - x_min) / (x_max - x_min) - 1.0
    Ys = 2.0 * (Qb[idx] - y_min) / (y_max - y_min) - 1.0
    X, T = mesh(512)
    out = dict(U_p=U_p, U_s=U_s, X_train=Xs, x_min=x_min, x_max=x_max, y_min=y_min, y_max=y_max)
    r = np.linalg.norm(Xs[:, None, :] - Xs[None, :, :], axis=2)
    for kernel, eps, proj in (("gaussian", 2.0, "LSPG"), ("imq", 1.5, "Galerkin")):
        Kmat = np.exp(-(eps * r) ** 2) if kernel == "gaussian" else 1.0 / np.sqrt(1.0 + (eps * r) ** 2)
        W = np.linalg.solve(Kmat + 1e-8 * np.eye(len(idx)), Ys)
        fem = fb.FEMBurgers(X, T)
        U, log = quiet(fem.pod_rbf_prom, 0.05, 4, np.ones(512), 4.75, 0.0, 0.02, U_p, U_s, Xs, W, eps,
                       x_min, x_max, y_min, y_max, projection=proj, kernel=kernel, tol_newton=1e-6, max_newton=20)
        its = np.array([blk.count("Newton it=") for blk in log.split("Time Step:")[1:]], dtype=np.int32)
        qp = U_p.T @ U[:, 2]
        out.update({f"W_{kernel}": W, f"eps_{kernel}": eps, f"U_{kernel}": U, f"iters_{kernel}": its,
                    f"val_{kernel}": fb.interpolate_with_rbf_scaled(qp, Xs, W, eps, kernel, x_min, x_max, y_min, y_max),
                    f"jac_{kernel}": fb.compute_rbf_jacobian_full(qp, Xs, W, eps, kernel, x_min, x_max, y_min, y_max),
                    f"qp_{kernel}": qp})
    save("rbf_n17.npz", At=0.05, nT=4, mu1=4.75, mu2=0.02, **out)


def fx_committed_pod_r96():
    """Committed 96-mode basis (tol 1e-04) and the first columns of its committed PROM outputs: pins the
    large-r path (r beyond the register-resident MFMA kernels)."""
    Phi = np.load(os.path.join(REF, "POD/modes/U_modes_tol_1e-04.npy"))
    out = {"Phi": Phi}
    for tag in ("galerkin", "lspg"):
        a = np.load(os.path.join(REF, f"POD/Results_thesis/rom_solutions/U_PROM_tol_1e-04_mu1_4.750_mu2_0.0200_{tag}.npy"))
        out["first9_" + tag] = a[:, :9]
    save("committed_pod_r96.npz", **out)


def fx_local_pod():
    """Local POD PROM (local_prom_burgers :979-1079).  The committed cluster artefacts are pickles (never
    loaded), so the fixture builds 4 clusters HERE from the reference's committed snapshots (nearest-centre
    clustering in the first 12 global POD coordinates, one overlapping local basis per cluster) and records
    the reference's output for them through a minimal stand-in for the fitted KMeans object."""
    fb = ref_solver()
    d = os.path.join(REF, "FEM/fem_training_data")
    files = sorted(f for f in os.listdir(d) if f.startswith("fem_simulation_") and f.endswith(".npy"))
    S = np.hstack([np.load(os.path.join(d, f))[:, ::4] for f in files])          # (512, 9*126)
    Ug = np.linalg.svd(S, full_matrices=False)[0][:, :12]
    Q = (Ug.T @ S).T
    rng = np.random.default_rng(3)
    centers = Q[rng.choice(len(Q), 4, replace=False)].copy()
    for _ in range(25):                                                          # plain Lloyd iterations
        lab = np.argmin(((Q[:, None, :] - centers[None]) ** 2).sum(2), 1)
        centers = np.stack([Q[lab == c].mean(0) if np.any(lab == c) else centers[c] for c in range(4)])
    lab = np.argmin(((Q[:, None, :] - centers[None]) ** 2).sum(2), 1)
    d2 = ((Q[:, None, :] - centers[None]) ** 2).sum(2)
    bases = {}
    for c in range(4):
        member = (lab == c) | (d2[:, c] < 1.5 * d2.min(1))                        # overlap with neighbours
        Uc = np.linalg.svd(S[:, member], full_matrices=False)[0]
        bases[c] = np.ascontiguousarray(Uc[:, :[14, 22, 30, 18][c]])

    class NearestCentre:                                                          # kmeans.predict stand-in
        cluster_centers_ = centers
        def predict(self, q):
            return np.argmin(((np.atleast_2d(q)[:, None, :] - centers[None]) ** 2).sum(2), 1)

    X, T = mesh(512)
    out = {"centers": centers, "U_global": Ug}
    out.update({f"basis{c}": bases[c] for c in range(4)})
    for proj in ("Galerkin", "LSPG"):
        U, log = quiet(fb.FEMBurgers(X, T).local_prom_burgers, 0.05, 150, np.ones(512), 4.9, 0.0, 0.022, NearestCentre(),
                       bases, Ug, 12, projection=proj)
        out["U_" + proj] = U[:, ::5]
        out["iters_" + proj] = iters_from_log(log)
    save("local_pod.npz", At=0.05, nT=150, stride=5, mu1=4.9, mu2=0.022, **out)


FIXTURES = {
    "fom_n256": fx_fom_n256, "fom_n1024": fx_fom_n1024, "fom_general": fx_fom_general,
    "committed_fom": fx_committed_fom, "committed_pod": fx_committed_pod, "pod_live": fx_pod_live,
    "committed_quadratic": fx_committed_quadratic, "quadratic_live": fx_quadratic_live, "ann": fx_ann,
    "nonintrusive": fx_nonintrusive, "fd": fx_fd, "rbf": fx_rbf, "committed_pod_r96": fx_committed_pod_r96,
    "local_pod": fx_local_pod,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    for name, fn in FIXTURES.items():
        if args.only and name not in args.only:
            continue
        print(f"[{name}]")
        fn()
