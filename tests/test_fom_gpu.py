"""GPU parity of the fused FOM path (through the C ABI) against the oracle, the golden
fixtures and size-independent properties.  Tolerance: BASELINE.json north_star, rel-L2 <= 1e-10
(fp64); iteration counts must match exactly."""
import numpy as np
import pytest
import torch

from conftest import load_golden, mesh, rel_l2
from oracle import burgers_ref as br
from oracle import burgers_ref_c as bc

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _run(hip, X, u0, mu1, mu2, dt, nsteps, **kw):
    """fom_run in the form the library picks AND, where both forms exist (64 < N <= 1536: one wavefront or one workgroup
    per sample), in each of them explicitly: they must agree (same iteration counts; the norms are summed in a different order, hence 1e-12 not bitwise).
    Returns the library's own choice."""
    from burgers_hip import fom
    res = fom.fom_run(X, u0, mu1, mu2, dt, nsteps, **kw)
    torch.cuda.synchronize()
    out = res.hist.cpu().numpy(), res.iters.cpu().numpy(), res.flags.cpu().numpy()
    if 64 < len(X) <= 1536 and "form" not in kw and "options" not in kw and not kw.get("trace"):
        for form in ("wide", "wave"):
            r2 = fom.fom_run(X, u0, mu1, mu2, dt, nsteps, form=form, **kw)
            torch.cuda.synchronize()
            assert np.array_equal(r2.iters.cpu().numpy(), out[1]) and np.array_equal(r2.flags.cpu().numpy(), out[2]), form
            h2 = r2.hist.cpu().numpy()
            fin = np.isfinite(out[0]) & np.isfinite(h2)
            assert np.array_equal(np.isfinite(out[0]), np.isfinite(h2)), form
            assert np.abs(h2[fin] - out[0][fin]).max() <= 1e-12 * max(1.0, np.abs(out[0][fin]).max()), form
    return out


def test_cross_lane_primitives_via_tridiag_solve(hip):
    """Random well-conditioned non-symmetric tridiagonal systems of every supported width."""
    from burgers_hip import fom
    rng = np.random.default_rng(1)
    for N in (2, 3, 64, 65, 100, 128, 256, 300, 512, 513, 600, 700, 1000, 1024):
        B = 5
        lo = rng.uniform(-0.2, 0.1, (B, N)); up = rng.uniform(-0.1, 0.3, (B, N))
        di = 0.45 + rng.uniform(0, 0.3, (B, N))
        lo[:, 0] = 0; up[:, -1] = 0
        rhs = rng.standard_normal((B, N))
        sol = fom.tridiag_solve(*[torch.tensor(a, device="cuda") for a in (lo, di, up, rhs)]).cpu().numpy()
        for b in range(B):
            ref = br.tridiag_solve(lo[b], di[b], up[b], rhs[b])
            assert rel_l2(sol[b], ref) < 1e-12, f"N={N}"


@pytest.mark.parametrize("N", [256, 512, 1024, 96, 513])
def test_assembly_matches_oracle(hip, N):
    """Pins compute_convection_matrix / compute_supg_term / compute_forcing_vector / A,b,R."""
    from burgers_hip import fom
    rng = np.random.default_rng(N)
    X, _ = mesh(N)
    B = 6
    uk = 1.0 + 4.0 * rng.random((B, N)); un = 1.0 + 4.0 * rng.random((B, N))
    uk[1, ::5] = 1e-13; uk[1, 1::5] = -1e-13           # eps_vel branch of tau_e
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt, E = 0.03, 0.02
    lo, di, up, rhs = [t.cpu().numpy() for t in fom.fom_assemble(X, uk, un, mu1, mu2, dt, E=E)]
    M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
    for b in range(B):
        C3 = br.convection_tridiag(X, uk[b])
        l, d, u = br.system_tridiag(M3, K3, C3, dt, E)
        bb = br.tridiag_matvec(*M3, un[b]) + dt * br.forcing_vector(X, mu2[b]) - dt * br.supg_term(X, uk[b], mu2[b])
        bb[0] = mu1[b]
        r = bb - br.tridiag_matvec(l, d, u, uk[b])
        scale = np.abs(d).max()
        assert np.abs(lo[b] - l).max() < 1e-13 * scale
        assert np.abs(di[b] - d).max() < 1e-13 * scale
        assert np.abs(up[b] - u).max() < 1e-13 * scale
        assert np.abs(rhs[b] - r).max() < 1e-12 * max(1.0, np.abs(r).max())


def test_config1_n256_golden(hip):
    g = load_golden("fom_n256.npz")
    h, it, fl = _run(hip, g["X"], np.ones(256), float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]))
    assert rel_l2(h[0].T, g["U"]) < TOL
    assert np.array_equal(it[0], g["iters"]) and int(it.sum()) == 840


def test_config2_n1024_golden(hip):
    g = load_golden("fom_n1024.npz")
    mus = g["mus"]
    h, it, fl = _run(hip, g["X"], np.ones(1024), mus[:, 0], mus[:, 1], float(g["At"]), int(g["nT"]))
    for b in range(2):
        assert rel_l2(h[b].T, g["U"][b]) < TOL
        assert np.array_equal(it[b], g["iters"][b])
    assert (fl & 2 == 0).all()


def test_committed_snapshots_n512_full_run(hip):
    """500 steps at the thesis setting against column slices of the reference's committed files."""
    g = load_golden("committed_fom_n512.npz")
    keys = [("4.250_0.0150", 4.25, 0.015), ("5.500_0.0300", 5.5, 0.03), ("4.750_0.0200", 4.75, 0.02),
            ("6.200_0.0400", 6.2, 0.04)]
    X, _ = mesh(512)
    h, it, fl = _run(hip, X, np.ones(512), [k[1] for k in keys], [k[2] for k in keys], 0.05, 500)
    for b, (key, _, _) in enumerate(keys):
        U = h[b].T
        assert rel_l2(U[:, :21], g["first21_" + key]) < TOL
        assert rel_l2(U[:, g["cols"]], g["U_" + key]) < TOL
    assert it[:, 0].tolist() == [20, 20, 20, 20] or it[:, 0].max() <= 20   # first step runs into the cap
    assert (fl & 1).any()


@pytest.mark.parametrize("N,dt,nsteps", [(1024, 0.025, 40), (512, 0.05, 40), (256, 0.05, 40), (513, 0.05, 25),
                                         (96, 0.2, 30), (640, 0.04, 25), (64, 0.3, 20), (700, 0.03, 20)])
def test_batch_vs_c_oracle(hip, N, dt, nsteps):
    """16 seeded samples per size, E != 0 and non-constant u0 included."""
    rng = np.random.default_rng(20251121 + N)
    X, _ = mesh(N)
    B = 16
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    u0 = 1.0 + 0.3 * np.sin(np.outer(rng.uniform(0.5, 2.0, B), X / 100 * np.pi))
    for E in (0.0, 0.01):
        h, it, fl = _run(hip, X, u0, mu1, mu2, dt, nsteps, E=E)
        ho, ito = bc.fom_run(X, u0, mu1, mu2, dt, nsteps, E=E)
        assert rel_l2(h, ho) < TOL
        assert np.array_equal(it, ito)
        for b in range(B):
            assert rel_l2(h[b], ho[b]) < TOL


def test_full_size_config2_properties(hip):
    """BASELINE config 2 shape (B=1024, N=1024, dt=0.025), shortened in time: checks that do
    not need the oracle at full size plus a 16-sample oracle subset."""
    rng = np.random.default_rng(20251121)
    N, B, nsteps, dt = 1024, 1024, 30, 0.025
    X, _ = mesh(N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    h, it, fl = _run(hip, X, np.ones(N), mu1, mu2, dt, nsteps)
    assert np.isfinite(h).all() and (fl & 2 == 0).all()
    assert np.allclose(h[:, 1:, 0], mu1[:, None], rtol=0, atol=1e-12)        # Dirichlet row from step 1 on
    assert (it >= 1).all() and (it <= 20).all()
    # batch-order independence: a permuted batch gives bitwise the same per-sample history
    perm = rng.permutation(B)
    h2, it2, _ = _run(hip, X, np.ones(N), mu1[perm], mu2[perm], dt, nsteps)
    assert np.array_equal(h2, h[perm]) and np.array_equal(it2, it[perm])
    # restart property: running 30 steps == running 10 then 20 from the stored column
    h3, it3, _ = _run(hip, X, h[:, 10, :], mu1, mu2, dt, 20)
    assert np.array_equal(h3, h[:, 10:, :]) and np.array_equal(it3, it[:, 10:])
    sub = rng.choice(B, 16, replace=False)
    ho, ito = bc.fom_run(X, np.ones(N), mu1[sub], mu2[sub], dt, nsteps)
    assert rel_l2(h[sub], ho) < TOL and np.array_equal(it[sub], ito)


def test_edge_cases(hip):
    from burgers_hip import fom, lib
    X, _ = mesh(256)
    # empty batch, zero steps
    r = fom.fom_run(X, np.ones(256), np.zeros(0), np.zeros(0), 0.05, 5)
    assert r.hist.shape == (0, 6, 256)
    r = fom.fom_run(X, np.ones(256), 4.5, 0.02, 0.05, 0)
    torch.cuda.synchronize()
    assert r.hist.shape == (1, 1, 256) and torch.equal(r.hist[0, 0].cpu(), torch.ones(256, dtype=torch.float64))
    # N beyond the workgroup-per-sample range is refused, not mis-computed
    with pytest.raises(lib.BurgersHipError):
        fom.fom_run(np.linspace(0, 100, 9000), np.ones(9000), 4.5, 0.02, 0.05, 1)
    with pytest.raises(lib.BurgersHipError):
        fom.fd_run(0.0, 100.0, 9000, np.ones(9000), 4.5, 0.02, 0.01, 1)
    Xbad = X.copy(); Xbad[7] = Xbad[9]
    with pytest.raises(ValueError):                     # nodes must be strictly increasing
        fom.fom_run(Xbad, np.ones(256), 4.5, 0.02, 0.05, 1)
    # divergence (reference finding: N=1024 at dt=0.05 hits the cap every step) is flagged, not hidden
    X2, _ = mesh(1024)
    r = fom.fom_run(X2, np.ones(1024), 5.5, 0.03, 0.05, 12)
    torch.cuda.synchronize()
    assert int(r.flags[0]) & lib.BG_FLAG_HIT_CAP
    ho, ito = bc.fom_run(X2, np.ones(1024), 5.5, 0.03, 0.05, 12)
    assert np.array_equal(r.iters.cpu().numpy(), ito)


def test_facade_drop_in(hip):
    """The reference's call pattern (FEM/paper_training_stage.py:33-49) through the drop-in class."""
    from fem_burgers import FEMBurgers
    a, b, m = 0, 100, 255
    X = np.linspace(a, b, m + 1)
    T = np.array([np.arange(1, m + 1), np.arange(2, m + 2)]).T
    u0 = np.ones_like(X)
    fem = FEMBurgers(X, T)
    U = fem.fom_burgers(0.05, 100, u0, 4.75, 0.00, 0.02)
    g = load_golden("fom_n256.npz")
    assert U.shape == (256, 101) and U.dtype == np.float64 and U.flags["C_CONTIGUOUS"]
    assert rel_l2(U, g["U"]) < TOL
    assert np.array_equal(fem.last_iters, g["iters"])
    assert np.array_equal(u0, np.ones_like(X))                     # inputs are never mutated
    Ub = fem.fom_burgers(0.05, 10, u0, np.array([4.75, 5.0]), 0.0, np.array([0.02, 0.025]))
    assert Ub.shape == (2, 256, 11) and rel_l2(Ub[0], g["U"][:, :11]) < TOL


def test_nonuniform_mesh_golden_and_oracle(hip):
    """The reference accepts any mesh; the non-uniform kernels are pinned by a live-reference run
    on a perturbed 96-node mesh with E != 0 and by the oracle on larger perturbed meshes."""
    from burgers_hip import fom
    g = load_golden("fom_general.npz")
    h, it, fl = _run(hip, g["X"], g["u0"], float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]),
                     E=float(g["E"]))
    assert rel_l2(h[0].T, g["U"]) < TOL and np.array_equal(it[0], g["iters"])
    # assembly pieces on the same mesh against the dumped reference matrices
    X, u, mu2 = g["X"], g["u"], float(g["mu2"])
    lo, di, up, rhs = [t.cpu().numpy()[0] for t in fom.fom_assemble(X, u, u, 3.3, mu2, 0.02, E=0.05)]
    A_ref = g["M"] + 0.02 * g["C"] + 0.02 * 0.05 * g["K"]
    A_ref[0, :] = 0; A_ref[0, 0] = 1
    A = br.tridiag_dense(lo, di, up)
    assert np.abs(A - A_ref).max() < 1e-13
    b_ref = g["M"] @ u + 0.02 * g["F"] - 0.02 * g["S"]; b_ref[0] = 3.3
    assert np.abs(rhs - (b_ref - A_ref @ u)).max() < 1e-12
    rng = np.random.default_rng(9)
    for N in (512, 1024, 300):
        X = np.linspace(0, 100, N) + rng.uniform(-0.3, 0.3, N) * (100 / (N - 1))
        X[0], X[-1] = 0.0, 100.0
        B = 5
        mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
        dt = 0.05 * 512 / N
        h, it, fl = _run(hip, X, np.ones(N), mu1, mu2, dt, 15, E=0.002)
        ho, ito = bc.fom_run(X, np.ones(N), mu1, mu2, dt, 15, E=0.002)
        assert rel_l2(h, ho) < TOL and np.array_equal(it, ito), N


def test_fd_newton_stepper(hip):
    """Widening row f.4: FDBurgers.fom_burgers_newton on the same wavefront-tridiagonal skeleton."""
    from burgers_hip import fom
    from fd_burgers import FDBurgers
    g = load_golden("fd_newton.npz")
    for tag in ("n128", "n512"):
        N, dt, nT, mu1, mu2 = g["par_" + tag]
        N, nT = int(N), int(nT)
        res = fom.fd_run(0.0, 100.0, N, np.ones(N), mu1, mu2, dt, nT)
        torch.cuda.synchronize()
        assert rel_l2(res.hist[0].cpu().numpy().T, g["U_" + tag]) < TOL
        assert np.array_equal(res.iters[0].cpu().numpy(), g["iters_" + tag])
    # the reference's committed training snapshot, full 500 steps, through the drop-in class
    fd = FDBurgers(0.0, 100.0, 512)
    U = fd.fom_burgers_newton(0.05, 500, np.ones(512), 4.25, 0.015)
    assert U.shape == (512, 501) and rel_l2(U[:, g["cols"]], g["committed_cols"]) < TOL
    assert rel_l2(U[:, :11], g["committed_first11"]) < TOL
    # batch against the oracle, sizes that need row masks, non-constant initial state
    rng = np.random.default_rng(5)
    for N in (100, 300, 1000, 1024):
        B = 6
        mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
        X = np.linspace(0, 100, N)
        u0 = 1.0 + 0.3 * np.sin(np.outer(rng.uniform(0.5, 2.0, B), X / 100 * np.pi))
        dt = 0.05 * 512 / N
        res = fom.fd_run(0.0, 100.0, N, u0, mu1, mu2, dt, 12)
        torch.cuda.synchronize()
        for b in range(B):
            Uo, ito = br.fd_newton(0.0, 100.0, N, dt, 12, u0[b], mu1[b], mu2[b], return_iters=True)
            assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < TOL, (N, b)
            assert np.array_equal(res.iters[b].cpu().numpy(), ito), (N, b)


@pytest.mark.parametrize("N", [1100, 1536, 2048])
def test_large_meshes_up_to_2048(hip, N):
    """24 / 32 rows per lane (part of the state lives in AGPRs / scratch): FEM and FD steppers vs the oracle."""
    from burgers_hip import fom
    rng = np.random.default_rng(N)
    X, _ = mesh(N)
    B = 4
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt = 0.05 * 512 / N
    h, it, fl = _run(hip, X, np.ones(N), mu1, mu2, dt, 10)
    ho, ito = bc.fom_run(X, np.ones(N), mu1, mu2, dt, 10)
    assert rel_l2(h, ho) < TOL and np.array_equal(it, ito)
    res = fom.fd_run(0.0, 100.0, N, np.ones(N), mu1[:2], mu2[:2], dt, 6)
    torch.cuda.synchronize()
    for b in range(2):
        Uo, ito = br.fd_newton(0.0, 100.0, N, dt, 6, np.ones(N), mu1[b], mu2[b], return_iters=True)
        assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < TOL and np.array_equal(res.iters[b].cpu().numpy(), ito)


@pytest.mark.parametrize("N", [2, 3, 5, 63, 64, 65, 127, 129])
def test_tiny_and_boundary_mesh_sizes(hip, N):
    """Smallest meshes and sizes around the rows-per-lane switch points (1 row per lane and its neighbours)."""
    rng = np.random.default_rng(1000 + N)
    X, _ = mesh(N)
    B = 5
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt = min(0.5, 0.05 * 512 / N)
    h, it, fl = _run(hip, X, np.ones(N), mu1, mu2, dt, 8, E=0.01)
    ho, ito = bc.fom_run(X, np.ones(N), mu1, mu2, dt, 8, E=0.01)
    assert np.isfinite(ho).all()
    assert rel_l2(h, ho) < TOL and np.array_equal(it, ito)


def test_nonfinite_state_exits_like_the_reference(hip):
    """The reference's loop test `error_U > 1e-6` is False for NaN, so a non-finite state leaves the Picard
    loop after one pass per step, silently.  Same exit here (identical iteration counts), plus the
    NONFINITE flag the reference does not have."""
    from burgers_hip import fom, lib
    X, _ = mesh(256)
    u0 = np.ones((2, 256)); u0[1, 40] = np.nan
    r = fom.fom_run(X, u0, [5.0, 5.0], [0.02, 0.02], 0.05, 6)
    torch.cuda.synchronize()
    ho, ito = bc.fom_run(X, u0, [5.0, 5.0], [0.02, 0.02], 0.05, 6)
    it = r.iters.cpu().numpy(); fl = r.flags.cpu().numpy()
    assert np.array_equal(it, ito) and (it[1] == 1).all()
    assert fl[1] & lib.BG_FLAG_NONFINITE and not (fl[0] & lib.BG_FLAG_NONFINITE)
    assert rel_l2(r.hist[0].cpu().numpy(), ho[0]) < TOL          # the healthy neighbour is untouched


def test_randomised_differential_sweep(hip):
    """Seeded random cases over the whole supported size range (every rows-per-lane instantiation,
    uniform and graded meshes, with and without diffusion, non-constant initial states, odd batch
    sizes) against the C oracle: rel-L2 <= 1e-10 and identical iteration counts."""
    rng = np.random.default_rng(4242)
    sizes = [int(n) for n in rng.integers(2, 2049, 22)] + [640, 641, 768, 1025, 1537]
    worst = 0.0
    for case, N in enumerate(sizes):
        B = int(rng.integers(1, 8))
        nsteps = int(rng.integers(3, 9))
        X = np.linspace(0.0, 100.0, N)
        if case % 3 == 1 and N > 3:                      # graded mesh, strictly increasing
            w = rng.uniform(0.6, 1.4, N - 1)
            X = np.concatenate([[0.0], np.cumsum(w)]) * (100.0 / w.sum())
        E = [0.0, 0.003, 0.02][case % 3]
        h = 100.0 / (N - 1)
        dt = float(min(0.4, rng.uniform(0.08, 0.2) * h))  # CFL <= 1.1 at mu1 <= 5.5: the Picard loop converges
        mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
        u0 = 1.0 + 0.1 * np.sin(np.outer(rng.uniform(0.02, 0.1, B), X)) * np.exp(-X / 60.0)
        hh, it, fl = _run(hip, X, u0, mu1, mu2, dt, nsteps, E=E)
        ho, ito = bc.fom_run(X, u0, mu1, mu2, dt, nsteps, E=E)
        assert np.isfinite(ho).all() and ito.max() < 20, f"case {case} N={N} left the convergent regime"
        e = rel_l2(hh, ho)
        worst = max(worst, e)
        assert e < TOL and np.array_equal(it, ito), f"case {case}: N={N} B={B} E={E} dt={dt:.4f} rel={e:.2e}"
        assert (fl == 0).all()
    assert worst < TOL


WIDE_SIZES = [1537, 1800, 2049, 3072, 3073, 4096, 5000, 6144, 8192]


def test_workgroup_wide_tridiag_solve(hip):
    """1536 < N <= 8192: Wang partition per thread + PCR across the four waves of a workgroup."""
    from burgers_hip import fom
    rng = np.random.default_rng(11)
    for N in WIDE_SIZES:
        B = 3
        lo = rng.uniform(-0.2, 0.1, (B, N)); up = rng.uniform(-0.1, 0.3, (B, N))
        di = 0.45 + rng.uniform(0, 0.3, (B, N))
        lo[:, 0] = 0; up[:, -1] = 0
        rhs = rng.standard_normal((B, N))
        sol = fom.tridiag_solve(*[torch.tensor(a, device="cuda") for a in (lo, di, up, rhs)]).cpu().numpy()
        for b in range(B):
            assert rel_l2(sol[b], br.tridiag_solve(lo[b], di[b], up[b], rhs[b])) < 1e-12, f"N={N}"


@pytest.mark.parametrize("N,B", [(1024, 128), (1024, 1), (512, 256), (256, 7), (777, 33), (1300, 5), (65, 3)])
def test_low_latency_form(hip, N, B):
    """The workgroup-per-sample kernels at 2 / 4 / 8 rows per thread (N <= 512 / 1024 / 1536), built as a low-latency
    form for small batches (form="wide"; measured slower than one wavefront per sample, hence opt-in): against the C
    oracle and the wavefront form."""
    from burgers_hip import fom
    rng = np.random.default_rng(N + B)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    X = np.linspace(0.0, 100.0, N)
    dt = 0.05 * 512 / max(N, 512)
    rw = fom.fom_run(X, np.ones(N), mu1, mu2, dt, 12, E=0.001, form="wide")
    rv = fom.fom_run(X, np.ones(N), mu1, mu2, dt, 12, E=0.001, form="wave")
    rd = fom.fom_run(X, np.ones(N), mu1, mu2, dt, 12, E=0.001)
    torch.cuda.synchronize()
    ho, ito = bc.fom_run(X, np.ones(N), mu1, mu2, dt, 12, E=0.001)
    for r in (rw, rv, rd):
        assert rel_l2(r.hist.cpu().numpy(), ho) < TOL and np.array_equal(r.iters.cpu().numpy(), ito)
    assert torch.equal(rd.hist, rv.hist)                     # the default is the wavefront form (bit for bit)


@pytest.mark.parametrize("N", WIDE_SIZES)
def test_workgroup_wide_fom(hip, N):
    """One workgroup per sample: assembly, full runs on uniform and graded meshes, with and without
    diffusion, against the oracle (rel-L2 <= 1e-10, identical iteration counts)."""
    from burgers_hip import fom
    rng = np.random.default_rng(5000 + N)
    B = 3
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt = 0.05 * 512 / N
    for graded in (False, True):
        X = np.linspace(0.0, 100.0, N)
        E = 0.0
        if graded:
            w = rng.uniform(0.7, 1.3, N - 1)
            X = np.concatenate([[0.0], np.cumsum(w)]) * (100.0 / w.sum())
            E = 0.004
        if N in (2049, 5000):                               # one Picard assembly, row by row
            u = 1.0 + 3.0 * rng.random(N)
            lo, di, up, rhs = [t.cpu().numpy()[0] for t in fom.fom_assemble(X, u, u, mu1[0], mu2[0], dt, E=E)]
            M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
            lo_o, di_o, up_o = br.system_tridiag(M3, K3, br.convection_tridiag(X, u), dt, E)
            bb = br.tridiag_matvec(*M3, u) + dt * br.forcing_vector(X, mu2[0]) - dt * br.supg_term(X, u, mu2[0])
            bb[0] = mu1[0]
            r_o = bb - br.tridiag_matvec(lo_o, di_o, up_o, u)
            scale = np.abs(di_o).max()
            assert np.abs(lo - lo_o).max() < 1e-13 * scale and np.abs(di - di_o).max() < 1e-13 * scale
            assert np.abs(up - up_o).max() < 1e-13 * scale
            assert np.abs(rhs - r_o).max() < 1e-12 * max(1.0, np.abs(r_o).max())
        h, it, fl = _run(hip, X, np.ones(N), mu1, mu2, dt, 6, E=E)
        ho, ito = bc.fom_run(X, np.ones(N), mu1, mu2, dt, 6, E=E)
        assert np.isfinite(ho).all() and ito.max() < 20
        assert rel_l2(h, ho) < TOL and np.array_equal(it, ito), f"N={N} graded={graded}"
        assert (fl == 0).all()


def test_full_bench_workload_against_the_oracle(hip):
    """The whole bench.py workload (BASELINE configs[1]: 1024 samples x N=1024 x 500 steps, dt=0.025, the
    bench's seed) against the C oracle on the host cores: every one of the 512 000 Picard iteration counts
    identical, worst per-sample rel-L2 of the full history <= 1e-10."""
    from burgers_hip import fom, lib
    rng = np.random.default_rng(20251121)
    N, B, nsteps, dt = 1024, 1024, 500, 0.025
    X, _ = mesh(N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    res = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nsteps)
    h = lib.to_host(res.hist)
    it = res.iters.cpu().numpy()
    fl = res.flags.cpu().numpy()
    assert (fl & lib.BG_FLAG_NONFINITE == 0).all()
    assert np.array_equal((fl & lib.BG_FLAG_HIT_CAP) != 0, (it >= 20).any(axis=1))   # cap exits are flagged, as in the oracle's counts
    worst = 0.0
    for b0 in range(0, B, 256):                              # oracle in four blocks: bounded host memory
        ho, ito = bc.fom_run(X, np.ones(N), mu1[b0:b0 + 256], mu2[b0:b0 + 256], dt, nsteps)
        assert np.array_equal(it[b0:b0 + 256], ito), f"iteration counts differ in block {b0}"
        d = np.linalg.norm((h[b0:b0 + 256] - ho).reshape(256, -1), axis=1) / np.linalg.norm(ho.reshape(256, -1), axis=1)
        worst = max(worst, float(d.max()))
    assert worst < TOL, worst
    assert int(it.sum()) == 8415010                          # the bench's newton_steps_per_pass


@pytest.mark.parametrize("N", [2049, 3072, 4100, 6144, 8192])
def test_workgroup_wide_fd_stepper(hip, N):
    """FD true-Newton stepper above N = 2048 (one workgroup per sample) against the oracle."""
    from burgers_hip import fom
    rng = np.random.default_rng(7000 + N)
    B = 2
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt = 0.05 * 512 / N
    res = fom.fd_run(0.0, 100.0, N, np.ones(N), mu1, mu2, dt, 5)
    torch.cuda.synchronize()
    for b in range(B):
        Uo, ito = br.fd_newton(0.0, 100.0, N, dt, 5, np.ones(N), mu1[b], mu2[b], return_iters=True)
        assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < TOL and np.array_equal(res.iters[b].cpu().numpy(), ito)


def test_facade_batch_broadcasting(hip):
    """u0, mu1, mu2 broadcast to one batch size; anything else is a ValueError, not a silent broadcast."""
    from fem_burgers import FEMBurgers
    N = 64
    X, T = mesh(N)
    fem = FEMBurgers(X, T)
    u0 = np.ones(N)
    assert fem.fom_burgers(0.05, 3, list(u0), 5, 0.0, np.array(0.02)).shape == (N, 4)
    assert fem.fom_burgers(0.05, 3, u0.astype(np.float32), np.array([4.5, 4.6, 4.7]), 0.0, 0.02).shape == (3, N, 4)
    U = fem.fom_burgers(0.05, 3, np.stack([u0, 1.1 * u0]), 4.5, 0.0, 0.02)        # batch from u0 alone
    assert U.shape == (2, N, 4) and np.allclose(U[1, :, 0], 1.1) and not np.allclose(U[0], U[1])
    with pytest.raises(ValueError):
        fem.fom_burgers(0.05, 3, u0, np.array([4.5, 4.6, 4.7]), 0.0, np.array([0.02, 0.03]))
    with pytest.raises(ValueError):
        fem.fom_burgers(0.05, 3, np.ones(N + 1), 4.5, 0.0, 0.02)
    Phi = np.linalg.qr(np.random.default_rng(0).standard_normal((N, 5)))[0]
    with pytest.raises(ValueError):
        fem.pod_prom_burgers(0.05, 3, u0, np.array([4.5, 4.6]), 0.0, np.array([0.02, 0.03, 0.04]), Phi)
    assert fem.pod_prom_burgers(0.05, 3, u0, 4.5, 0.0, 0.02, np.asfortranarray(Phi)).shape == (N, 4)


def test_out_buffers_are_validated(hip):
    """fom_run(out=...) hands raw pointers to the kernel: a buffer of the wrong shape / dtype / layout is refused."""
    from burgers_hip import fom
    X, _ = mesh(64)
    ok = fom.FomResult(torch.empty((3, 5, 64), dtype=torch.float64, device="cuda"),
                       torch.empty((3, 4), dtype=torch.int32, device="cuda"), torch.empty((3,), dtype=torch.int32, device="cuda"))
    fom.fom_run(X, np.ones(64), [4.5, 5.0, 5.2], 0.02, 0.05, 4, out=ok)
    for bad in (fom.FomResult(ok.hist[:, :4], ok.iters, ok.flags),                                  # too few time levels
                fom.FomResult(ok.hist, ok.iters.to(torch.int64), ok.flags),                         # wrong dtype
                fom.FomResult(ok.hist.transpose(1, 2).contiguous().transpose(1, 2), ok.iters, ok.flags),   # not contiguous
                fom.FomResult(ok.hist.cpu(), ok.iters, ok.flags)):                                  # wrong device
        with pytest.raises(ValueError):
            fom.fom_run(X, np.ones(64), [4.5, 5.0, 5.2], 0.02, 0.05, 4, out=bad)


def test_failed_launch_reports_the_hip_error(hip):
    """Every entry point records the hipError_t of a failed launch (one check_launch for all translation units): a
    launch on a stream handle that is not a stream fails, and bg_last_hip_error() is non-zero afterwards.  Run in a
    child process: what a runtime does with a bad handle is its own business, the parent must not depend on it."""
    import subprocess, sys
    from conftest import PKG
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from burgers_hip import lib\n"
        "L = lib.load()\n"
        "t = torch.zeros((4, 4), dtype=torch.float32, device='cuda')\n"
        "bad = torch.zeros(64, dtype=torch.int64, device='cpu')          # host memory that is no hipStream_t\n"
        "rc = L.bg_mlp_act_jvp(1, 4, 4, lib.ptr(t), None, lib.BG_ACT_RELU, 1.0, ctypes.c_void_p(bad.data_ptr()))\n"
        "print('RC', rc, L.bg_last_hip_error())\n" % PKG)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RC")]
    if out.returncode != 0 or not line:
        pytest.skip("the HIP runtime does not survive an invalid stream handle: " + out.stderr[-200:])
    rc, err = map(int, line[0].split()[1:])
    if rc == hip.BG_OK:
        pytest.skip("the HIP runtime accepted the handle")
    assert rc == hip.BG_ERR_LAUNCH and err != 0


def test_fd_jacobian_option(hip):
    """FDBurgers.fom_burgers_newton(use_fd_jacobian=True) (FD/fd_burgers.py:46-57): dense finite-difference Jacobian,
    library path on the device; against the live-reference fixture and the oracle on a batch."""
    from fd_burgers import FDBurgers
    g = load_golden("fd_newton.npz")
    fd = FDBurgers(0.0, 100.0, 64)
    U = fd.fom_burgers_newton(0.1, 5, np.ones(64), 4.7, 0.02, use_fd_jacobian=True)
    assert U.shape == (64, 6) and rel_l2(U, g["U_fdjac_n64"]) < 1e-9 and np.array_equal(fd.last_iters, g["iters_fdjac_n64"])
    mu1 = np.array([4.3, 5.0, 5.4]); mu2 = np.array([0.016, 0.022, 0.029])
    Ub = fd.fom_burgers_newton(0.1, 4, np.ones(64), mu1, mu2, use_fd_jacobian=True)
    for b in range(3):
        Uo, ito = br.fd_newton(0.0, 100.0, 64, 0.1, 4, np.ones(64), mu1[b], mu2[b], return_iters=True, use_fd_jacobian=True)
        assert rel_l2(Ub[b], Uo) < 1e-9 and np.array_equal(fd.last_iters[b], ito)      # eps = 1e-8 differences: 1e-9, not 1e-10


def test_traced_run_and_reference_console_lines(hip, capsys):
    """bg_fom_run_traced: same history and iteration counts as bg_fom_run, plus error_U of every iteration (:698) equal to
    the oracle's; with ``verbose`` the facade prints the reference's own lines (`Time Step: n. Time: t` :659,
    `Iteration: k, Error: e` :664, the error shown at iteration k being the previous one, 1 before the first)."""
    from burgers_hip import fom
    from fem_burgers import FEMBurgers
    N, nT, dt = 256, 6, 0.05
    X, T = mesh(N)
    mu1, mu2 = np.array([4.4, 5.3]), np.array([0.018, 0.027])
    a = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nT)
    t = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nT, trace=True)
    torch.cuda.synchronize()
    assert torch.equal(a.hist, t.hist) and torch.equal(a.iters, t.iters) and torch.equal(a.flags, t.flags)
    errs = t.errs.cpu().numpy()
    for b in range(2):
        U, it, eo = br.fom_burgers(X, dt, nT, np.ones(N), mu1[b], 0.0, mu2[b], return_iters=True, return_errs=True)
        ran = ~np.isnan(eo)
        assert np.array_equal(np.isnan(errs[b]), ~ran)
        assert np.allclose(errs[b][ran], eo[ran], rtol=1e-6, atol=0.0)          # ratios of norms of converging updates
    fem = FEMBurgers(X, T)
    fem.verbose = True
    capsys.readouterr()
    fem.fom_burgers(dt, nT, np.ones(N), 4.4, 0.0, 0.018)
    lines = capsys.readouterr().out.splitlines()
    it0 = a.iters[0].cpu().numpy()
    assert len(lines) == nT + int(it0.sum())
    assert lines[0] == f"Time Step: 0. Time: {0 * dt}" and lines[1] == "Iteration: 0, Error: 1"
    assert lines[2].startswith("Iteration: 1, Error: ") and abs(float(lines[2].split("Error: ")[1]) - errs[0, 0, 0]) < 1e-12
    assert lines[1 + int(it0[0])] == f"Time Step: 1. Time: {1 * dt}"
