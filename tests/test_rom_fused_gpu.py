"""GPU parity of bg_rom_run, the device-side POD-PROM time loop (csrc/rom_fused.hip), against the reference's
committed outputs, live reference runs, the oracle, and the host-driven batched path (bg_rom_reduce +
bg_lu_solve_update, pinned by tests/test_rom_gpu.py).  reference: FEM/fem_burgers.py:709-785."""
import numpy as np
import pytest
import torch

from conftest import load_golden, mesh, rel_l2
from oracle import burgers_ref as br

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_fused_golden_and_live(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    live = load_golden("pod_live_r40.npz")
    X, _ = mesh(512)
    for tag, proj in (("galerkin", "Galerkin"), ("lspg", "LSPG")):
        res = rom.pod_prom_run(X, np.ones(512), [4.75, float(live["mu1"])], [0.02, float(live["mu2"])], 0.05, 12,
                               g["Phi"], projection=proj, fused=True)
        torch.cuda.synchronize()
        assert hasattr(res, "info")                                     # really the fused path
        h = res.hist.cpu().numpy(); it = res.iters.cpu().numpy()
        assert rel_l2(h[0].T, g["first13_" + tag]) < TOL                # reference's committed .npy
        nT = int(live["nT"])
        assert rel_l2(h[1].T[:, :nT + 1], live["U_" + proj]) < TOL       # live reference run
        assert np.array_equal(it[1][:nT], live["iters_" + proj])


@pytest.mark.parametrize("N,r,B,nT", [(512, 40, 300, 6), (512, 21, 70, 8), (512, 5, 33, 8), (256, 40, 40, 6), (255, 17, 37, 6),
                                      (100, 8, 29, 5), (301, 24, 300, 4)])
def test_fused_equals_batched_path_and_oracle(hip, N, r, B, nT):
    """Every instantiation (S = 4 / 8, NB = 2 / 6 / 10), even and odd N, more samples than workgroups: identical
    iteration counts and 1e-12 agreement with the host-driven path; oracle on a subset."""
    from burgers_hip import rom
    rng = np.random.default_rng(N * 7 + r)
    X, _ = mesh(N)
    # a smooth orthonormal basis that can carry the solution: snapshots of a short FOM sweep
    from burgers_hip import fom, pod
    m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 2), indexing="ij")
    snap = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 120)
    Phi = pod.pod_basis(pod.snapshot_matrix(snap.hist).contiguous(), n_modes=r)[0].cpu().numpy()
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("Galerkin", "LSPG"):
        f = rom.pod_prom_run(X, np.ones(N), mu1, mu2, 0.05, nT, Phi, projection=proj, E=0.002, fused=True)
        b = rom.pod_prom_run(X, np.ones(N), mu1, mu2, 0.05, nT, Phi, projection=proj, E=0.002, fused=False)
        torch.cuda.synchronize()
        assert hasattr(f, "info") and not hasattr(b, "info")
        assert torch.equal(f.iters, b.iters) and torch.equal(f.flags, b.flags), proj
        # samples that ran into the iteration cap (a crude basis does that) are not contractive: there the two paths'
        # different rounding (Gauss-Jordan vs LU + back substitution) is amplified, everywhere else it stays at 1e-12
        ok = f.flags == 0
        scale = float(b.hist.abs().max())
        assert float((f.hist[ok] - b.hist[ok]).abs().max()) < 1e-12 * scale, proj
        if bool((~ok).any()):
            assert float((f.hist[~ok] - b.hist[~ok]).abs().max()) < 1e-6 * scale, proj
        for s in np.unique(np.linspace(0, B - 1, 3).astype(int)):
            U, ito = br.pod_prom_burgers(X, 0.05, nT, np.ones(N), mu1[s], 0.002, mu2[s], Phi, projection=proj, return_iters=True)
            assert rel_l2(f.hist[s].cpu().numpy().T, U) < TOL, (proj, s)
            assert np.array_equal(f.iters[s].cpu().numpy(), ito), (proj, s)


def test_fused_pivoted_branch(hip):
    """The partial-pivoting branch of the reduced solve: forced on a well-conditioned problem it gives the unpivoted
    branch's result; on a basis whose reduced matrices are NOT diagonally dominant (random columns mixed into the POD
    basis) the multiplier guard trips by itself, and the result still matches the oracle (np.linalg.solve)."""
    from burgers_hip import lib, rom
    g = load_golden("committed_pod_r40.npz")
    X, _ = mesh(512)
    rng = np.random.default_rng(2)
    B, nT = 20, 5
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("Galerkin", "LSPG"):
        pj = rom.PROJ[proj.lower()]
        a = rom.pod_prom_run_fused(X, np.ones(512), mu1, mu2, 0.05, nT, g["Phi"], pj)
        p = rom.pod_prom_run_fused(X, np.ones(512), mu1, mu2, 0.05, nT, g["Phi"], pj, options=lib.BG_OPT_FORCE_PIVOTED)
        torch.cuda.synchronize()
        assert torch.equal(a.iters, p.iters) and float((a.hist - p.hist).abs().max()) < 1e-12 * float(a.hist.abs().max())
    # A basis whose reduced matrix needs real pivoting: Phi2 = Phi T with a dense random T (same span, not orthonormal).
    # The update q = Phi2^T u + dq of :770 is then no projection and the Picard loop diverges, so ONE iteration
    # (max_it = 1) of one time step is compared -- enough to run the guard and the pivot search.
    import scipy.linalg as sl
    T = rng.standard_normal((40, 40))
    Phi2 = g["Phi"] @ T
    M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
    lo, di, up = br.system_tridiag(M3, K3, br.convection_tridiag(X, np.ones(512)), 0.05, 0.0)
    for proj in ("Galerkin", "LSPG"):
        Ar, _ = br._reduce(lo, di, up, np.zeros(512), Phi2, proj.lower())
        assert not np.array_equal(sl.lu_factor(Ar)[1], np.arange(40))            # LAPACK does leave the diagonal here
        res = rom.pod_prom_run(X, np.ones(512), mu1[:6], mu2[:6], 0.05, 1, Phi2, projection=proj, max_it=1, fused=True)
        torch.cuda.synchronize()
        for s_ in range(6):
            U = br.pod_prom_burgers(X, 0.05, 1, np.ones(512), mu1[s_], 0.0, mu2[s_], Phi2, projection=proj, max_it=1)
            scale = np.linalg.cond(Ar) * 1e-13
            assert rel_l2(res.hist[s_].cpu().numpy().T, U) < max(1e-10, scale), (proj, s_, scale)


def test_fused_nonuniform_mesh(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    rng = np.random.default_rng(21)
    N = 512
    X = np.linspace(0, 100, N) + rng.uniform(-0.3, 0.3, N) * (100 / (N - 1))
    X[0], X[-1] = 0.0, 100.0
    mu1 = np.array([4.6, 5.2, 4.9]); mu2 = np.array([0.02, 0.027, 0.016])
    for proj in ("Galerkin", "LSPG"):
        res = rom.pod_prom_run(X, np.ones(N), mu1, mu2, 0.05, 8, g["Phi"], projection=proj, E=0.003, fused=True)
        torch.cuda.synchronize()
        for b in range(3):
            Uo, ito = br.pod_prom_burgers(X, 0.05, 8, np.ones(N), mu1[b], 0.003, mu2[b], g["Phi"], projection=proj,
                                          return_iters=True)
            assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < TOL
            assert np.array_equal(res.iters[b].cpu().numpy(), ito)


def test_fused_singular_system_raises(hip):
    """A basis with a zero column makes Ar exactly singular: np.linalg.solve raises LinAlgError (:767) and so does the
    device loop (info per sample -> SingularReducedSystem)."""
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    X, _ = mesh(512)
    Phi = g["Phi"].copy(); Phi[:, 7] = 0.0
    with pytest.raises(np.linalg.LinAlgError):
        rom.pod_prom_run(X, np.ones(512), [4.75, 5.0], [0.02, 0.02], 0.05, 2, Phi, projection="Galerkin", fused=True)
    with pytest.raises(np.linalg.LinAlgError):
        br.pod_prom_burgers(X, 0.05, 2, np.ones(512), 4.75, 0.0, 0.02, Phi, projection="Galerkin")


def test_fused_edge_cases(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    X, _ = mesh(512)
    r = rom.pod_prom_run(X, np.ones(512), np.zeros(0), np.zeros(0), 0.05, 3, g["Phi"], projection="LSPG")
    assert r.hist.shape == (0, 4, 512) and r.iters.shape == (0, 3)
    r = rom.pod_prom_run(X, np.ones(512), 4.75, 0.02, 0.05, 0, g["Phi"], projection="Galerkin")
    assert r.hist.shape == (1, 1, 512) and torch.equal(r.hist[0, 0].cpu(), torch.ones(512, dtype=torch.float64))
    # iteration cap: max_it = 2 -> HIT_CAP on every sample, two iterations per step
    r = rom.pod_prom_run(X, np.ones(512), [4.5, 5.0], [0.02, 0.03], 0.05, 3, g["Phi"], projection="LSPG", max_it=2)
    assert bool((r.iters == 2).all()) and bool((r.flags & 1).ne(0).all())


@pytest.mark.parametrize("N,r,B", [(2, 1, 3), (3, 2, 1), (17, 3, 5), (64, 40, 2), (512, 1, 4), (130, 9, 270)])
def test_fused_tiny_and_odd_shapes(hip, N, r, B):
    """Degenerate shapes of bg_rom_run: two-node meshes, one mode, a single sample, r = N-limited bases, padded blocks
    (r not a multiple of 4), more samples than workgroups on a small mesh -- against the oracle, on an orthonormal basis."""
    from burgers_hip import rom
    rng = np.random.default_rng(100 * N + r)
    X, _ = mesh(N)
    r = min(r, N)
    Phi = np.linalg.qr(np.concatenate([np.ones((N, 1)), rng.standard_normal((N, r - 1))], axis=1))[0] if r > 1 else \
        np.ones((N, 1)) / np.sqrt(N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("Galerkin", "LSPG"):
        res = rom.pod_prom_run(X, np.ones(N), mu1, mu2, 0.05, 3, Phi, projection=proj, fused=True)
        torch.cuda.synchronize()
        assert hasattr(res, "info")
        for b in np.unique(np.linspace(0, B - 1, 3).astype(int)):
            U, ito = br.pod_prom_burgers(X, 0.05, 3, np.ones(N), mu1[b], 0.0, mu2[b], Phi, projection=proj, return_iters=True)
            fin = np.isfinite(U).all()
            if fin:
                # crude bases do not converge (20-iteration cap): rounding differences are amplified there
                tol = 1e-10 if (ito < 20).all() else 1e-6
                assert rel_l2(res.hist[b].cpu().numpy().T, U) < tol, (proj, b)
                assert np.array_equal(res.iters[b].cpu().numpy(), ito), (proj, b)


def test_fused_full_size_properties(hip):
    """BASELINE configs[2] at full batch (4096 samples, r = 40) through bg_rom_run, size-independent properties: the result
    does not depend on where a sample sits in the batch (permutation, bit-for-bit), a run restarted from a stored column
    continues bit-for-bit (the lifted state and the stored history row are the same doubles), column 0 is u0, nothing is
    flagged.  (The Dirichlet value is NOT a property of the PROM: u = Phi q imposes it only through the projection.)"""
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    rng = np.random.default_rng(20251121)
    X, _ = mesh(512)
    B, nT = 4096, 8
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    perm = rng.permutation(B)
    for proj in ("Galerkin", "LSPG"):
        a = rom.pod_prom_run(X, np.ones(512), mu1, mu2, 0.05, nT, g["Phi"], projection=proj)
        p = rom.pod_prom_run(X, np.ones(512), mu1[perm], mu2[perm], 0.05, nT, g["Phi"], projection=proj)
        permd = torch.as_tensor(perm, device="cuda")
        assert torch.equal(a.hist[permd], p.hist) and torch.equal(a.iters[permd], p.iters) and torch.equal(a.flags[permd], p.flags)
        first = rom.pod_prom_run(X, np.ones(512), mu1, mu2, 0.05, 4, g["Phi"], projection=proj)
        rest = rom.pod_prom_run(X, first.hist[:, 4].contiguous(), mu1, mu2, 0.05, nT - 4, g["Phi"], projection=proj)
        assert torch.equal(first.hist, a.hist[:, :5]) and torch.equal(rest.hist, a.hist[:, 4:])
        assert torch.equal(torch.cat([first.iters, rest.iters], 1), a.iters)
        assert torch.equal(a.hist[:, 0], torch.ones((B, 512), dtype=torch.float64, device="cuda"))
        assert int(a.flags.sum().item()) == 0 and int(a.info.sum().item()) == 0
