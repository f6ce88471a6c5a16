"""GPU parity of bg_rom_run_wide, the device-side POD-PROM time loop for bases of 41 .. 96 modes (csrc/rom_wide.hip): the
basis streams through LDS, the reduced system's accumulators are spread over the four waves of a sample's workgroup.

Pinned against the reference's committed r = 96 PROM outputs (POD/Results_thesis/rom_solutions, basis
POD/modes/U_modes_tol_1e-04.npy; fixture committed_pod_r96.npz), the oracle (oracle/burgers_ref.py) and the library path
(rocBLAS GEMMs + rocSOLVER LU, fused=False), with identical Picard iteration counts.
reference: FEM/fem_burgers.py:709-785 (pod_prom_burgers), POD/Results_thesis/prom_pod.py:35-58.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, mesh, rel_l2
from oracle import burgers_ref as br

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_wide_committed_r96_and_oracle(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r96.npz")
    X, _ = mesh(512)
    assert g["Phi"].shape == (512, 96)
    for tag, proj in (("galerkin", "Galerkin"), ("lspg", "LSPG")):
        res = rom.pod_prom_run(X, np.ones(512), [4.75, 5.3], [0.02, 0.018], 0.05, 8, g["Phi"], projection=proj)
        torch.cuda.synchronize()
        assert hasattr(res, "PhiP") and res.redone == 0            # the device-side loop ran, nothing handed back
        assert rel_l2(res.hist[0].cpu().numpy().T, g["first9_" + tag]) < TOL
        U, ito = br.pod_prom_burgers(X, 0.05, 8, np.ones(512), 5.3, 0.0, 0.018, g["Phi"], projection=proj, return_iters=True)
        assert rel_l2(res.hist[1].cpu().numpy().T, U) < TOL and np.array_equal(res.iters[1].cpu().numpy(), ito)


@pytest.mark.parametrize("N,r,B,nT", [(512, 96, 300, 4), (512, 64, 9, 6), (384, 41, 5, 5), (200, 50, 3, 5), (333, 77, 4, 4)])
def test_wide_vs_library_path_and_oracle(hip, N, r, B, nT):
    """Ragged sizes (N not a multiple of 64, r not a multiple of 4), more samples than workgroups, E != 0."""
    from burgers_hip import rom
    g = load_golden("committed_pod_r96.npz")
    if N == 512:
        Phi = np.ascontiguousarray(g["Phi"][:, :r])
    else:
        xi = np.linspace(0, 1, N)
        cols = [np.ones(N), xi] + [np.tanh((xi - c0) * 14) for c0 in np.linspace(0.05, 0.95, r - 2)]
        Phi = np.linalg.qr(np.stack(cols, 1))[0]
    X, _ = mesh(N)
    rng = np.random.default_rng(N + r)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("Galerkin", "LSPG"):
        f = rom.pod_prom_run(X, np.ones(N), mu1, mu2, 0.05, nT, Phi, projection=proj, E=0.002)
        b = rom.pod_prom_run(X, np.ones(N), mu1[:4], mu2[:4], 0.05, nT, Phi, projection=proj, E=0.002, fused=False)
        torch.cuda.synchronize()
        assert hasattr(f, "PhiP") and not hasattr(b, "PhiP")
        assert torch.equal(f.iters[:4], b.iters) and rel_l2(f.hist[:4].cpu().numpy(), b.hist.cpu().numpy()) < 1e-11, proj
        for s in (0, B - 1):
            U, ito = br.pod_prom_burgers(X, 0.05, nT, np.ones(N), mu1[s], 0.002, mu2[s], Phi, projection=proj, return_iters=True)
            assert rel_l2(f.hist[s].cpu().numpy().T, U) < TOL and np.array_equal(f.iters[s].cpu().numpy(), ito), (proj, s)


def test_wide_results_do_not_depend_on_the_batch_and_pivoting_fallback(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r96.npz")
    X, _ = mesh(512)
    rng = np.random.default_rng(3)
    B = 270                                                  # > 256 workgroups: the persistent loop
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    full = rom.pod_prom_run(X, np.ones(512), mu1, mu2, 0.05, 3, g["Phi"], projection="LSPG")
    one = rom.pod_prom_run(X, np.ones(512), mu1[269], mu2[269], 0.05, 3, g["Phi"], projection="LSPG")
    torch.cuda.synchronize()
    assert torch.equal(full.hist[269], one.hist[0]) and torch.equal(full.iters[269], one.iters[0])
    # Samples handed back by the kernel (info = BG_INFO_NEEDS_PIVOTING: the multiplier guard of its pivot-free elimination) are
    # redone through the library path, whose LU pivots.  With an orthonormal basis -- which the reference's update
    # q = Phi^T u + dq presupposes (:770) -- the guard does not trip (cond(Ar) < 10), so the branch is forced.
    forced = rom.pod_prom_run_wide(X, np.ones(512), mu1[:5], mu2[:5], 0.05, 3, g["Phi"], rom.PROJ["lspg"],
                                   options=hip.BG_OPT_FORCE_PIVOTED)
    torch.cuda.synchronize()
    assert forced.redone == 5 and bool((forced.info == 0).all())
    assert torch.equal(forced.iters, full.iters[:5]) and rel_l2(forced.hist.cpu().numpy(), full.hist[:5].cpu().numpy()) < 1e-11
