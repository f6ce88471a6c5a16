"""World-size-2 (and 3, uneven) gloo rehearsal of the multi-GPU path on CPU: mu-sharding,
the single all-gather of snapshot blocks, and the offline SVD on the gathered matrix.
The per-rank solver is injected; on CPU the oracle stands in for the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, out_dir):
    import sys
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from burgers_hip import dist as bd, pod
    from oracle import burgers_ref_c as bc
    r, w = bd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(11)
    mu1 = rng.uniform(4.25, 5.5, total); mu2 = rng.uniform(0.015, 0.03, total)
    X = np.linspace(0, 100, 64)

    def runner(m1, m2):
        h, _ = bc.fom_run(X, np.ones(64), m1, m2, 0.2, 6, nthreads=1)
        return torch.from_numpy(h)

    local = bd.sweep(runner, mu1, mu2, rank, world)
    lo, hi = bd.shard_bounds(total, rank, world)
    assert local.shape[0] == hi - lo
    full = bd.all_gather_blocks(local, total)
    assert full.shape == (total, 7, 64)
    S = pod.snapshot_matrix(full)
    U, s, s_all = pod.pod_basis(S, epsilon_squared=1e-6)
    t = bd.max_over_ranks(float(rank), torch.device("cpu"))
    assert t == world - 1
    assert bd.sum_over_ranks(1.0, torch.device("cpu")) == world
    torch.save({"full": full, "U": U, "s": s}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 8), (3, 7)])
def test_sharded_sweep_allgather_svd(tmp_path, world, total):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    from oracle import burgers_ref_c as bc
    from burgers_hip import pod
    rng = np.random.default_rng(11)
    mu1 = rng.uniform(4.25, 5.5, total); mu2 = rng.uniform(0.015, 0.03, total)
    X = np.linspace(0, 100, 64)
    h, _ = bc.fom_run(X, np.ones(64), mu1, mu2, 0.2, 6, nthreads=1)
    outs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    for o in outs:
        assert np.array_equal(o["full"].numpy(), h)          # same full matrix everywhere, sample order kept
        assert torch.equal(o["U"], outs[0]["U"])
    # the gathered snapshot matrix is np.hstack of the per-sample (N, nT+1) blocks (POD/pod.py:80-82)
    S = pod.snapshot_matrix(torch.from_numpy(h)).numpy()
    assert np.array_equal(S, np.hstack([h[b].T for b in range(total)]))
    Un, sn, _ = np.linalg.svd(S, full_matrices=False)
    K = outs[0]["U"].shape[1]
    assert np.allclose(outs[0]["s"].numpy(), sn[:K], rtol=1e-12)
    Ua = pod.align_signs(outs[0]["U"], torch.from_numpy(Un[:, :K].copy())).numpy()
    assert np.abs(Ua - Un[:, :K]).max() < 1e-8


def test_shard_bounds_cover_everything():
    from burgers_hip import dist as bd
    for total in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            b = [bd.shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_pod_truncation_and_file_contracts(tmp_path):
    """Energy rule against the reference's committed singular values; .npy layout contracts."""
    from conftest import load_golden
    from burgers_hip import pod
    g = load_golden("committed_pod_r40.npz")
    assert [pod.n_modes_for_tolerance(g["s_all"], e) for e in g["eps2"]] == list(g["K_expected"])
    snaps = torch.arange(2 * 5 * 3, dtype=torch.float64).reshape(2, 5, 3)      # (B, N, nT+1)
    paths = pod.save_snapshots(tmp_path, snaps, [4.25, 5.5], [0.015, 0.03])
    assert [os.path.basename(p) for p in paths] == ["fem_simulation_mu1_4.250_mu2_0.0150.npy",
                                                    "fem_simulation_mu1_5.500_mu2_0.0300.npy"]
    a = np.load(paths[1])
    assert a.shape == (5, 3) and a.flags["C_CONTIGUOUS"] and a.dtype == np.float64
    pu, ps = pod.save_modes(tmp_path, torch.eye(4, 2, dtype=torch.float64), torch.tensor([2.0, 1.0]), 1e-3)
    assert os.path.basename(pu) == "U_modes_tol_1e-03.npy"
    assert os.path.basename(ps) == "Singular_values_modes_tol_1e-03.npy"
    assert np.load(pu).flags["C_CONTIGUOUS"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/FEM/fem_training_data"),
                    reason="reference checkout not present (build container only)")
def test_pod_basis_reproduces_committed_modes():
    """The 9 committed training snapshots -> the committed 40-mode basis and quadratic Phi."""
    from burgers_hip import pod
    d = "/root/reference/FEM/fem_training_data"
    files = [f for f in os.listdir(d) if f.endswith(".npy") and f.startswith("fem_simulation_")]
    S = torch.from_numpy(np.hstack([np.load(os.path.join(d, f)) for f in files]))
    U, s, _ = pod.pod_basis(S, epsilon_squared=1e-3)
    ref = np.load("/root/reference/POD/modes/U_modes_tol_1e-03.npy")
    assert U.shape == ref.shape == (512, 40)
    Ua = pod.align_signs(U, torch.from_numpy(ref)).numpy()
    assert np.abs(Ua - ref).max() < 1e-10
    Phi, H, q = pod.build_quadratic_manifold(S, 21, alpha=1e-2)
    Phi_ref = np.load("/root/reference/Quadratic_manifold/Phi.npy")
    assert np.abs(pod.align_signs(Phi, torch.from_numpy(Phi_ref)).numpy() - Phi_ref).max() < 1e-9
    H_ref = np.ascontiguousarray(np.load("/root/reference/Quadratic_manifold/H.npy"))
    sg = np.sign((Phi.numpy() * Phi_ref).sum(0))
    I, J = np.triu_indices(21)
    assert np.linalg.norm(H.numpy() * (sg[I] * sg[J]) - H_ref) < 1e-8 * np.linalg.norm(H_ref)
