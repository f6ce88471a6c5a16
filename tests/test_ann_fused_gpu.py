"""GPU parity of bg_ann_rom_run, the device-side POD-ANN PROM time loop (csrc/rom_ann_fused.hip), against the live
reference fixture, the oracle, and the host-driven batched path (GEMM layers + bg_mlp_act_jvp + bg_rom_reduce +
bg_lu_solve_update, pinned by tests/test_rom_gpu.py).  reference: FEM/fem_burgers.py:1177-1275.

The reference evaluates the closure MLP and its Jacobian in float32, so every comparison here is float32-limited:
5e-6 relative L2 on the history (the bound the oracle itself meets against the live reference run)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, mesh, rel_l2
from oracle import burgers_ref as br

pytestmark = pytest.mark.gpu
TOL32 = 5e-6


def _golden_model(g):
    import torch.nn as nn
    dims = [5, 32, 64, 128, 256, 256, 91]
    layers = []
    for i in range(6):
        lin = nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(g[f"W{i}"])); lin.bias.copy_(torch.from_numpy(g[f"b{i}"]))
        layers.append(lin)
        if i < 5:
            layers.append(nn.ELU())
    return nn.Sequential(*layers).eval()


def _golden_wb(g):
    return [g[f"W{i}"] for i in range(6)], [g[f"b{i}"] for i in range(6)]


def test_ann_fused_live_reference(hip):
    """The fixture recorded from the reference's own pod_ann_prom run (tests/golden/make_golden.py)."""
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    X, _ = mesh(512)
    res = rom.pod_ann_run(X, np.ones(512), float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]),
                          g["U_p"], g["U_s"], _golden_model(g))
    torch.cuda.synchronize()
    assert hasattr(res, "info")                                   # really the device-side loop
    assert rel_l2(res.hist[0].cpu().numpy().T, g["U"]) < TOL32


@pytest.mark.parametrize("proj", ["LSPG", "Galerkin"])
def test_ann_fused_equals_batched_path_and_oracle(hip, proj):
    """More samples than workgroups; iteration counts may differ by one where float32 noise sits on the threshold
    (the two paths sum the layers in different orders)."""
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    model = _golden_model(g)
    X, _ = mesh(512)
    rng = np.random.default_rng(11)
    B, nT = 300, 6
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    f = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, nT, g["U_p"], g["U_s"], model, projection=proj)
    b = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, nT, g["U_p"], g["U_s"], model, projection=proj, fused=False)
    torch.cuda.synchronize()
    assert hasattr(f, "info") and not hasattr(b, "info")
    assert int(f.info.abs().max()) == 0
    fi, bi = f.iters.cpu().numpy(), b.iters.cpu().numpy()
    assert np.abs(fi - bi).max() <= 1 and (fi != bi).mean() < 0.02
    assert np.array_equal(fi >= 50, bi >= 50)                     # same capped steps
    assert torch.equal(f.flags, b.flags)
    fh, bh = f.hist.cpu().numpy(), b.hist.cpu().numpy()
    for s in range(B):
        assert rel_l2(fh[s], bh[s]) < TOL32, s
    Ws, bs = _golden_wb(g)
    for s in (0, 77, 299):
        Uo, ito = br.pod_ann_prom(X, 0.05, nT, np.ones(512), mu1[s], 0.0, mu2[s], g["U_p"], g["U_s"], Ws, bs,
                                  projection=proj, return_iters=True)
        assert rel_l2(fh[s].T, Uo) < TOL32 and np.abs(fi[s] - ito).max() <= 1


def _random_mlp(widths, act, bias, seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    layers = []
    for i in range(len(widths) - 1):
        lin = nn.Linear(widths[i], widths[i + 1], bias=bias)
        with torch.no_grad():
            lin.weight.mul_(0.5)
        layers.append(lin)
        if i < len(widths) - 2:
            layers.append(act())
    return nn.Sequential(*layers).eval()


@pytest.mark.parametrize("N,n,nbar,hidden,act,bias", [(512, 8, 128, [256, 33], "Tanh", True), (256, 3, 20, [7], "ReLU", False),
                                                     (255, 1, 5, [16, 16, 16, 16, 16, 16, 16], "ELU", True),
                                                     (100, 4, 12, [64], "ELU", True), (301, 6, 40, [130, 50], "Tanh", False)])
def test_ann_fused_shapes_and_activations(hip, N, n, nbar, hidden, act, bias):
    """Both row tilings (N <= 256 / <= 512), odd N, every activation, no-bias layers, the limits n = 8, nbar = 128,
    width 256, 8 layers; closure = a random small MLP scaled so that the PROM stays well-posed.  Checked against the
    host-driven path and (one sample) the oracle's pod_ann_prom."""
    import torch.nn as nn
    from burgers_hip import fom, pod, rom
    X, _ = mesh(N)
    m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 2), indexing="ij")
    snap = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 100)
    Phi = pod.pod_basis(pod.snapshot_matrix(snap.hist).contiguous(), n_modes=n + nbar)[0].cpu().numpy()
    U_p, U_s = Phi[:, :n], Phi[:, n:]
    model = _random_mlp([n] + hidden + [nbar], getattr(nn, act), bias, seed=N + n)
    with torch.no_grad():                                    # a gentle closure: O(1e-2) of the primary coordinates
        list(model)[-1].weight.mul_(0.02)
        if bias:
            list(model)[-1].bias.mul_(0.02)
    rng = np.random.default_rng(n)
    B, nT = 19, 5
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("LSPG", "Galerkin"):
        f = rom.pod_ann_run(X, np.ones(N), mu1, mu2, 0.05, nT, U_p, U_s, model, projection=proj, E=0.001)
        b = rom.pod_ann_run(X, np.ones(N), mu1, mu2, 0.05, nT, U_p, U_s, model, projection=proj, E=0.001, fused=False)
        torch.cuda.synchronize()
        assert hasattr(f, "info") and not hasattr(b, "info")
        fi, bi = f.iters.cpu().numpy(), b.iters.cpu().numpy()
        assert np.abs(fi - bi).max() <= 1
        # the two paths must agree on WHICH samples hit the cap / went non-finite (one sample of slack, as for the
        # iteration counts: the float32 closure can put a sample on either side of the threshold), and most samples
        # must be clean so that the trajectories below are really compared
        assert int((f.flags != b.flags).sum().item()) <= 1, (proj, f.flags.tolist(), b.flags.tolist())
        ok = ((f.flags == 0) & (b.flags == 0)).cpu().numpy()
        assert ok.mean() >= 0.8, (proj, f.flags.tolist())
        fh, bh = f.hist.cpu().numpy(), b.hist.cpu().numpy()
        for s in np.flatnonzero(ok):
            assert rel_l2(fh[s], bh[s]) < TOL32, (proj, s)
        s = int(np.flatnonzero(ok)[0])
        lins = [m for m in model if isinstance(m, nn.Linear)]
        Ws = [m.weight.detach().cpu().numpy() for m in lins]
        bs = [np.zeros(m.out_features, np.float32) if m.bias is None else m.bias.detach().cpu().numpy() for m in lins]
        if act == "ELU":                                     # the oracle restates the reference's ELU network only
            Uo, ito = br.pod_ann_prom(X, 0.05, nT, np.ones(N), mu1[s], 0.001, mu2[s], U_p, U_s, Ws, bs, projection=proj,
                                      return_iters=True)
            assert rel_l2(fh[s].T, Uo) < TOL32 and np.abs(fi[s] - ito).max() <= 1


def test_ann_fused_nonuniform_mesh(hip):
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    model = _golden_model(g)
    rng = np.random.default_rng(5)
    X = np.linspace(0.0, 100.0, 512)
    X[1:-1] += rng.uniform(-0.03, 0.03, 510)
    mu1 = np.array([4.4, 5.2]); mu2 = np.array([0.017, 0.026])
    f = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, 5, g["U_p"], g["U_s"], model)
    torch.cuda.synchronize()
    assert hasattr(f, "info")
    Ws, bs = _golden_wb(g)
    for s in range(2):
        Uo = br.pod_ann_prom(X, 0.05, 5, np.ones(512), mu1[s], 0.0, mu2[s], g["U_p"], g["U_s"], Ws, bs)
        assert rel_l2(f.hist[s].cpu().numpy().T, Uo) < TOL32


def test_ann_fused_limits_and_edge_cases(hip):
    """Outside bg_ann_rom_limits the C entry refuses and the Python path falls back to the batched iteration; B = 0
    and nsteps = 0 are no-ops that still fill hist[:, 0]."""
    import ctypes
    import torch.nn as nn
    from burgers_hip import rom
    L = hip.load()
    lim = [ctypes.c_int() for _ in range(4)]
    assert L.bg_ann_rom_limits(*[ctypes.byref(v) for v in lim]) == hip.BG_OK
    assert [v.value for v in lim] == [8, 128, 256, 8]
    g = load_golden("ann_n5.npz")
    model = _golden_model(g)
    X, _ = mesh(512)
    r = rom.pod_ann_run(X, np.ones(512), [4.5], [0.02], 0.05, 0, g["U_p"], g["U_s"], model)
    torch.cuda.synchronize()
    assert hasattr(r, "info") and r.hist.shape == (1, 1, 512) and float((r.hist - 1.0).abs().max()) == 0.0
    r = rom.pod_ann_run(X, np.ones(512), np.zeros(0), np.zeros(0), 0.05, 3, g["U_p"], g["U_s"], model)
    assert r.hist.shape == (0, 4, 512)
    # too wide a layer: the device-side loop declines, the host-driven path takes over
    wide = _random_mlp([5, 300, 91], nn.ELU, True, 0)
    with torch.no_grad():
        list(wide)[-1].weight.mul_(0.01); list(wide)[-1].bias.mul_(0.01)
    assert rom.pod_ann_run_fused(X, np.ones(512), [4.5], [0.02], 0.05, 2, g["U_p"], g["U_s"], wide, rom.PROJ["lspg"]) is None
    r = rom.pod_ann_run(X, np.ones(512), [4.5], [0.02], 0.05, 2, g["U_p"], g["U_s"], wide)
    assert not hasattr(r, "info")
    # a module that is not a plain MLP
    class Odd(nn.Module):
        def __init__(self):
            super().__init__()
            self.l = nn.Linear(5, 91)
        def forward(self, x):
            return 0.01 * torch.sin(self.l(x))
    assert rom.pod_ann_run_fused(X, np.ones(512), [4.5], [0.02], 0.05, 2, g["U_p"], g["U_s"], Odd(), rom.PROJ["lspg"]) is None
    # C-level argument checks
    z = ctypes.c_void_p(0)
    i1 = (ctypes.c_int * 2)(5, 91); vp = (ctypes.c_void_p * 1)(None); a1 = (ctypes.c_int * 1)(0); f1 = (ctypes.c_float * 1)(1.0)
    assert L.bg_ann_rom_run(600, 1, 5, 91, 1, 1, z, z, z, z, z, 1, i1, vp, vp, a1, f1, 0.05, 0.0, 1e-6, 50, 0, z, z, z, z, z, z) == hip.BG_ERR_UNSUPPORTED_N
    assert L.bg_ann_rom_run(512, 1, 9, 91, 1, 1, z, z, z, z, z, 1, i1, vp, vp, a1, f1, 0.05, 0.0, 1e-6, 50, 0, z, z, z, z, z, z) == hip.BG_ERR_UNSUPPORTED_R
    assert L.bg_ann_rom_run(512, 1, 5, 91, 1, 7, z, z, z, z, z, 1, i1, vp, vp, a1, f1, 0.05, 0.0, 1e-6, 50, 0, z, z, z, z, z, z) == hip.BG_ERR_PROJECTION
    assert L.bg_ann_rom_run(512, 1, 5, 91, 1, 1, z, z, z, z, z, 1, i1, vp, vp, a1, f1, 0.05, 0.0, 1e-6, 50, 0, z, z, z, z, z, z) == hip.BG_ERR_BAD_ARG


def test_ann_fused_full_size_cap_pattern(hip):
    """BASELINE config 5's shape at its full batch (B = 2048, the bench draw): every sample finite, the samples that
    run into the 50-iteration cap are the ones the host-driven path (pinned against the oracle by
    test_closure_roms_cap_pattern_matches_the_oracle) reports."""
    import bench
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    model = _golden_model(g)
    X, _ = mesh(512)
    mu1, mu2 = bench.mu_shard(2048, 1, 0)
    f = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, 4, g["U_p"], g["U_s"], model)
    b = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, 4, g["U_p"], g["U_s"], model, fused=False)
    torch.cuda.synchronize()
    assert hasattr(f, "info") and bool(torch.isfinite(f.hist).all())
    # float32 noise on the threshold moves a count by one (by two on a handful of the slowly converging first steps);
    # where that count is the cap itself the flag moves with it
    d = (f.iters - b.iters).abs()
    assert int(d.max()) <= 3 and float((d > 1).float().mean()) < 0.002
    diff = f.flags != b.flags
    assert float(diff.float().mean()) < 0.01
    assert bool((torch.minimum(f.iters, b.iters).max(dim=1).values[diff] >= 49).all())
    err = (f.hist - b.hist).flatten(1).norm(dim=1) / b.hist.flatten(1).norm(dim=1)
    assert float(err.max()) < TOL32


def test_ann_fused_tangent_reuse_is_exact(hip):
    """At a step start the kernel skips the closure evaluation when float32(U_p^T u^n) is bitwise the input of the previous
    step's last evaluation (its tangent is still in LDS).  With BG_OPT_NO_TANGENT_REUSE every step start evaluates: the
    histories, iteration counts and flags must be identical bit for bit."""
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    model = _golden_model(g)
    X, _ = mesh(512)
    rng = np.random.default_rng(3)
    B = 520
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("lspg", "galerkin"):
        a = rom.pod_ann_run_fused(X, np.ones(512), mu1, mu2, 0.05, 12, g["U_p"], g["U_s"], model, rom.PROJ[proj])
        b = rom.pod_ann_run_fused(X, np.ones(512), mu1, mu2, 0.05, 12, g["U_p"], g["U_s"], model, rom.PROJ[proj],
                                  options=hip.BG_OPT_NO_TANGENT_REUSE)
        torch.cuda.synchronize()
        assert torch.equal(a.hist, b.hist) and torch.equal(a.iters, b.iters) and torch.equal(a.flags, b.flags), proj
