"""GPU parity of the projection-ROM path (bg_rom_reduce, bg_lu_solve and the three batched
time-steppers) against the oracle and the golden fixtures.  fp64 parts: rel-L2 <= 1e-10;
POD-ANN is fp32-limited by the reference's own float32 MLP evaluation."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden, mesh, rel_l2
from oracle import burgers_ref as br

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


@pytest.mark.parametrize("n", [1, 5, 8, 21, 40, 47, 64])
def test_lu_solve_vs_numpy(hip, n):
    from burgers_hip import rom
    rng = np.random.default_rng(n)
    B = 37
    A = rng.standard_normal((B, n, n)) + 0.1 * np.eye(n)        # needs pivoting
    A[3] = A[3][::-1].copy()                                     # tiny leading pivots
    b = rng.standard_normal((B, n))
    x, info = rom.lu_solve(_dev(A), _dev(b), -1.0)
    torch.cuda.synchronize()
    ref = np.linalg.solve(A, -b[..., None])[..., 0]
    assert (info.cpu().numpy() == 0).all()
    for i in range(B):
        cond = np.linalg.cond(A[i])
        assert rel_l2(x[i].cpu().numpy(), ref[i]) < 1e-13 * max(10.0, cond), (n, i)
    # zero matrix -> singular, reported (numpy raises LinAlgError there)
    A0 = np.zeros((2, n, n)); A0[1] = np.eye(n)
    _, info = rom.lu_solve(_dev(A0), _dev(np.ones((2, n))), 1.0)
    assert info.cpu().numpy()[0] != 0 and info.cpu().numpy()[1] == 0


@pytest.mark.parametrize("N,r,shared", [(512, 40, True), (512, 21, False), (512, 5, False), (256, 16, True),
                                        (300, 33, True), (96, 7, False), (512, 47, True), (128, 40, False)])
def test_rom_reduce_vs_oracle(hip, N, r, shared):
    from burgers_hip import rom
    rng = np.random.default_rng(N * 100 + r)
    X, _ = mesh(N)
    B = 7
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt, E = 0.05, 0.01
    U = 1.0 + 4.0 * rng.random((B, N)); Un = 1.0 + 4.0 * rng.random((B, N))
    W = rng.standard_normal((N, r)) if shared else rng.standard_normal((B, N, r))
    c = rom._setup(X, Un, mu1, mu2, dt, E, None)
    G = torch.empty((B, N), dtype=torch.float64, device="cuda")
    rom._mass_rhs(c, _dev(Un), G)
    M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
    for supg in (True, False):
        for pname, proj in (("galerkin", 0), ("lspg", 1)):
            Ar = torch.zeros((B, r, r), dtype=torch.float64, device="cuda")
            brr = torch.zeros((B, r), dtype=torch.float64, device="cuda")
            wtu = torch.zeros((B, r), dtype=torch.float64, device="cuda")
            active = torch.ones(B, dtype=torch.int32, device="cuda"); active[2] = 0
            rom.rom_reduce(c, _dev(W), _dev(U), G, proj, supg, active, Ar, brr, wtu)
            torch.cuda.synchronize()
            Ar, brr, wtu = Ar.cpu().numpy(), brr.cpu().numpy(), wtu.cpu().numpy()
            for b in range(B):
                if b == 2:
                    assert not Ar[b].any() and not brr[b].any()        # skipped sample untouched
                    continue
                Wb = W if shared else W[b]
                lo, di, up = br.system_tridiag(M3, K3, br.convection_tridiag(X, U[b]), dt, E)
                bb = br.tridiag_matvec(*M3, Un[b]) + dt * br.forcing_vector(X, mu2[b])
                if supg:
                    bb = bb - dt * br.supg_term(X, U[b], mu2[b])
                bb[0] = mu1[b]
                R = br.tridiag_matvec(lo, di, up, U[b]) - bb
                Ar_ref, br_ref = br._reduce(lo, di, up, R, Wb, pname)
                assert rel_l2(Ar[b], Ar_ref) < 1e-13, (pname, supg, b)
                assert rel_l2(brr[b], br_ref) < 1e-12, (pname, supg, b)
                assert rel_l2(wtu[b], Wb.T @ U[b]) < 1e-13


def test_pod_prom_golden_and_live(hip):
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    live = load_golden("pod_live_r40.npz")
    X, _ = mesh(512)
    for tag, proj in (("galerkin", "Galerkin"), ("lspg", "LSPG")):
        res = rom.pod_prom_run(X, np.ones(512), [4.75, float(live["mu1"])], [0.02, float(live["mu2"])], 0.05, 12,
                               g["Phi"], projection=proj, fused=False)  # host-driven path (the fused one: test_rom_fused_gpu.py)
        torch.cuda.synchronize()
        h = res.hist.cpu().numpy(); it = res.iters.cpu().numpy()
        assert rel_l2(h[0].T, g["first13_" + tag]) < TOL                # reference's committed .npy
        nT = int(live["nT"])
        assert rel_l2(h[1].T[:, :nT + 1], live["U_" + proj]) < TOL       # live reference run
        assert np.array_equal(it[1][:nT], live["iters_" + proj])
    with pytest.raises(ValueError):
        rom.pod_prom_run(X, np.ones(512), 4.75, 0.02, 0.05, 1, g["Phi"], projection="lspg")


@pytest.mark.parametrize("fused", [False, True])
def test_pod_prom_batch_vs_oracle(hip, fused):
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    rng = np.random.default_rng(3)
    X, _ = mesh(512)
    B = 12
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    for proj in ("Galerkin", "LSPG"):
        res = rom.pod_prom_run(X, np.ones(512), mu1, mu2, 0.05, 25, g["Phi"], projection=proj, fused=fused)
        torch.cuda.synchronize()
        h = res.hist.cpu().numpy(); it = res.iters.cpu().numpy()
        for b in range(B):
            U, ito = br.pod_prom_burgers(X, 0.05, 25, np.ones(512), mu1[b], 0.0, mu2[b], g["Phi"], projection=proj,
                                         return_iters=True)
            assert rel_l2(h[b].T, U) < TOL, (proj, b)
            assert np.array_equal(it[b], ito), (proj, b)


def test_quadratic_golden_and_live(hip):
    from burgers_hip import rom
    c = load_golden("committed_quadratic_n21.npz")
    live = load_golden("quadratic_live_n21.npz")
    X, _ = mesh(512)
    res = rom.quadratic_run(X, np.ones(512), float(c["mu1"]), float(c["mu2"]), 0.05, 6, c["Phi"], c["H"])
    torch.cuda.synchronize()
    assert rel_l2(res.hist[0].cpu().numpy().T, c["first7"]) < TOL       # committed quadratic PROM output
    for proj in ("Galerkin", "LSPG"):
        res = rom.quadratic_run(X, np.ones(512), float(live["mu1"]), float(live["mu2"]), float(live["At"]),
                                int(live["nT"]), c["Phi"], c["H"], projection=proj)
        torch.cuda.synchronize()
        assert rel_l2(res.hist[0].cpu().numpy().T, live["U_" + proj]) < TOL
        assert np.array_equal(res.iters[0].cpu().numpy(), live["iters_" + proj])
    with pytest.raises(ValueError):
        rom.quadratic_run(X, np.ones(512), 4.5, 0.02, 0.05, 1, c["Phi"], c["H"], projection="petrov")


def test_quadratic_batch_vs_oracle(hip):
    from burgers_hip import rom
    c = load_golden("committed_quadratic_n21.npz")
    rng = np.random.default_rng(5)
    X, _ = mesh(512)
    B = 6
    mu1 = rng.uniform(4.4, 5.3, B); mu2 = rng.uniform(0.017, 0.028, B)
    res = rom.quadratic_run(X, np.ones(512), mu1, mu2, 0.05, 10, c["Phi"], c["H"])
    torch.cuda.synchronize()
    for b in range(B):
        U, ito = br.pod_quadratic_manifold(X, 0.05, 10, np.ones(512), mu1[b], 0.0, mu2[b], c["Phi"], c["H"],
                                           return_iters=True)
        assert rel_l2(res.hist[b].cpu().numpy().T, U) < TOL
        assert np.array_equal(res.iters[b].cpu().numpy(), ito)


def _ann_model(g):
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


def test_pod_ann_live_reference(hip):
    """The reference evaluates the MLP and its autograd Jacobian in float32, so parity is
    fp32-limited: tolerance 5e-6 (same bound the oracle meets against the live reference)."""
    from burgers_hip import rom
    g = load_golden("ann_n5.npz")
    model = _ann_model(g)
    X, _ = mesh(512)
    # forward / Jacobian vectors recorded from the reference's torch model
    q = _dev(g["qp"]).float()
    with torch.no_grad():
        fwd = model.cuda()(q).cpu().numpy()
        jac = rom.ann_jacobian(model, q).cpu().numpy()
    assert np.abs(fwd - g["fwd"]).max() < 2e-5 * max(1.0, np.abs(g["fwd"]).max())
    assert np.abs(jac - g["jac"]).max() < 2e-4 * max(1.0, np.abs(g["jac"]).max())
    # the fused evaluation (one GEMM per layer over the 1 + n rows, bg_mlp_act_jvp per activation)
    ev = rom.AnnEvaluator(model.cuda(), q.shape[1], torch.float32)
    jt = torch.zeros((q.shape[0], q.shape[1], g["fwd"].shape[1]), dtype=torch.float64, device="cuda")
    qs = torch.zeros((q.shape[0], g["fwd"].shape[1]), dtype=torch.float64, device="cuda")
    ev.bind(q.shape[0], q.shape[1], torch.device("cuda", 0), jt, qs)
    assert ev.fused
    ev.eval(q.double())
    torch.cuda.synchronize()
    assert np.abs(qs.cpu().numpy() - g["fwd"]).max() < 2e-5 * max(1.0, np.abs(g["fwd"]).max())
    assert np.abs(jt.transpose(1, 2).cpu().numpy() - g["jac"]).max() < 2e-4 * max(1.0, np.abs(g["jac"]).max())
    res = rom.pod_ann_run(X, np.ones(512), float(g["mu1"]), float(g["mu2"]), float(g["At"]), int(g["nT"]),
                          g["U_p"], g["U_s"], model)
    torch.cuda.synchronize()
    assert rel_l2(res.hist[0].cpu().numpy().T, g["U"]) < 5e-6
    Ws = [g[f"W{i}"] for i in range(6)]; bs = [g[f"b{i}"] for i in range(6)]
    Uo = br.pod_ann_prom(X, float(g["At"]), int(g["nT"]), np.ones(512), float(g["mu1"]), 0.0, float(g["mu2"]),
                         g["U_p"], g["U_s"], Ws, bs)
    assert rel_l2(res.hist[0].cpu().numpy().T, Uo) < 5e-6


def test_facade_rom_methods(hip):
    """Reference call patterns: POD/Results_thesis/prom_pod.py:58, quadratic_prom_simulation.py:49-55,
    POD-ANN/pod_ann_prom_burgers.py:81."""
    from fem_burgers import FEMBurgers
    X, T = mesh(512)
    fem = FEMBurgers(X, T)
    g = load_golden("committed_pod_r40.npz")
    U = fem.pod_prom_burgers(0.05, 12, np.ones(512), 4.75, 0.0, 0.02, g["Phi"], projection="LSPG")
    assert U.shape == (512, 13) and rel_l2(U, g["first13_lspg"]) < TOL
    with pytest.raises(ValueError):
        fem.pod_prom_burgers(0.05, 1, np.ones(512), 4.75, 0.0, 0.02, g["Phi"], projection="Petrov")
    c = load_golden("committed_quadratic_n21.npz")
    Hf = np.asfortranarray(c["H"])                       # the committed H.npy is Fortran-ordered
    U = fem.pod_quadratic_manifold(0.05, 6, np.ones(512), float(c["mu1"]), 0.0, float(c["mu2"]), c["Phi"], Hf,
                                   projection="LSPG")
    assert rel_l2(U, c["first7"]) < TOL
    a = load_golden("ann_n5.npz")
    U = fem.pod_ann_prom(0.05, 4, np.ones(512), 4.56, 0.0, 0.019, a["U_p"], a["U_s"], _ann_model(a))
    assert U.shape == (512, 5) and rel_l2(U, a["U"]) < 5e-6


def test_rom_nonuniform_mesh_vs_oracle(hip):
    """ROM path on a perturbed mesh (per-element lengths in the fused assembly): the reduce
    kernel against the oracle's projection, then full POD runs."""
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    rng = np.random.default_rng(21)
    N = 512
    X = np.linspace(0, 100, N) + rng.uniform(-0.3, 0.3, N) * (100 / (N - 1))
    X[0], X[-1] = 0.0, 100.0
    Phi = g["Phi"]
    B, dt, E = 3, 0.05, 0.003
    mu1 = np.array([4.6, 5.2, 4.9]); mu2 = np.array([0.02, 0.027, 0.016])
    U = 1.0 + 4.0 * rng.random((B, N)); Un = 1.0 + 4.0 * rng.random((B, N))
    c = rom._setup(X, Un, mu1, mu2, dt, E, None)
    assert c.mesh_opt == 2
    G = torch.empty((B, N), dtype=torch.float64, device="cuda")
    rom._mass_rhs(c, _dev(Un), G)
    M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
    for pname, proj in (("galerkin", 0), ("lspg", 1)):
        Ar = torch.zeros((B, 40, 40), dtype=torch.float64, device="cuda")
        brr = torch.zeros((B, 40), dtype=torch.float64, device="cuda")
        rom.rom_reduce(c, _dev(Phi), _dev(U), G, proj, True, None, Ar, brr, None)
        torch.cuda.synchronize()
        for b in range(B):
            lo, di, up = br.system_tridiag(M3, K3, br.convection_tridiag(X, U[b]), dt, E)
            bb = br.tridiag_matvec(*M3, Un[b]) + dt * br.forcing_vector(X, mu2[b]) - dt * br.supg_term(X, U[b], mu2[b])
            bb[0] = mu1[b]
            R = br.tridiag_matvec(lo, di, up, U[b]) - bb
            Ar_ref, br_ref = br._reduce(lo, di, up, R, Phi, pname)
            assert rel_l2(Ar[b].cpu().numpy(), Ar_ref) < 1e-13
            assert rel_l2(brr[b].cpu().numpy(), br_ref) < 1e-12
    for proj in ("Galerkin", "LSPG"):
        res = rom.pod_prom_run(X, np.ones(N), mu1, mu2, dt, 8, Phi, projection=proj, E=E)
        torch.cuda.synchronize()
        for b in range(B):
            Uo, ito = br.pod_prom_burgers(X, dt, 8, np.ones(N), mu1[b], E, mu2[b], Phi, projection=proj, return_iters=True)
            assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < TOL
            assert np.array_equal(res.iters[b].cpu().numpy(), ito)


def test_nonintrusive_decoder_fp32_and_bf16(hip):
    """Config-5 decoder-only variant: fp32 matches the reference's own prediction; the bf16 tier
    is reported against it (bf16 cannot meet an fp64 tolerance)."""
    import torch.nn as nn
    from burgers_hip import decoder
    g = load_golden("nonintrusive_decoder.npz")
    dims = [3, 32, 64, 128, 160]
    layers = []
    for i, key in enumerate((0, 2, 4, 6)):
        lin = nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(g[f"{key}_weight"])); lin.bias.copy_(torch.from_numpy(g[f"{key}_bias"]))
        layers.append(lin)
        if i < 3:
            layers.append(nn.ELU())
    model = nn.Sequential(*layers).eval()
    import copy
    U = decoder.predict_on_grid([float(g["mu1"]), 5.0], [float(g["mu2"]), 0.025], int(g["Nt"]), g["U_modes"],
                                copy.deepcopy(model), g["mean"], g["std"])
    assert U.shape == (2, 512, 501)
    assert rel_l2(U[0].cpu().numpy()[:, g["cols"]], g["Uhat_cols"]) < 1e-5
    Ub = decoder.predict_on_grid(float(g["mu1"]), float(g["mu2"]), int(g["Nt"]), g["U_modes"], copy.deepcopy(model),
                                 g["mean"], g["std"], dtype=torch.bfloat16)
    err = rel_l2(Ub[0].cpu().numpy()[:, g["cols"]], g["Uhat_cols"])
    # bf16 tier of config 5: MEASURED 1.73e-2 on these nine columns (they include the steep early front), 6.0e-3 .. 6.8e-3
    # over whole trajectories on the bench's mu range (tests/fuzz/measure_config5.py): 8-bit significands through a
    # 4-layer MLP and a 160-term contraction.  Gated at twice the measured value.
    assert 1e-3 < err < 3.5e-2, err


def _decoder_model(g, dims=(3, 32, 64, 128, 160)):
    import torch.nn as nn
    layers = []
    for i, key in enumerate((0, 2, 4, 6)):
        lin = nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(g[f"{key}_weight"])); lin.bias.copy_(torch.from_numpy(g[f"{key}_bias"]))
        layers.append(lin)
        if i < 3:
            layers.append(nn.ELU())
    return nn.Sequential(*layers).eval()


def test_decoder_mlp_in_kernel_matches_the_pytorch_bf16_module(hip):
    """bg_decode_mlp_bf16 (MLP + contraction in one kernel) against the PyTorch bf16 module followed by
    bg_decode_modes_bf16: same rounding points (Linear output -> bf16, ELU in float32 -> bf16), so what differs is the
    summation order inside the float32 accumulators -- a coefficient that lands on the other side of a bf16 rounding
    boundary moves by 2^-8 relative.  Measured 1.5e-3 relative L2 between the two; both sit at the same distance from the
    float32 reference (the bf16 tier's own error, 6e-3 .. 1.7e-2)."""
    import copy
    from burgers_hip import decoder
    g = load_golden("nonintrusive_decoder.npz")
    model = _decoder_model(g)
    rng = np.random.default_rng(5)
    mu1, mu2 = rng.uniform(4.25, 5.5, 37), rng.uniform(0.015, 0.03, 37)          # 37 x 501 columns: a ragged last workgroup
    mu1[0], mu2[0] = float(g["mu1"]), float(g["mu2"])
    Nt = int(g["Nt"])
    dec = decoder.GridDecoder(Nt, g["U_modes"], copy.deepcopy(model), g["mean"], g["std"], dtype=torch.bfloat16)
    assert dec.plan is not None, "the committed decoder is a plain MLP: it must take bg_decode_mlp_bf16"
    ref = decoder.GridDecoder(Nt, g["U_modes"], copy.deepcopy(model), g["mean"], g["std"], dtype=torch.bfloat16, fused=False)
    assert ref.plan is None
    U, V = dec.predict(mu1, mu2), ref.predict(mu1, mu2)
    assert U.shape == V.shape == (37, 512, Nt) and torch.isfinite(U).all()
    assert rel_l2(U.cpu().numpy(), V.cpu().numpy()) < 5e-3
    F = decoder.GridDecoder(Nt, g["U_modes"], copy.deepcopy(model), g["mean"], g["std"]).predict(mu1, mu2)   # float32 tier
    eu, ev = rel_l2(U.cpu().numpy(), F.cpu().numpy()), rel_l2(V.cpu().numpy(), F.cpu().numpy())
    assert eu < 2e-2 and ev < 2e-2 and eu < 1.5 * ev + 1e-3, (eu, ev)
    # the committed reference prediction (nine columns of the training sample, float64 modes @ float32 MLP)
    assert rel_l2(U[0].cpu().numpy()[:, g["cols"]], g["Uhat_cols"]) < 3.5e-2
    # chunking the batch does not change a bit: a column depends on its own (mu1, mu2, t) only
    assert torch.equal(dec.predict(mu1[5:19], mu2[5:19]), U[5:19])
    # a model the in-kernel form does not cover falls back to the PyTorch module (odd width)
    odd = torch.nn.Sequential(torch.nn.Linear(3, 20), torch.nn.ELU(), torch.nn.Linear(20, 160)).eval()
    d2 = decoder.GridDecoder(Nt, g["U_modes"], odd, g["mean"], g["std"], dtype=torch.bfloat16)
    assert d2.plan is not None        # widths are padded to 32 on the host: 20 -> 32 with zero rows
    W = d2.predict(mu1[:3], mu2[:3])
    W0 = decoder.GridDecoder(Nt, g["U_modes"], odd, g["mean"], g["std"], dtype=torch.bfloat16, fused=False).predict(mu1[:3], mu2[:3])
    assert rel_l2(W.cpu().numpy(), W0.cpu().numpy()) < 1e-2
    class Gated(torch.nn.Module):
        def __init__(self):
            super().__init__(); self.a = torch.nn.Linear(3, 160)
        def forward(self, x):
            return self.a(x) * torch.sigmoid(x[:, :1])
    assert decoder.GridDecoder(Nt, g["U_modes"], Gated().eval(), g["mean"], g["std"], dtype=torch.bfloat16).plan is None


@pytest.mark.parametrize("acts", [("relu", "tanh", "none"), ("elu", "none")])
def test_decode_mlp_c_abi_random_models_and_refusals(hip, acts):
    """bg_decode_mlp_bf16 through the C ABI on random networks (every activation kind, a 256-wide layer = the sixteen-
    fragment instantiation, a ragged last workgroup) against a PyTorch emulation with the same rounding points; refusals."""
    import ctypes
    from burgers_hip import lib as L_
    L = L_.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(len(acts))
    N, n, B, Nt = 64, 64, 3, 77
    widths = [16, 256, 96, n] if len(acts) == 3 else [16, 32, n]            # padded input width first
    kinds = {"none": hip.BG_ACT_NONE, "elu": hip.BG_ACT_ELU, "relu": hip.BG_ACT_RELU, "tanh": hip.BG_ACT_TANH}
    bf = lambda t: t.to(torch.bfloat16)
    Ws = [bf(torch.randn((widths[i + 1], widths[i]), device="cuda", generator=g) / widths[i] ** 0.5) for i in range(len(acts))]
    Ws[0][:, 3:] = 0                                                        # three real inputs
    bs = [bf(0.1 * torch.randn((widths[i + 1],), device="cuda", generator=g)) for i in range(len(acts))]
    Um = bf(torch.randn((N, n), device="cuda", generator=g))
    z1 = torch.randn((B,), dtype=torch.float64, device="cuda", generator=g)
    z2 = torch.randn((B,), dtype=torch.float64, device="cuda", generator=g)
    zt = torch.linspace(-1.7, 1.7, Nt, dtype=torch.float64, device="cuda")
    out = torch.full((B, N, Nt), -7.0, dtype=torch.float64, device="cuda")
    nl = len(acts)
    arr = lambda ty, v: (ty * nl)(*v)
    args = (arr(ctypes.c_int, widths[:-1]), arr(ctypes.c_int, widths[1:]), arr(ctypes.c_void_p, [w.data_ptr() for w in Ws]),
            arr(ctypes.c_void_p, [b.data_ptr() for b in bs]), arr(ctypes.c_int, [kinds[a] for a in acts]), arr(ctypes.c_float, [1.0] * nl))
    L_.check(L.bg_decode_mlp_bf16(N, n, B, Nt, L_.ptr(Um), L_.ptr(z1), L_.ptr(z2), L_.ptr(zt), nl, *args, L_.ptr(out),
                                  L_.stream_ptr(dev)), "bg_decode_mlp_bf16")
    torch.cuda.synchronize()
    # the same network with the same rounding points: bf16 operands, float32 accumulate, Linear output -> bf16, activation in float32 -> bf16
    x = torch.zeros((B * Nt, 16), dtype=torch.float32, device="cuda")
    x[:, 0] = bf(z1.float())[:, None].expand(B, Nt).reshape(-1).float()
    x[:, 1] = bf(z2.float())[:, None].expand(B, Nt).reshape(-1).float()
    x[:, 2] = bf(zt.float())[None, :].expand(B, Nt).reshape(-1).float()
    fn = {"none": lambda v: v, "elu": torch.nn.functional.elu, "relu": torch.relu, "tanh": torch.tanh}
    for W, b, a in zip(Ws, bs, acts):
        x = bf(fn[a](bf(x @ W.float().t() + b.float()).float())).float()
    ref = torch.matmul(Um.double(), x.double().reshape(B, Nt, n).transpose(1, 2))
    # a float32 summation-order difference may move a bf16 rounding by one unit in the last place (2^-8 relative) somewhere
    assert float((out - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    assert float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)) < 2e-3
    z = ctypes.c_void_p(0)
    one = lambda ty, v: (ty * 1)(v)
    vp = one(ctypes.c_void_p, Ws[0].data_ptr())
    call = lambda N_, n_, B_, win, wout: L.bg_decode_mlp_bf16(N_, n_, B_, 1, z, z, z, z, 1, one(ctypes.c_int, win), one(ctypes.c_int, wout),
                                                            vp, vp, one(ctypes.c_int, 0), one(ctypes.c_float, 1.0), z, z)
    assert call(33, 32, 1, 16, 32) == hip.BG_ERR_UNSUPPORTED_N
    assert call(32, 48, 1, 16, 48) == hip.BG_ERR_UNSUPPORTED_R             # n not a multiple of 32
    assert call(32, 32, 1, 16, 512) == hip.BG_ERR_UNSUPPORTED_R            # wider than 256
    assert call(32, 32, 1, 32, 32) == hip.BG_ERR_BAD_ARG                   # the first layer reads the 16 padded inputs
    assert call(32, 32, 1, 16, 64) == hip.BG_ERR_BAD_ARG                   # the last layer must produce n coefficients
    assert call(32, 32, 1, 16, 32) == hip.BG_ERR_BAD_ARG                   # null operands
    assert call(32, 32, 0, 16, 32) == hip.BG_OK                            # empty batch


def test_pod_rbf_prom_live_reference(hip):
    """Widening row f.3: pod_rbf_prom with a 300-centre closure; both kernels, both projections.
    The closure weights reach 3.6e2, which amplifies rounding in the decoder: tolerance 1e-9."""
    from burgers_hip import rom
    from fem_burgers import FEMBurgers
    g = load_golden("rbf_n17.npz")
    X, T = mesh(512)
    for kernel, proj in (("gaussian", "LSPG"), ("imq", "Galerkin")):
        args = (g["U_p"], g["U_s"], g["X_train"], g["W_" + kernel], float(g["eps_" + kernel]), g["x_min"], g["x_max"],
                g["y_min"], g["y_max"])
        res = rom.pod_rbf_run(X, np.ones(512), [float(g["mu1"]), 5.1], [float(g["mu2"]), 0.024], float(g["At"]),
                              int(g["nT"]), *args, projection=proj, kernel=kernel, max_newton=20)
        torch.cuda.synchronize()
        assert rel_l2(res.hist[0].cpu().numpy().T, g["U_" + kernel]) < 1e-9
        assert np.array_equal(res.iters[0].cpu().numpy(), g["iters_" + kernel])
        Uo, ito = br.pod_rbf_prom(X, float(g["At"]), int(g["nT"]), np.ones(512), 5.1, 0.0, 0.024, *args,
                                  projection=proj, kernel=kernel, max_newton=20, return_iters=True)
        assert rel_l2(res.hist[1].cpu().numpy().T, Uo) < 1e-9 and np.array_equal(res.iters[1].cpu().numpy(), ito)
    fem = FEMBurgers(X, T)
    U = fem.pod_rbf_prom(0.05, 4, np.ones(512), 4.75, 0.0, 0.02, g["U_p"], g["U_s"], g["X_train"], g["W_gaussian"],
                         float(g["eps_gaussian"]), g["x_min"], g["x_max"], g["y_min"], g["y_max"], projection="LSPG",
                         kernel="gaussian", tol_newton=1e-6, max_newton=20)
    assert U.shape == (512, 5) and rel_l2(U, g["U_gaussian"]) < 1e-9
    with pytest.raises(ValueError):
        fem.pod_rbf_prom(0.05, 1, np.ones(512), 4.75, 0.0, 0.02, g["U_p"], g["U_s"], g["X_train"], g["W_imq"], 1.0,
                         g["x_min"], g["x_max"], g["y_min"], g["y_max"], kernel="multiquadric")


def test_pod_prom_large_basis_library_path(hip):
    """r = 96 (the thesis' tol 1e-04 basis) is beyond the register-resident kernels: HIP assembly +
    library GEMM / LU path, against the reference's committed PROM outputs and the oracle."""
    from burgers_hip import rom
    g = load_golden("committed_pod_r96.npz")
    X, _ = mesh(512)
    assert g["Phi"].shape == (512, 96)
    for tag, proj in (("galerkin", "Galerkin"), ("lspg", "LSPG")):
        res = rom.pod_prom_run(X, np.ones(512), [4.75, 5.3], [0.02, 0.018], 0.05, 8, g["Phi"], projection=proj, fused=False)
        torch.cuda.synchronize()
        assert not hasattr(res, "PhiP")                       # the library path (r = 96 defaults to bg_rom_run_wide since round 3)
        assert rel_l2(res.hist[0].cpu().numpy().T, g["first9_" + tag]) < TOL
        U, ito = br.pod_prom_burgers(X, 0.05, 8, np.ones(512), 5.3, 0.0, 0.018, g["Phi"], projection=proj, return_iters=True)
        assert rel_l2(res.hist[1].cpu().numpy().T, U) < TOL and np.array_equal(res.iters[1].cpu().numpy(), ito)


def test_local_prom_live_reference(hip):
    """Widening row f.2: local_prom_burgers, bases of different widths, a basis switch mid-run."""
    from burgers_hip import rom
    from fem_burgers import FEMBurgers
    g = load_golden("local_pod.npz")
    X, T = mesh(512)
    bases = {c: g[f"basis{c}"] for c in range(4)}
    nT, stride = int(g["nT"]), int(g["stride"])
    for proj in ("Galerkin", "LSPG"):
        res = rom.local_prom_run(X, np.ones(512), [float(g["mu1"]), 5.3], [float(g["mu2"]), 0.017], float(g["At"]), nT,
                                 g["centers"], bases, g["U_global"], 12, projection=proj)
        torch.cuda.synchronize()
        assert rel_l2(res.hist[0].cpu().numpy().T[:, ::stride], g["U_" + proj]) < TOL
        assert np.array_equal(res.iters[0].cpu().numpy(), g["iters_" + proj])
    Uo, ito, cl = br.local_prom_burgers(X, float(g["At"]), 60, np.ones(512), 5.3, 0.0, 0.017, g["centers"], bases,
                                        g["U_global"], 12, projection="LSPG", return_iters=True)
    assert rel_l2(res.hist[1].cpu().numpy().T[:, :61], Uo) < TOL and np.array_equal(res.iters[1].cpu().numpy()[:60], ito)

    class KM:                                             # what the reference's drivers pass (joblib-loaded KMeans)
        cluster_centers_ = g["centers"]
    U = FEMBurgers(X, T).local_prom_burgers(0.05, 10, np.ones(512), float(g["mu1"]), 0.0, float(g["mu2"]), KM(), bases,
                                            g["U_global"], 12, projection="Galerkin")
    assert U.shape == (512, 11) and rel_l2(U[:, ::stride], g["U_Galerkin"][:, :3]) < TOL


def test_library_fallbacks_beyond_kernel_limits(hip):
    """Sizes the register-resident kernels do not cover still run (HIP assembly + library GEMM / LU):
    a quadratic manifold on N = 640 > 512, and local POD with 80-mode bases (r > 47, n > 64)."""
    from burgers_hip import rom
    rng = np.random.default_rng(17)
    N, n = 640, 6
    X, _ = mesh(N)
    xi = np.linspace(0, 1, N)
    modes = np.stack([np.ones(N)] + [np.tanh((xi - c0) * 12) for c0 in (0.2, 0.4, 0.6, 0.8)] + [xi], 1)
    Phi = np.linalg.qr(modes)[0]
    H = 1e-3 * rng.standard_normal((N, n * (n + 1) // 2))
    res = rom.quadratic_run(X, np.ones(N), [4.7, 5.2], [0.02, 0.025], 0.04, 6, Phi, H, projection="LSPG")
    torch.cuda.synchronize()
    for b, (m1, m2) in enumerate([(4.7, 0.02), (5.2, 0.025)]):
        U, ito = br.pod_quadratic_manifold(X, 0.04, 6, np.ones(N), m1, 0.0, m2, Phi, H, return_iters=True)
        assert rel_l2(res.hist[b].cpu().numpy().T, U) < 1e-9 and np.array_equal(res.iters[b].cpu().numpy(), ito)
    g = load_golden("committed_pod_r96.npz")
    lp = load_golden("local_pod.npz")
    X5, _ = mesh(512)
    bases = {c0: np.ascontiguousarray(g["Phi"][:, :w0]) for c0, w0 in zip(range(4), (80, 72, 66, 80))}
    res = rom.local_prom_run(X5, np.ones(512), 4.9, 0.022, 0.05, 12, lp["centers"], bases, lp["U_global"], 12,
                             projection="Galerkin")
    torch.cuda.synchronize()
    U, ito, _ = br.local_prom_burgers(X5, 0.05, 12, np.ones(512), 4.9, 0.0, 0.022, lp["centers"], bases, lp["U_global"], 12,
                                      projection="Galerkin", return_iters=True)
    assert rel_l2(res.hist[0].cpu().numpy().T, U) < 1e-9 and np.array_equal(res.iters[0].cpu().numpy(), ito)


def test_rom_edge_cases_and_bf16_tier(hip):
    """Empty batch, zero steps, projection spelling, and the bf16 MLP tier of config 5 (reported, not gated)."""
    from burgers_hip import rom
    g = load_golden("committed_pod_r40.npz")
    X, _ = mesh(512)
    r = rom.pod_prom_run(X, np.ones(512), np.zeros(0), np.zeros(0), 0.05, 3, g["Phi"], projection="LSPG")
    assert r.hist.shape == (0, 4, 512) and r.iters.shape == (0, 3)
    r = rom.pod_prom_run(X, np.ones(512), 4.75, 0.02, 0.05, 0, g["Phi"], projection="Galerkin")
    assert r.hist.shape == (1, 1, 512) and torch.equal(r.hist[0, 0].cpu(), torch.ones(512, dtype=torch.float64))
    c = load_golden("committed_quadratic_n21.npz")
    for spelling in ("lspg", "LSPG", "Lspg", "GALERKIN"):            # case-insensitive in this variant (:1150)
        rom.quadratic_run(X, np.ones(512), 5.19, 0.026, 0.05, 1, c["Phi"], c["H"], projection=spelling)
    a = load_golden("ann_n5.npz")
    model = _ann_model(a)
    ref = rom.pod_ann_run(X, np.ones(512), 4.56, 0.019, 0.05, 3, a["U_p"], a["U_s"], model)
    low = rom.pod_ann_run(X, np.ones(512), 4.56, 0.019, 0.05, 3, a["U_p"], a["U_s"], model, ann_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    err = rel_l2(low.hist[0].cpu().numpy(), ref.hist[0].cpu().numpy())
    # bf16 closure inside the Newton loop: MEASURED 0.6 % .. 2.2 % relative to the float32 closure over the bench's mu range
    # (tests/fuzz/measure_config5.py); gated at about twice the largest value
    assert np.isfinite(low.hist.cpu().numpy()).all() and err < 5e-2, err


def test_rom_reduce_randomised_sizes(hip):
    """Seeded random (N, r) over the fused kernels' whole range -- every column-block count of the
    4x4x4 kernel, the 16x16x4 kernel above r = 40, odd N (no LDS-DMA prefetch), tiny meshes --
    shared and per-sample bases, both projections, against the oracle."""
    from burgers_hip import rom
    rng = np.random.default_rng(777)
    cases = [(int(rng.integers(2, 513)), int(rng.integers(1, 48))) for _ in range(14)]
    cases += [(2, 1), (3, 2), (511, 40), (512, 41), (257, 24), (64, 47)]
    for N, r in cases:
        X, _ = mesh(N)
        B = int(rng.integers(1, 6))
        mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
        dt, E = 0.04, 0.005
        U = 1.0 + 4.0 * rng.random((B, N)); Un = 1.0 + 4.0 * rng.random((B, N))
        shared = bool(rng.integers(0, 2))
        W = rng.standard_normal((N, r)) if shared else rng.standard_normal((B, N, r))
        c = rom._setup(X, Un, mu1, mu2, dt, E, None)
        G = torch.empty((B, N), dtype=torch.float64, device="cuda")
        rom._mass_rhs(c, _dev(Un), G)
        M3, K3 = br.mass_tridiag(X), br.diffusion_tridiag(X)
        for pname, proj in (("galerkin", 0), ("lspg", 1)):
            Ar = torch.zeros((B, r, r), dtype=torch.float64, device="cuda")
            brr = torch.zeros((B, r), dtype=torch.float64, device="cuda")
            wtu = torch.zeros((B, r), dtype=torch.float64, device="cuda")
            rom.rom_reduce(c, _dev(W), _dev(U), G, proj, True, None, Ar, brr, wtu)
            torch.cuda.synchronize()
            Ar, brr, wtu = Ar.cpu().numpy(), brr.cpu().numpy(), wtu.cpu().numpy()
            for b in range(B):
                Wb = W if shared else W[b]
                lo, di, up = br.system_tridiag(M3, K3, br.convection_tridiag(X, U[b]), dt, E)
                bb = br.tridiag_matvec(*M3, Un[b]) + dt * br.forcing_vector(X, mu2[b]) - dt * br.supg_term(X, U[b], mu2[b])
                bb[0] = mu1[b]
                R = br.tridiag_matvec(lo, di, up, U[b]) - bb
                Ar_ref, br_ref = br._reduce(lo, di, up, R, Wb, pname)
                assert rel_l2(Ar[b], Ar_ref) < 1e-13, (N, r, pname, shared, b)
                assert rel_l2(brr[b], br_ref) < 1e-12, (N, r, pname, shared, b)
                assert rel_l2(wtu[b], Wb.T @ U[b]) < 1e-13


def test_batches_beyond_one_grid_dimension(hip):
    """More than 65535 samples: the setup, mass-RHS and transpose kernels stride over the batch."""
    from burgers_hip import fom, rom
    N, B = 16, 70001
    X, _ = mesh(N)
    rng = np.random.default_rng(3)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    Un = 1.0 + rng.random((B, N))
    c = rom._setup(X, Un, mu1, mu2, 0.05, 0.0, None)
    G = torch.empty((B, N), dtype=torch.float64, device="cuda")
    rom._mass_rhs(c, _dev(Un), G)
    M3 = br.mass_tridiag(X)
    for b in (0, 65534, 65535, 65536, B - 1):
        ref = br.tridiag_matvec(*M3, Un[b]) + 0.05 * br.forcing_vector(X, mu2[b])
        assert rel_l2(G[b].cpu().numpy(), ref) < 1e-13
    t = torch.as_tensor(rng.random((B, 3, 5)), device="cuda")
    assert torch.equal(fom.transpose_batched(t), t.transpose(1, 2).contiguous())


@pytest.mark.parametrize("N,r", [(512, 5), (300, 17), (512, 40), (128, 44)])
def test_rom_reduce_column_major_basis(hip, N, r):
    """BG_OPT_W_COLMAJOR: per-sample (r, N) blocks give bit-identical results to the (N, r) layout."""
    from burgers_hip import rom
    rng = np.random.default_rng(N + r)
    X, _ = mesh(N)
    B = 5
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    U = 1.0 + 4.0 * rng.random((B, N)); Un = 1.0 + 4.0 * rng.random((B, N))
    c = rom._setup(X, Un, mu1, mu2, 0.05, 0.0, None)
    G = torch.empty((B, N), dtype=torch.float64, device="cuda")
    rom._mass_rhs(c, _dev(Un), G)
    for W in (rng.standard_normal((B, N, r)), rng.standard_normal((N, r))):
        Wd = _dev(W)
        WdT = Wd.transpose(-1, -2).contiguous()
        for proj in (0, 1):
            out = []
            for colmajor in (False, True):
                Ar = torch.zeros((B, r, r), dtype=torch.float64, device="cuda")
                brr = torch.zeros((B, r), dtype=torch.float64, device="cuda")
                wtu = torch.zeros((B, r), dtype=torch.float64, device="cuda")
                rom.rom_reduce(c, WdT if colmajor else Wd, _dev(U), G, proj, True, None, Ar, brr, wtu, colmajor=colmajor)
                out.append((Ar, brr, wtu))
            torch.cuda.synchronize()
            for a, b in zip(*out):
                assert torch.equal(a, b)


def test_offline_basis_builders_on_device(hip):
    """POD basis (device QR + one-sided Jacobi, bg_jacobi_sweep) and quadratic-manifold fit built on the device
    from a device-resident sweep agree with a host LAPACK SVD of the same snapshots: every mode down to sigma/sigma_1 = 1e-8 (rocSOLVER's own SVD
    loses those: 1e-1 error, tools/time_pod.py), H to 1e-7.  reference: POD/pod.py:80-90,
    Quadratic_manifold/build_quadratic_manifold.py:25-48."""
    from burgers_hip import fom, pod
    N = 256
    X, _ = mesh(N)
    m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 2), indexing="ij")
    res = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 200)
    S = pod.snapshot_matrix(res.hist).contiguous()
    Sc = S.cpu()
    Uc, sc, _ = torch.linalg.svd(Sc, full_matrices=False)
    U, s, s_all = pod.pod_basis(S, epsilon_squared=1e-6)
    K = pod.n_modes_for_tolerance(sc, 1e-6)
    assert U.shape == (N, K) and U.is_cuda
    assert float(((s_all.cpu() - sc).abs() / sc[0]).max()) < 1e-12
    Ud, _, _ = pod.thin_svd(S)
    keep = int((sc / sc[0] > 1e-8).sum())
    assert keep > 100
    Ua = pod.align_signs(Ud[:, :keep].cpu(), Uc[:, :keep])
    assert float((Ua - Uc[:, :keep]).abs().max()) < 1e-8
    assert float((Ua[:, :K] - Uc[:, :K]).abs().max()) < 1e-11
    n = 12
    Pg, Hg, _ = pod.build_quadratic_manifold(S, n, alpha=1e-2)
    Pc, Hc, _ = pod.build_quadratic_manifold(Sc, n, alpha=1e-2)
    sg = torch.sign((Pg.cpu() * Pc).sum(0)); I, J = np.triu_indices(n)
    assert float((Pg.cpu() * sg - Pc).abs().max()) < 1e-11
    assert float(torch.linalg.norm(Hg.cpu() * (sg[I] * sg[J]) - Hc) / torch.linalg.norm(Hc)) < 1e-7


@pytest.mark.parametrize("m", [1, 2, 7, 64, 129])
def test_jacobi_svd_core(hip, m):
    """bg_jacobi_sweep on graded random matrices (condition 1e10): singular values, orthogonal factors,
    reconstruction, all at the eps * sigma_1 level the input itself carries."""
    from burgers_hip import pod
    rng = np.random.default_rng(m)
    U0, _ = np.linalg.qr(rng.standard_normal((m, m)))
    V0, _ = np.linalg.qr(rng.standard_normal((m, m)))
    s0 = np.logspace(0, -10, m) if m > 1 else np.array([3.0])
    A = (U0 * s0) @ V0.T
    U, s, Vh = pod.jacobi_svd(_dev(A))
    U, s, Vh = U.cpu().numpy(), s.cpu().numpy(), Vh.cpu().numpy()
    assert np.all(np.diff(s) <= 0)
    assert np.abs(s - s0).max() / s0[0] < 1e-13 and np.abs(s / s0 - 1).max() < 1e-5    # A itself carries eps*sigma_1
    assert np.abs(U.T @ U - np.eye(m)).max() < 1e-13 and np.abs(Vh @ Vh.T - np.eye(m)).max() < 1e-13
    assert np.abs((U * s) @ Vh - A).max() < 1e-13


def test_rbf_closure_kernel(hip):
    """bg_rbf_eval + the two GEMMs against the closure value / Jacobian recorded from the reference
    (FEM/fem_burgers.py:160-260) and against the oracle on a random batch, both kernels."""
    from burgers_hip import rom
    g = load_golden("rbf_n17.npz")
    rng = np.random.default_rng(5)
    for kernel in ("gaussian", "imq"):
        args = (g["X_train"], g["W_" + kernel], float(g["eps_" + kernel]), kernel, g["x_min"], g["x_max"], g["y_min"], g["y_max"])
        cl = rom.RbfClosure(*args, torch.device("cuda", 0))
        qp = np.vstack([g["qp_" + kernel], g["qp_" + kernel] + 0.05 * rng.standard_normal((6, g["qp_" + kernel].size))])
        val = cl.value(_dev(qp)).cpu().numpy()
        jac = cl.jacobian(_dev(qp)).cpu().numpy()
        assert np.abs(val[0] - g["val_" + kernel]).max() < 1e-11 * max(1.0, np.abs(g["val_" + kernel]).max())
        assert np.abs(jac[0] - g["jac_" + kernel]).max() < 1e-10 * np.abs(g["jac_" + kernel]).max()
        for b in range(1, 7):
            assert np.abs(val[b] - br.rbf_value(qp[b], *args)).max() < 1e-11 * max(1.0, np.abs(val[b]).max())
            jo = br.rbf_jacobian(qp[b], *args)
            assert np.abs(jac[b] - jo).max() < 1e-10 * np.abs(jo).max()


def test_both_mfma_kernels_agree(hip):
    """The 16x16x4 kernel (default only for 40 < r <= 47) forced for r = 40 / 24 (BG_OPT_MFMA_16X16) gives the 4x4x4 kernel's results."""
    from burgers_hip import rom
    rng = np.random.default_rng(99)
    N, B = 512, 6
    X, _ = mesh(N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    U = 1.0 + 4.0 * rng.random((B, N)); Un = 1.0 + 4.0 * rng.random((B, N))
    c = rom._setup(X, Un, mu1, mu2, 0.05, 0.0, None)
    G = torch.empty((B, N), dtype=torch.float64, device="cuda")
    rom._mass_rhs(c, _dev(Un), G)
    for r in (40, 24):
        W = _dev(np.linalg.qr(rng.standard_normal((N, r)))[0])
        for proj in (0, 1):
            out = []
            for force in (False, True):
                Ar = torch.zeros((B, r, r), dtype=torch.float64, device="cuda")
                brr = torch.zeros((B, r), dtype=torch.float64, device="cuda")
                wtu = torch.zeros((B, r), dtype=torch.float64, device="cuda")
                rom.rom_reduce(c, W, _dev(U), G, proj, True, None, Ar, brr, wtu,
                               extra_opts=hip.BG_OPT_MFMA_16X16 if force else 0)
                torch.cuda.synchronize()
                out.append((Ar.cpu().numpy(), brr.cpu().numpy(), wtu.cpu().numpy()))
            for a, b in zip(*out):
                assert rel_l2(a, b) < 1e-13


def test_closure_roms_cap_pattern_matches_the_oracle(hip):
    """bench.py --config ann reports about half of its samples with BG_FLAG_HIT_CAP.  That is the algorithm, not the
    kernels: on the bench's own (mu1, mu2) draw the oracle runs into the 50-iteration cap on exactly the same samples
    and time steps (the first step of the samples with mu1 > 4.9), and the POD-RBF closure into its 30-iteration cap
    likewise.  POD-ANN iteration counts may differ by one where float32 noise sits on the threshold."""
    import bench
    from burgers_hip import rom
    from test_rom_large_batch_gpu import _ann_model
    X, _ = mesh(512)
    mu1a, mu2a = bench.mu_shard(2048, 1, 0)
    idx = np.linspace(0, 2047, 8).astype(int)
    mu1, mu2 = mu1a[idx], mu2a[idx]
    g = load_golden("ann_n5.npz")
    nT = 12
    res = rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, nT, g["U_p"], g["U_s"], _ann_model(g))
    torch.cuda.synchronize()
    Ws = [g[f"W{i}"] for i in range(6)]; bs = [g[f"b{i}"] for i in range(6)]
    it = res.iters.cpu().numpy(); fl = res.flags.cpu().numpy()
    capped = 0
    for b in range(len(idx)):
        Uo, ito = br.pod_ann_prom(X, 0.05, nT, np.ones(512), mu1[b], 0.0, mu2[b], g["U_p"], g["U_s"], Ws, bs, return_iters=True)
        assert np.array_equal(it[b] >= 50, ito >= 50), b                        # same capped steps
        assert bool(fl[b] & 1) == bool((ito >= 50).any()) and np.abs(it[b] - ito).max() <= 1
        assert rel_l2(res.hist[b].cpu().numpy().T, Uo) < 5e-6
        capped += int((ito >= 50).any())
    assert 0 < capped < len(idx)                                                # the draw has both kinds
    r = load_golden("rbf_n17.npz")
    cl = (r["U_p"], r["U_s"], r["X_train"], r["W_gaussian"], float(r["eps_gaussian"]), r["x_min"], r["x_max"], r["y_min"], r["y_max"])
    res = rom.pod_rbf_run(X, np.ones(512), mu1[:4], mu2[:4], 0.05, nT, *cl)
    torch.cuda.synchronize()
    for b in range(4):
        Uo, ito = br.pod_rbf_prom(X, 0.05, nT, np.ones(512), mu1[b], 0.0, mu2[b], *cl, return_iters=True)
        assert np.array_equal(res.iters[b].cpu().numpy(), ito) and rel_l2(res.hist[b].cpu().numpy().T, Uo) < 1e-9


@pytest.mark.parametrize("N,n,B,Nt", [(512, 160, 5, 501), (64, 16, 3, 7), (96, 256, 2, 130), (512, 160, 1, 1), (32, 48, 300, 3)])
def test_decode_modes_bf16_kernel(hip, N, n, B, Nt):
    """bg_decode_modes_bf16 (bf16 MFMA, float32 accumulate, float64 result in the (B, N, Nt) snapshot layout) against the
    same product of the bf16-rounded operands in float64: columns that straddle samples, a ragged last workgroup, every
    k-block count class; then the refusals."""
    from burgers_hip import lib as L_
    L = L_.load()
    g = torch.Generator(device="cuda").manual_seed(N + n)
    Um = torch.randn((N, n), device="cuda", generator=g).to(torch.bfloat16)
    Q = torch.randn((B * Nt, n), device="cuda", generator=g).to(torch.bfloat16)
    out = torch.full((B, N, Nt), -7.0, dtype=torch.float64, device="cuda")
    L_.check(L.bg_decode_modes_bf16(N, n, B, Nt, L_.ptr(Um), L_.ptr(Q), L_.ptr(out), L_.stream_ptr(torch.device("cuda", 0))),
             "bg_decode_modes_bf16")
    torch.cuda.synchronize()
    ref = torch.matmul(Um.double(), Q.double().reshape(B, Nt, n).transpose(1, 2))
    assert float((out - ref).abs().max()) < 2e-5 * float(ref.abs().max())          # float32 accumulation over n terms
    z = ctypes.c_void_p(0)
    assert L.bg_decode_modes_bf16(33, 16, 1, 1, z, z, z, z) == hip.BG_ERR_UNSUPPORTED_N
    assert L.bg_decode_modes_bf16(32, 20, 1, 1, z, z, z, z) == hip.BG_ERR_UNSUPPORTED_R
    assert L.bg_decode_modes_bf16(32, 16, 1, 1, z, z, z, z) == hip.BG_ERR_BAD_ARG
    assert L.bg_decode_modes_bf16(32, 16, 0, 1, z, z, z, z) == hip.BG_OK
