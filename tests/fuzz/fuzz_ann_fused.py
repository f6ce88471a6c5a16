#!/usr/bin/env python3
"""One-off differential fuzz of bg_ann_rom_run (device-side POD-ANN loop) against the host-driven batched path and, for
ELU networks, the oracle: random mesh sizes, reduced dimensions, layer counts and widths (every fold path of the in-kernel
MLP: spans 2 ... 64, 6- and 9-row variants, ragged widths), activations, biases, projections.
usage: fuzz_ann_fused.py [n_cases] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
import torch.nn as nn
from burgers_hip import fom, pod, rom
from oracle import burgers_ref as br

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
bases = {}


def basis(N):
    if N not in bases:
        X = np.linspace(0.0, 100.0, N)
        m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 2), indexing="ij")
        snap = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 100)
        bases[N] = pod.pod_basis(pod.snapshot_matrix(snap.hist).contiguous(), n_modes=min(N - 2, 136))[0].cpu().numpy()
    return bases[N]


bad, worst, compared, empty, oracle_checked = 0, 0.0, 0, 0, 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([64, 100, 255, 256, 257, 301, 400, 511, 512]))
    n = int(rng.integers(1, 9))
    nbar = int(rng.integers(1, min(128, N - 2 - n) + 1))
    nh = int(rng.integers(0, 7))
    hidden = [int(rng.choice([1, 3, 7, 8, 16, 17, 32, 33, 64, 100, 128, 130, 200, 255, 256])) for _ in range(nh)]
    act = [nn.ELU, nn.ReLU, nn.Tanh][int(rng.integers(0, 3))]
    bias = bool(rng.random() < 0.7)
    Phi = basis(N)
    U_p, U_s = Phi[:, :n], Phi[:, n:n + nbar]
    torch.manual_seed(case + 1000 * seed)
    widths = [n] + hidden + [nbar]
    mods = []
    for i in range(len(widths) - 1):
        lin = nn.Linear(widths[i], widths[i + 1], bias=bias)
        with torch.no_grad():
            lin.weight.mul_(0.5)
        mods.append(lin)
        if i < len(widths) - 2:
            mods.append(act(alpha=float(rng.choice([1.0, 0.7]))) if act is nn.ELU else act())
    with torch.no_grad():
        mods[-1].weight.mul_(0.02)
        if bias:
            mods[-1].bias.mul_(0.02)
    model = nn.Sequential(*mods).eval()
    B, nT = int(rng.integers(1, 9)), 4
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    X = np.linspace(0.0, 100.0, N)
    E = float(rng.choice([0.0, 0.002]))
    proj = str(rng.choice(["LSPG", "Galerkin"]))
    f = rom.pod_ann_run(X, np.ones(N), mu1, mu2, 0.05, nT, U_p, U_s, model, projection=proj, E=E)
    b = rom.pod_ann_run(X, np.ones(N), mu1, mu2, 0.05, nT, U_p, U_s, model, projection=proj, E=E, fused=False)
    torch.cuda.synchronize()
    tag = f"case {case}: N={N} n={n} nbar={nbar} hidden={hidden} {act.__name__} bias={bias} {proj} E={E} B={B}"
    if not hasattr(f, "info"):
        print("NOT FUSED", tag); bad += 1; continue
    ok = ((f.flags == 0) & (b.flags == 0)).cpu().numpy()
    fi, bi = f.iters.cpu().numpy(), b.iters.cpu().numpy()
    fh, bh = f.hist.cpu().numpy(), b.hist.cpu().numpy()
    errs = [rel(fh[s], bh[s]) for s in np.flatnonzero(ok)]
    compared += len(errs); empty += int(not ok.any())
    e = max(errs) if errs else 0.0
    it_ok = (np.abs(fi[ok] - bi[ok]).max() <= 1) if ok.any() else True
    oracle_e = 0.0
    if act is nn.ELU and ok.any() and all(getattr(m, "alpha", 1.0) == 1.0 for m in mods):
        s = int(np.flatnonzero(ok)[0])
        lins = [m for m in mods if isinstance(m, nn.Linear)]
        Ws = [m.weight.detach().cpu().numpy() for m in lins]
        bs = [np.zeros(m.out_features, np.float32) if m.bias is None else m.bias.detach().cpu().numpy() for m in lins]
        Uo = br.pod_ann_prom(X, 0.05, nT, np.ones(N), mu1[s], E, mu2[s], U_p, U_s, Ws, bs, projection=proj)
        oracle_e = rel(fh[s].T, Uo)
        oracle_checked += 1
    worst = max(worst, e, oracle_e)
    if e > 5e-6 or oracle_e > 5e-6 or not it_ok or not np.isfinite(fh[ok]).all():
        bad += 1
        print("MISMATCH", tag, "vs batched", e, "vs oracle", oracle_e, "iters ok", it_ok)
print(f"{n_cases} cases, {bad} bad, {compared} samples compared with the batched path ({empty} cases with none converged in both), "
      f"{oracle_checked} oracle runs, worst rel-L2 {worst:.2e}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
