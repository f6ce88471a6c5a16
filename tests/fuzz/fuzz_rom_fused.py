#!/usr/bin/env python3
"""One-off differential fuzz of bg_rom_run (device-side POD-PROM loop: Galerkin, and LSPG in its pentadiagonal form
Phi^T (A^T A Phi)) against the oracle's pod_prom_burgers: random mesh sizes, uniform / non-uniform meshes, reduced
dimensions 1 .. 40 (every column-block count), orthonormal random bases mixed with smooth modes, diffusion on / off,
time steps.  Reports the worst relative L2 distance of the trajectories and every iteration-count mismatch.
usage: fuzz_rom_fused.py [n_cases] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import rom
from oracle import burgers_ref as br

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
worst = {"galerkin": 0.0, "lspg": 0.0}
mism = 0; ran = 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([rng.integers(8, 513), 512, 256, 257, 64]))
    r = int(rng.integers(1, min(40, N - 2) + 1))
    X = np.linspace(0.0, 100.0, N)
    if rng.random() < 0.3:
        w = rng.uniform(0.7, 1.3, N - 1); X = np.concatenate([[0.0], np.cumsum(w)]) * (100.0 / w.sum())
    # a basis that can represent the solution family reasonably: smooth modes + noise, orthonormalised
    k = np.arange(r)[None, :]
    smooth = np.cos(np.pi * (k + 0.5) * (X[:, None] / 100.0)) + 0.05 * rng.standard_normal((N, r))
    smooth[:, 0] = 1.0
    Phi, _ = np.linalg.qr(smooth)
    B = int(rng.integers(1, 4))
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt, E = float(rng.choice([0.05, 0.025, 0.07])), float(rng.choice([0.0, 0.0, 0.004]))
    nT = int(rng.integers(2, 6))
    u0 = np.ones(N)
    for pname in ("Galerkin", "LSPG"):
        try:
            res = rom.pod_prom_run_fused(X, u0, mu1, mu2, dt, nT, Phi, rom.PROJ[pname.lower()], E=E)
            torch.cuda.synchronize()
        except Exception as e:
            print(f"case {case}: N={N} r={r} {pname}: {type(e).__name__}: {e}"); continue
        info = res.info.cpu().numpy()
        for b in range(B):
            try:
                U, it = br.pod_prom_burgers(X, dt, nT, u0, mu1[b], E, mu2[b], Phi, projection=pname, return_iters=True)
            except np.linalg.LinAlgError:
                assert info[b] != 0, f"case {case}: the oracle met a singular system, the kernel did not"
                continue
            if not np.isfinite(U).all():
                continue                                   # the reference itself diverged on this random basis
            ran += 1
            e = float(np.linalg.norm(res.hist[b].cpu().numpy().T - U) / np.linalg.norm(U))
            worst[pname.lower()] = max(worst[pname.lower()], e)
            same = np.array_equal(res.iters[b].cpu().numpy(), it)
            if e > 1e-9 or not same:
                mism += 1
                print(f"MISMATCH case {case}: N={N} r={r} B={B} {pname} dt={dt} E={E} nonuniform={not np.allclose(np.diff(X), X[1] - X[0])} "
                      f"b={b}: rel-L2 {e:.2e}, iters {res.iters[b].cpu().numpy().tolist()} vs {list(it)}", flush=True)
print(f"{n_cases} cases, {ran} sample runs in {time.time() - t0:.0f} s: worst rel-L2 Galerkin {worst['galerkin']:.2e}, LSPG {worst['lspg']:.2e}; mismatches {mism}")
