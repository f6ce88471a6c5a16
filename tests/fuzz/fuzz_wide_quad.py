#!/usr/bin/env python3
"""One-off differential fuzz of the round-3 device-side loops against the oracle:
  bg_rom_run_wide   r = 41 .. 96, random smooth + noise orthonormal bases, uniform / non-uniform meshes, both projections
  bg_quad_rom_run   n = 2 .. 40, the same kind of basis with a small random quadratic correction H, both projections
usage: fuzz_wide_quad.py [n_cases] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import rom
from oracle import burgers_ref as br

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = torch.device("cuda", 0)
def basis(X, r):
    k = np.arange(r)[None, :]
    S = np.cos(np.pi * (k + 0.5) * (X[:, None] / 100.0)) + 0.05 * rng.standard_normal((len(X), r))
    S[:, 0] = 1.0
    return np.linalg.qr(S)[0]
def mesh(N):
    X = np.linspace(0.0, 100.0, N)
    if rng.random() < 0.3:
        w = rng.uniform(0.7, 1.3, N - 1); X = np.concatenate([[0.0], np.cumsum(w)]) * (100.0 / w.sum())
    return X
worst = {"wide": 0.0, "quad": 0.0}; mism = 0; ran = {"wide": 0, "quad": 0}
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([rng.integers(128, 513), 512, 257]))
    X = mesh(N); u0 = np.ones(N)
    B = int(rng.integers(1, 4)); mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt = float(rng.choice([0.05, 0.025])); nT = int(rng.integers(2, 4))
    # ---- wide
    r = int(rng.integers(41, min(96, N - 2) + 1)); Phi = basis(X, r)
    for pname in ("Galerkin", "LSPG"):
        res = rom.pod_prom_run_wide(X, u0, mu1, mu2, dt, nT, Phi, rom.PROJ[pname.lower()]); torch.cuda.synchronize()
        for b in range(B):
            try:
                U, it = br.pod_prom_burgers(X, dt, nT, u0, mu1[b], 0.0, mu2[b], Phi, projection=pname, return_iters=True)
            except np.linalg.LinAlgError:
                continue
            if not np.isfinite(U).all(): continue
            ran["wide"] += 1
            e = float(np.linalg.norm(res.hist[b].cpu().numpy().T - U) / np.linalg.norm(U)); worst["wide"] = max(worst["wide"], e)
            if e > 1e-9 or not np.array_equal(res.iters[b].cpu().numpy(), it):
                mism += 1; print(f"WIDE MISMATCH case {case}: N={N} r={r} {pname} b={b} redone={getattr(res, 'redone', None)}: {e:.2e} iters {res.iters[b].tolist()} vs {list(it)}", flush=True)
    # ---- quadratic manifold
    n = int(rng.integers(2, 41)); Phi = basis(X, n); k = n * (n + 1) // 2
    H = 0.02 * rng.standard_normal((N, k)) / np.sqrt(N)
    plan = rom.QuadFusedPlan(Phi, H, dev)
    for pname in ("LSPG", "Galerkin"):
        res = rom.quadratic_run_fused(X, u0, mu1, mu2, dt, nT, plan, rom.PROJ[pname.lower()]); torch.cuda.synchronize()
        info = res.info.cpu().numpy()
        for b in range(B):
            try:
                U, it = br.pod_quadratic_manifold(X, dt, nT, u0, mu1[b], 0.0, mu2[b], Phi, H, projection=pname, return_iters=True)
            except np.linalg.LinAlgError:
                assert info[b] != 0; continue
            if not np.isfinite(U).all() or info[b] != 0: continue
            ran["quad"] += 1
            e = float(np.linalg.norm(res.hist[b].cpu().numpy().T - U) / np.linalg.norm(U)); worst["quad"] = max(worst["quad"], e)
            if e > 1e-8 or not np.array_equal(res.iters[b].cpu().numpy(), it):
                mism += 1; print(f"QUAD MISMATCH case {case}: N={N} n={n} {pname} b={b}: {e:.2e} iters {res.iters[b].tolist()} vs {list(it)}", flush=True)
print(f"{n_cases} cases in {time.time() - t0:.0f} s: bg_rom_run_wide {ran['wide']} runs, worst rel-L2 {worst['wide']:.2e}; bg_quad_rom_run {ran['quad']} runs, worst {worst['quad']:.2e}; mismatches {mism}")
