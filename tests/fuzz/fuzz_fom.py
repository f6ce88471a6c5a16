#!/usr/bin/env python3
"""One-off differential fuzz of bg_fom_run / bg_fd_run against the oracles over many random sizes and settings.
usage: fuzz_fom.py [n_cases] [seed]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import fom
from oracle import burgers_ref as br
from oracle import burgers_ref_c as bc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
worst, bad = 0.0, 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([int(rng.integers(2, 129)), int(rng.integers(129, 1537)), int(rng.integers(1537, 8193))]))
    B = int(rng.integers(1, 6)); nsteps = int(rng.integers(2, 7))
    X = np.linspace(0.0, 100.0, N)
    graded = rng.random() < 0.35 and N > 3
    if graded:
        w = rng.uniform(0.6, 1.4, N - 1); X = np.concatenate([[0.0], np.cumsum(w)]) * (100.0 / w.sum())
    E = float(rng.choice([0.0, 0.0, 0.003, 0.02]))
    h = 100.0 / (N - 1)
    dt = float(min(0.4, rng.uniform(0.08, 0.2) * h))
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    u0 = 1.0 + 0.1 * np.sin(np.outer(rng.uniform(0.02, 0.1, B), X)) * np.exp(-X / 60.0)
    supg = bool(rng.random() < 0.85)
    r = fom.fom_run(X, u0, mu1, mu2, dt, nsteps, E=E, supg=supg); torch.cuda.synchronize()
    ho, ito = bc.fom_run(X, u0, mu1, mu2, dt, nsteps, E=E, supg=supg)
    e = float(np.linalg.norm(r.hist.cpu().numpy() - ho) / np.linalg.norm(ho))
    ok = e < 1e-10 and np.array_equal(r.iters.cpu().numpy(), ito)
    worst = max(worst, e)
    if not ok:
        bad += 1
        print(f"FOM MISMATCH case {case}: N={N} B={B} graded={graded} E={E} dt={dt:.5f} supg={supg} rel={e:.2e} iters_equal={np.array_equal(r.iters.cpu().numpy(), ito)}", flush=True)
    if case % 4 == 0 and N >= 3:                       # FD stepper on the uniform mesh of the same size
        Nf = max(N, 3)
        rf = fom.fd_run(0.0, 100.0, Nf, np.ones(Nf), mu1[:2], mu2[:2], dt, 3); torch.cuda.synchronize()
        for b in range(min(2, B)):
            Uo, itf = br.fd_newton(0.0, 100.0, Nf, dt, 3, np.ones(Nf), mu1[b], mu2[b], return_iters=True)
            ef = float(np.linalg.norm(rf.hist[b].cpu().numpy().T - Uo) / np.linalg.norm(Uo))
            if not (ef < 1e-10 and np.array_equal(rf.iters[b].cpu().numpy(), itf)):
                bad += 1
                print(f"FD MISMATCH case {case}: N={Nf} dt={dt:.5f} rel={ef:.2e}", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, worst FOM rel-L2 {worst:.2e}, mismatches {bad}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, worst FOM rel-L2 {worst:.2e}, mismatches {bad}")
sys.exit(1 if bad else 0)
