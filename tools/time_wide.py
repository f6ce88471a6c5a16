#!/usr/bin/env python3
"""Throughput of the workgroup-per-sample FOM kernels (2048 < N <= 8192) beside the wave-per-sample ones."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import fom
rng = np.random.default_rng(0)
SIZES = eval(os.environ.get('SIZES', '((1024, 1024), (2048, 1024), (3072, 256), (4096, 256), (6144, 256), (8192, 256), (4096, 1024))'))
for N, B in SIZES:
    X = np.linspace(0, 100, N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    dt, nT = 0.05 * 512 / N, 50
    fom.fom_run(X, np.ones(N), mu1[:8], mu2[:8], dt, 3); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nT); torch.cuda.synchronize(); t = time.perf_counter() - t0
    steps = int(r.iters.sum())
    print(f"N={N:5d} B={B:5d}: {t*1e3:8.1f} ms, {steps/t:.3e} sample-Newton-steps/s, {steps*N/t:.3e} row-steps/s")
