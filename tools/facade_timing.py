#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in facade (device kernel + transpose + device->host copy)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from fem_burgers import FEMBurgers
from burgers_hip import fom
N, B, nT, dt = 1024, 1024, 500, 0.025
X = np.linspace(0, 100, N); T = np.array([np.arange(1, N), np.arange(2, N + 1)]).T
rng = np.random.default_rng(20251121)
mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
fem = FEMBurgers(X, T)
fem.fom_burgers(dt, 5, np.ones(N), mu1[:8], 0.0, mu2[:8])
for label in ("first call (pins the host block)", "steady state (cached pinned block)", "steady state"):
    t0 = time.perf_counter(); U = fem.fom_burgers(dt, nT, np.ones(N), mu1, 0.0, mu2); t = time.perf_counter() - t0
    steps = int(fem.last_iters.sum())
    print(f"facade, {label}: host ndarray out, {U.nbytes/1e9:.2f} GB over PCIe: {t*1e3:.1f} ms -> {steps/t:.3e} sample-Newton-steps/s")
    chk = float(U[3, 5, -1]); del U
os.environ["BG_PINNED_RESULTS"] = "0"
t0 = time.perf_counter(); U = fem.fom_burgers(dt, nT, np.ones(N), mu1, 0.0, mu2); t = time.perf_counter() - t0
print(f"facade, pageable (BG_PINNED_RESULTS=0): {t*1e3:.1f} ms -> {steps/t:.3e}; same value: {float(U[3, 5, -1]) == chk}")
del U
res = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nT); torch.cuda.synchronize()
t0 = time.perf_counter(); res = fom.fom_run(X, np.ones(N), mu1, mu2, dt, nT); torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"device-resident fom_run: {t*1e3:.1f} ms -> {steps/t:.3e}")
t0 = time.perf_counter(); r = fom.fd_run(0.0, 100.0, N, np.ones(N), mu1, mu2, dt, nT); torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"FD Newton stepper, same sweep: {t*1e3:.1f} ms, {int(r.iters.sum())} Newton steps -> {int(r.iters.sum())/t:.3e} steps/s")
