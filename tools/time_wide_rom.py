#!/usr/bin/env python3
"""Throughput of bg_rom_run_wide (POD PROM, 41 .. 96 modes) against the library path, committed r = 96 basis.
usage: python tools/time_wide_rom.py [--batch 1024] [--steps 40] [--r 96] [--library]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--r", type=int, default=96); ap.add_argument("--library", action="store_true")
ap.add_argument("--phases", action="store_true", help="BG_WIDE_TIMING build (BG_LIB_PATH): kilo-clocks per phase and pass")
a = ap.parse_args()
from burgers_hip import rom
g = np.load(os.path.join(REPO, "tests", "golden", "committed_pod_r96.npz"))
Phi = np.ascontiguousarray(g["Phi"][:, :a.r])
X = np.linspace(0, 100, 512)
rng = np.random.default_rng(20251121)
mu1, mu2 = rng.uniform(4.25, 5.5, a.batch), rng.uniform(0.015, 0.03, a.batch)
for proj in ("Galerkin", "LSPG"):
    run = lambda: rom.pod_prom_run(X, np.ones(512), mu1, mu2, 0.05, a.steps, Phi, projection=proj, fused=not a.library)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); res = run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); its = int(res.iters.sum().item())
    if a.phases:
        it = res.iters[:, :7].double().cpu().numpy(); npass = np.median(it[:, 6])
        names = ["wait slab + barrier", "lift + assembly", "projection (+ coefficient barrier)", "park", "solve", "update / per-step / pass start"]
        print(f"{proj}: {ms:.1f} ms, passes per sample (median) {npass:.0f}; kilo-clocks per pass: " +
              ", ".join(f"{nm} {np.median(it[:, i]) * 1.024 / npass:.1f}" for i, nm in enumerate(names)) +
              f"; total {np.median(it[:, :6].sum(1)) * 1.024 / npass:.1f} k")
        continue
    print(f"{'library path' if a.library else 'bg_rom_run_wide'} {proj} r={a.r} B={a.batch} steps={a.steps}: {ms:.1f} ms, {its} sample-iterations, "
          f"{its / ms * 1e3:.3g} sample-Newton-steps/s" + (f", handed back {res.redone}" if hasattr(res, 'redone') else ""))
