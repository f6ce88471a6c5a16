#!/usr/bin/env python3
"""Time bg_rom_run alone (HIP events) on one MI355X: microseconds per sample-iteration per workgroup.
With BG_LIB_PATH pointing at an ablation build (see csrc/rom_fused.hip, BG_FUSED_ABLATE) the iteration count is
fixed at 5 per time step, so builds with phases compiled out can be subtracted from each other.
usage: python tools/time_fused.py [--batch 256] [--steps 40] [--proj Galerkin|LSPG] [--r 40]"""
import argparse, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--same-mu", action="store_true", help="every sample the same (mu1, mu2): equal iteration counts")
ap.add_argument("--force-pivoted", action="store_true")
ap.add_argument("--proj", default="Galerkin"); ap.add_argument("--r", type=int, default=40); ap.add_argument("--n", type=int, default=512)
a = ap.parse_args()
from burgers_hip import rom, lib
g = np.load(os.path.join(REPO, "tests", "golden", "committed_pod_r40.npz"))
Phi = g["Phi"][:a.n, :a.r] if a.n == 512 else np.linalg.qr(np.random.default_rng(0).standard_normal((a.n, a.r)))[0]
X = np.linspace(0, 100, a.n)
rng = np.random.default_rng(1)
mu1, mu2 = rng.uniform(4.25, 5.5, a.batch), rng.uniform(0.015, 0.03, a.batch)
if a.same_mu:
    mu1[:], mu2[:] = 4.9, 0.022
pj = rom.PROJ[a.proj.lower()]
run = lambda: rom.pod_prom_run_fused(X, np.ones(a.n), mu1, mu2, 0.05, a.steps, Phi, pj, options=lib.BG_OPT_FORCE_PIVOTED if a.force_pivoted else 0)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record(); res = run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
timing_build = "abl" in os.environ.get("BG_LIB_PATH", "")
if timing_build:
    st = res.iters[:, :2].double().cpu().numpy(); clk = np.median(st[:, 0] / st[:, 1]) * 100.0
    if a.steps >= 8:
        ph = np.median(res.iters[:, 2:8].double().cpu().numpy(), axis=0) * 1024.0 / (5.0 * a.steps)
        print("in-kernel shader clocks per iteration (median over samples): " +
              ", ".join(f"{nm} {v:.0f}" for nm, v in zip(["assembly", "mfma", "solve", "update", "lift", "per-step"], ph)) + f", total {ph.sum():.0f}")
    res.iters[:, :8] = 5
    res.iters[:, :2] = 5
its = int(res.iters.sum().item())
cus = torch.cuda.get_device_properties(0).multi_processor_count
waves = -(-a.batch // cus)
if timing_build:
    print(f"in-kernel clock (median over samples): {clk:.0f} MHz")
print(f"{os.path.basename(os.environ.get('BG_LIB_PATH', 'product'))}: {a.proj} r={a.r} B={a.batch} steps={a.steps}: {best:.2f} ms, {its} sample-iterations, "
      f"{best * 1e3 / (its / a.batch * waves):.2f} us per sample-iteration per workgroup, {its / best * 1e3:.3g} sample-steps/s")
