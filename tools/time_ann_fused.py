#!/usr/bin/env python3
"""Time bg_ann_rom_run alone (HIP events) on one MI355X: microseconds per sample-iteration per workgroup slot.
With BG_LIB_PATH pointing at an ablation build (-DBG_FUSED_ABLATE=bits: 1 MFMA pass, 2 elimination, 16 assembly,
128 MLP layers, 256 closure sweep; 8192: the "mlp layer 0..4" lines then are sub-phases summed over the layers --
loads + FMAs, DPP folds, swap fold + activation + write, barrier wait, second stage) the iteration count is fixed at 5 per time step, so builds can be subtracted.
usage: python tools/time_ann_fused.py [--batch 512] [--steps 20] [--proj LSPG|Galerkin] [--fused 0|1]"""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--same-mu", action="store_true"); ap.add_argument("--proj", default="LSPG"); ap.add_argument("--fused", type=int, default=1)
a = ap.parse_args()
from burgers_hip import rom
import bench
g = bench.golden("ann_n5.npz")
model = bench.ann_model(g)
X = np.linspace(0, 100, 512)
rng = np.random.default_rng(1)
mu1, mu2 = rng.uniform(4.25, 5.5, a.batch), rng.uniform(0.015, 0.03, a.batch)
if a.same_mu:
    mu1[:], mu2[:] = 4.6, 0.022
run = lambda: rom.pod_ann_run(X, np.ones(512), mu1, mu2, 0.05, a.steps, g["U_p"], g["U_s"], model, projection=a.proj, fused=bool(a.fused))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record(); res = run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
timing_build = "abl" in os.environ.get("BG_LIB_PATH", "")
if timing_build and a.steps >= 16:
    ph = res.iters[:, :16].double().cpu().numpy() * 1024.0 / (5.0 * a.steps)          # shader clocks per iteration and phase
    names = ["assembly+fragments", "mfma pass", "solve+update"] + [f"mlp layer {i}" for i in range(8)] + ["mlp output table", "sweep", "per-step work", "mlp input set-up", "-"]
    med = np.median(ph, axis=0)
    print("in-kernel shader clocks per iteration (median over samples):")
    for nm, v in zip(names, med):
        if v > 0:
            print(f"  {nm:20s} {v:9.0f}")
    print(f"  {'total':20s} {med.sum():9.0f}")
    res.iters[:] = 5
its = int(res.iters.sum().item())
slots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
rounds = -(-a.batch // slots)
print(f"{os.path.basename(os.environ.get('BG_LIB_PATH', 'product'))}: {a.proj} fused={a.fused} B={a.batch} steps={a.steps}: {best:.2f} ms, "
      f"{its} sample-iterations, {best * 1e3 / (its / a.batch * rounds):.2f} us per sample-iteration per workgroup, "
      f"{its / best * 1e3:.3g} sample-steps/s")
