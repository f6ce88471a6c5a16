#!/usr/bin/env python3
"""Per-phase shader clocks of the fused quadratic-manifold kernel (bg_quad_rom_run) from a BG_QUAD_TIMING build:
  tools/build_variant.sh qt quad_fused.hip -DBG_QUAD_TIMING [-DBG_QUAD_CHUNK=..]
  BG_LIB_PATH=1d-burgers-equation-roms_amd/build/libvar_qt.so python tools/time_quad_fused.py [--batch 1024] [--steps 40]
Bases: this framework's own training sweep, n = 40 (as bench.py --config quadratic)."""
import argparse, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--proj", default="LSPG")
a = ap.parse_args()
from burgers_hip import fom, pod, rom
N = 512
X = np.linspace(0, 100, N)
m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 3), indexing="ij")
res = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 500)
Phi, H, _ = pod.build_quadratic_manifold(pod.snapshot_matrix(res.hist).contiguous(), 40, alpha=1e-2)
dev = torch.device("cuda", torch.cuda.current_device())
plan = rom.QuadFusedPlan(Phi, H, dev)
rng = np.random.default_rng(20251121)
mu1, mu2 = rng.uniform(4.25, 5.5, a.batch), rng.uniform(0.015, 0.03, a.batch)
run = lambda: rom.quadratic_run(X, np.ones(N), mu1, mu2, 0.05, a.steps, Phi, H, projection=a.proj, plan=plan)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); r = run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
it = r.iters[:, :9].double().cpu().numpy()
names = ["pass start (Phi q)", "tangent tiles", "wait: tangent done", "decode", "assembly + projection", "wait: slab done", "reduce + solve + update", "step start / end"]
npass = np.median(it[:, 8])
print(f"{a.proj} B={a.batch} steps={a.steps}: {ms:.2f} ms; passes per workgroup (median) {npass:.0f}; kilo-clocks per pass (median over samples):")
tot = 0.0
for i, nm in enumerate(names):
    v = np.median(it[:, i]) * 1.024 / npass
    tot += v
    print(f"  {nm:28s} {v:7.2f} k")
print(f"  {'total':28s} {tot:7.2f} k   = {ms * 1e3 / npass:.1f} us per pass at {tot * 1e3 / (ms * 1e3 / npass):.0f} MHz")
