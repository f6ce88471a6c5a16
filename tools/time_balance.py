#!/usr/bin/env python3
"""Effect of the sample order (burgers_hip/rom.py::sample_order) on the device-side time loops at the bench sizes:
kernel time with the caller's order against the balanced one."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
import bench
from burgers_hip import rom
def timed(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(2):
        e0.record(); r = f(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best, r
for cfg in sys.argv[1:] or ["quadratic", "ann", "pod_lspg", "pod_r96_lspg"]:
    a = bench.parse_args(["--config", cfg])
    w = bench.WORKLOADS[cfg](a, 0, 1, torch.device("cuda", 0))
    X, u0, m1, m2 = w.X, w.u0, w.mu1d, w.mu2d
    for bal in (False, True):
        if cfg == "quadratic":
            f = lambda: rom.quadratic_run_fused(X, u0, m1, m2, a.dt, a.time_steps, w.plan, rom.PROJ["lspg"], balance=bal)
        elif cfg == "ann":
            f = lambda: rom.pod_ann_run_fused(X, u0, m1, m2, a.dt, a.time_steps, w.Up, w.Us, w.model, rom.PROJ["lspg"], plan=w.plan, balance=bal)
        elif cfg.startswith("pod_r96"):
            f = lambda: rom.pod_prom_run_wide(X, u0, m1, m2, a.dt, a.time_steps, w.Phi, rom.PROJ[w.proj.lower()], balance=bal)
        else:
            f = lambda: rom.pod_prom_run_fused(X, u0, m1, m2, a.dt, a.time_steps, w.Phi, rom.PROJ[w.proj.lower()], balance=bal)
        ms, r = timed(f)
        its = int(r.iters.sum().item())
        print(f"{cfg} balance={bal}: {ms:.1f} ms, {its / ms * 1e3:.4g} sample-Newton-steps/s")
