#!/usr/bin/env python3
"""bench_rom.py -- secondary benchmarks: the projection-ROM configs of BASELINE.json
(configs[2..4]) on one GPU.  The driver's contract lives in bench.py (FOM, configs[1]); this
script reports the same unit (sample-Newton-steps/s) for the ROM families, priced against
the fp64-MFMA roofline with SURVEY.md section 8d's algorithmic flop counts.

  python bench_rom.py [--which pod_galerkin pod_lspg quadratic ann] [--time-steps 50]

The bases are rebuilt from this framework's own snapshots (FOM sweep -> SVD), as
BASELINE.json's config notes ask; data are synthetic (mu ~ U[4.25,5.5] x U[0.015,0.03]).
One JSON line per workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "1d-burgers-equation-roms_amd")
for _p in (REPO, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

FP64_MFMA_PEAK_TF = 78.6      # MI355X fp64 peak as priced by SURVEY 8d (vector = matrix figure)
FP64_MFMA_MEASURED_TF = 33.3  # bare v_mfma_f64_16x16x4_f64 loop, every SIMD busy (tools/mfma_bench.hip): 128 cyc/MFMA
SEED = 20251121


def build_bases(N, n_pod, n_quad):
    """Training sweep (3x3 grid of FEM/paper_training_stage.py:8-10) -> snapshots -> SVD."""
    from burgers_hip import fom, pod
    X = np.linspace(0.0, 100.0, N)
    m1, m2 = np.meshgrid(np.linspace(4.25, 5.5, 3), np.linspace(0.015, 0.03, 3), indexing="ij")
    res = fom.fom_run(X, np.ones(N), m1.ravel(), m2.ravel(), 0.05, 500)
    S = pod.snapshot_matrix(res.hist).contiguous()
    Phi, s, _ = pod.pod_basis(S, n_modes=n_pod)
    PhiQ, H, _ = pod.build_quadratic_manifold(S, n_quad, alpha=1e-2)
    return X, Phi, PhiQ, H


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fn()
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", nargs="*", default=["pod_galerkin", "pod_lspg", "quadratic", "ann"],
                    help="also available: rbf (POD-RBF closure of tests/golden/rbf_n17.npz, SURVEY 8f.3), "
                         "local (local POD bases of tests/golden/local_pod.npz, SURVEY 8f.2)")
    ap.add_argument("--time-steps", type=int, default=40)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--r", type=int, default=40)
    ap.add_argument("--batch-pod", type=int, default=4096)
    ap.add_argument("--batch-quad", type=int, default=1024)
    ap.add_argument("--batch-ann", type=int, default=2048)
    args = ap.parse_args()
    assert torch.cuda.is_available()
    from burgers_hip import rom, lib
    lib.load()
    N, r, nT, dt = args.n, args.r, args.time_steps, 0.05
    X, Phi, PhiQ, H = build_bases(N, r, r)
    rng = np.random.default_rng(SEED)

    def mus(B):
        return rng.uniform(4.25, 5.5, B), rng.uniform(0.015, 0.03, B)

    def report(name, B, res, secs, flops_per_step, extra=None):
        steps = int(res.iters.sum().item())
        val = steps / secs
        tf = val * flops_per_step / 1e12
        line = {"metric": "sample-Newton-steps/s", "workload": name, "value": val, "unit": "sample-Newton-steps/s",
                "batch": B, "N": N, "r": r, "time_steps": nT, "newton_steps": steps, "seconds": secs,
                "iters_per_step": steps / (B * nT), "dtype": "f64", "data": "synthetic",
                "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": tf / FP64_MFMA_PEAK_TF, "algorithmic_flops_per_step": flops_per_step,
                             "measured_mfma_ceiling": FP64_MFMA_MEASURED_TF, "frac_of_measured_ceiling": tf / FP64_MFMA_MEASURED_TF},
                "nonfinite": int((res.flags & 2).ne(0).sum().item()), "hit_cap": int((res.flags & 1).ne(0).sum().item())}
        if extra:
            line.update(extra)
        print(json.dumps(line), flush=True)

    pod_flops = 2 * N * r * r + 11 * N * r + (2 * r ** 3) / 3
    k = r * (r + 1) // 2
    quad_flops = 2 * N * (r + k) + 4 * N * k + 2 * N * r * r + 11 * N * r + (2 * r ** 3) / 3
    for which in args.which:
        if which in ("pod_galerkin", "pod_lspg"):
            proj = "Galerkin" if which == "pod_galerkin" else "LSPG"
            B = args.batch_pod
            m1, m2 = mus(B)
            rom.pod_prom_run(X, np.ones(N), m1[:64], m2[:64], dt, 2, Phi, projection=proj)   # warm-up
            res, secs = timed(lambda: rom.pod_prom_run(X, np.ones(N), m1, m2, dt, nT, Phi, projection=proj))
            report(f"configs[2]: POD-{proj} r={r}, {B} samples, N={N}", B, res, secs, pod_flops)
        elif which == "quadratic":
            B = args.batch_quad
            m1, m2 = mus(B)
            rom.quadratic_run(X, np.ones(N), m1[:32], m2[:32], dt, 2, PhiQ, H)
            res, secs = timed(lambda: rom.quadratic_run(X, np.ones(N), m1, m2, dt, nT, PhiQ, H, projection="LSPG"))
            report(f"configs[3]: quadratic-manifold LSPG n={r} (k={k}), {B} samples/GPU, N={N}", B, res, secs, quad_flops)
        elif which == "ann":
            g = np.load(os.path.join(REPO, "tests", "golden", "ann_n5.npz"))
            import torch.nn as nn
            dims = [5, 32, 64, 128, 256, 256, 91]
            layers = []
            for i in range(6):
                lin = nn.Linear(dims[i], dims[i + 1])
                with torch.no_grad():
                    lin.weight.copy_(torch.from_numpy(g[f"W{i}"])); lin.bias.copy_(torch.from_numpy(g[f"b{i}"]))
                layers.append(lin)
                if i < 5:
                    layers.append(nn.ELU())
            model = nn.Sequential(*layers).eval()
            B = args.batch_ann
            m1, m2 = mus(B)
            rom.pod_ann_run(X, np.ones(N), m1[:32], m2[:32], dt, 2, g["U_p"], g["U_s"], model)
            res, secs = timed(lambda: rom.pod_ann_run(X, np.ones(N), m1, m2, dt, nT, g["U_p"], g["U_s"], model))
            n, nb = 5, 91
            ann_flops = 2 * 132000 * (1 + n) + 2 * N * (n + nb) * (1 + n) + 2 * N * n * n + 11 * N * n
            report(f"configs[4]: POD-ANN n={n}, nbar={nb} (fp32 MLP), {B} samples/GPU, N={N}", B, res, secs, ann_flops,
                   {"r": n})
        elif which == "local":
            g = np.load(os.path.join(REPO, "tests", "golden", "local_pod.npz"))
            B = args.batch_pod
            m1, m2 = mus(B)
            bases = {c: g[f"basis{c}"] for c in range(4)}
            cl = (g["centers"], bases, g["U_global"], 12)
            rom.local_prom_run(X, np.ones(N), m1[:32], m2[:32], dt, 2, *cl, projection="LSPG")
            res, secs = timed(lambda: rom.local_prom_run(X, np.ones(N), m1, m2, dt, nT, *cl, projection="LSPG"))
            rl = max(b.shape[1] for b in bases.values())
            report(f"local POD-LSPG, 4 clusters, widths 14..30, {B} samples/GPU, N={N}", B, res, secs,
                   2 * N * rl * rl + 11 * N * rl + (2 * rl ** 3) / 3, {"r": rl})
        elif which == "rbf":
            g = np.load(os.path.join(REPO, "tests", "golden", "rbf_n17.npz"))
            B = args.batch_ann
            m1, m2 = mus(B)
            cl = (g["U_p"], g["U_s"], g["X_train"], g["W_gaussian"], float(g["eps_gaussian"]), g["x_min"], g["x_max"],
                  g["y_min"], g["y_max"])
            rom.pod_rbf_run(X, np.ones(N), m1[:32], m2[:32], dt, 2, *cl)
            res, secs = timed(lambda: rom.pod_rbf_run(X, np.ones(N), m1, m2, dt, nT, *cl))
            n, nb, ns = g["U_p"].shape[1], g["U_s"].shape[1], g["X_train"].shape[0]
            rbf_flops = 2 * ns * (n + nb) * (1 + n) + 2 * N * (n + nb) * (1 + n) + 2 * N * n * n + 11 * N * n
            report(f"POD-RBF (gaussian, {ns} centres) n={n}, nbar={nb}, {B} samples/GPU, N={N}", B, res, secs, rbf_flops, {"r": n})


if __name__ == "__main__":
    main()
