#!/usr/bin/env python3
"""How much the four-samples-per-workgroup lockstep of bg_quad_rom_run costs on the bench workload: passes per step of a
workgroup = max over its four samples; compare the sum of those maxima with the mean, for the bench order and for
samples grouped by mu1 / by their own iteration totals (the best any grouping could do)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
import bench
a = bench.parse_args(["--config", "quadratic", "--time-steps", "100"])
w = bench.WORKLOADS["quadratic"](a, 0, 1, torch.device("cuda", 0))
res = w.one_pass(); torch.cuda.synchronize()
it = res.iters.cpu().numpy().astype(np.int64)             # (B, steps)
B = it.shape[0]
def cost(order):
    g = it[order].reshape(B // 4, 4, -1)
    return g.max(axis=1).sum() * 4 / it.sum()
print("iterations per step: mean %.3f, min %d, max %d" % (it.mean(), it.min(), it.max()))
print("lockstep passes / sample-iterations: bench order %.4f, grouped by mu1 %.4f, by mu2 %.4f, by own totals %.4f"
      % (cost(np.arange(B)), cost(np.argsort(w.mu1)), cost(np.argsort(w.mu2)), cost(np.argsort(it.sum(1)))))
tot = it.sum(1)
print("per-sample totals: mean %.1f, std %.1f, max/mean %.3f; corr with mu1 %.2f, mu2 %.2f" % (tot.mean(), tot.std(), tot.max() / tot.mean(),
      np.corrcoef(tot, w.mu1)[0, 1], np.corrcoef(tot, w.mu2)[0, 1]))
