import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "1d-burgers-equation-roms_amd"))
import numpy as np
from fem_burgers import FEMBurgers
N = 64
X = np.linspace(0, 100, N); T = np.array([np.arange(1, N), np.arange(2, N + 1)]).T
fem = FEMBurgers(X, T)
def tryit(name, f):
    try:
        r = f(); print(name, "->", getattr(r, "shape", r), getattr(r, "dtype", ""), getattr(r, "flags", None) and r.flags["C_CONTIGUOUS"])
    except Exception as e:
        print(name, "-> EXC", type(e).__name__, str(e)[:100])
u0 = np.ones(N)
tryit("list u0", lambda: fem.fom_burgers(0.05, 3, list(u0), 4.5, 0.0, 0.02))
tryit("float32 u0", lambda: fem.fom_burgers(0.05, 3, u0.astype(np.float32), 4.5, 0.0, 0.02))
tryit("int mu", lambda: fem.fom_burgers(0.05, 3, u0, 5, 0.0, 0.02))
tryit("np scalar mu", lambda: fem.fom_burgers(0.05, 3, u0, np.float64(4.5), 0.0, np.float64(0.02)))
tryit("0-d array mu", lambda: fem.fom_burgers(0.05, 3, u0, np.array(4.5), 0.0, np.array(0.02)))
tryit("mismatched mu arrays", lambda: fem.fom_burgers(0.05, 3, u0, np.array([4.5, 4.6, 4.7]), 0.0, np.array([0.02, 0.03])))
tryit("mu1 array mu2 scalar", lambda: fem.fom_burgers(0.05, 3, u0, np.array([4.5, 4.6, 4.7]), 0.0, 0.02))
tryit("wrong u0 length", lambda: fem.fom_burgers(0.05, 3, np.ones(N + 1), 4.5, 0.0, 0.02))
tryit("nTimeSteps float", lambda: fem.fom_burgers(0.05, 3.0, u0, 4.5, 0.0, 0.02))
tryit("nTimeSteps 0", lambda: fem.fom_burgers(0.05, 0, u0, 4.5, 0.0, 0.02))
tryit("negative dt", lambda: fem.fom_burgers(-0.05, 3, u0, 4.5, 0.0, 0.02))
tryit("u0 (B,N) with scalar mu", lambda: fem.fom_burgers(0.05, 3, np.ones((2, N)), 4.5, 0.0, 0.02))
Phi = np.linalg.qr(np.random.default_rng(0).standard_normal((N, 5)))[0]
tryit("pod bad projection", lambda: fem.pod_prom_burgers(0.05, 3, u0, 4.5, 0.0, 0.02, Phi, projection="galerkin"))
tryit("pod Phi wrong rows", lambda: fem.pod_prom_burgers(0.05, 3, u0, 4.5, 0.0, 0.02, Phi[:-1]))
tryit("pod Fortran Phi", lambda: fem.pod_prom_burgers(0.05, 3, u0, 4.5, 0.0, 0.02, np.asfortranarray(Phi)))
