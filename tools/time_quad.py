#!/usr/bin/env python3
"""Time the quadratic-manifold pieces alone (HIP events): tangent (HIP vs addmm), reduce (frag vs row-major), decode."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import rom, lib
B, N, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 512, 40
rng = np.random.default_rng(0)
X = np.linspace(0, 100, N)
c = rom._setup(X, np.ones(N), rng.uniform(4.25, 5.5, B), rng.uniform(0.015, 0.03, B), 0.05, 0.0, None)
dev = c.device
Phi = torch.as_tensor(np.linalg.qr(rng.standard_normal((N, n)))[0], device=dev)
k = n * (n + 1) // 2
H = torch.as_tensor(1e-3 * rng.standard_normal((N, k)), device=dev)
I, J, idx, fac = rom.sym_index_tables(n, dev)
H3 = (H[:, idx] * fac).reshape(N * n, n).contiguous(); H3t = H3.t().contiguous()
q = torch.as_tensor(rng.standard_normal((B, n)), device=dev)
U = torch.as_tensor(1 + 4 * rng.random((B, N)), device=dev); G = torch.empty_like(U); rom._mass_rhs(c, U, G)
Ar = torch.zeros((B, n, n), dtype=torch.float64, device=dev); br = torch.zeros((B, n), dtype=torch.float64, device=dev)
per = int(c.L.bg_rom_frag_elems(N, n)); Wf = torch.zeros((B, per), dtype=torch.float64, device=dev)
NP = int(c.L.bg_rom_frag_pad(n)); H3p = torch.nn.functional.pad(H3.reshape(N, n, n), (0, NP - n)).contiguous(); qpad = torch.nn.functional.pad(q, (0, NP - n)).contiguous()
act = torch.ones(B, dtype=torch.int32, device=dev)
Phi_flat = Phi.reshape(1, N * n)
def t_hip(): lib.check(c.L.bg_quad_tangent(N, B, n, lib.ptr(Phi), lib.ptr(H3p), lib.ptr(qpad), lib.ptr(act), lib.ptr(Wf), c.stream()), "t")
def t_addmm(): return torch.addmm(Phi_flat, q, H3t).reshape(B, N, n)
T = t_addmm()
def r_frag(): lib.check(c.L.bg_rom_reduce_frag(N, B, n, 1, lib.ptr(c.X), lib.ptr(Wf), lib.ptr(U), lib.ptr(G), lib.ptr(c.hfs), lib.ptr(c.mu1), c.dt, c.E, 0, lib.ptr(act), lib.ptr(Ar), lib.ptr(br), None, c.stream()), "r")
def r_row(): rom.rom_reduce(c, T, U, G, 1, False, act, Ar, br, None)
HT = H.t().contiguous(); PhiT = Phi.t().contiguous()
def decode(): return q @ PhiT + (q[:, I] * q[:, J]) @ HT
for name, fn in (("tangent HIP (frag)", t_hip), ("tangent addmm", t_addmm), ("reduce frag", r_frag), ("reduce row-major", r_row), ("decode", decode)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:22s} B={B}: {e0.elapsed_time(e1)*100:8.1f} us")
# consistency: frag vs row-major reduce
t_hip(); r_frag(); torch.cuda.synchronize(); A1 = Ar.clone(); r_row(); torch.cuda.synchronize()
print("frag vs row-major Ar rel diff:", float((A1 - Ar).norm() / Ar.norm()))
