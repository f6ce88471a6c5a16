#!/usr/bin/env python3
"""Time bg_rom_reduce / bg_rom_reduce_lifted / bg_lu_solve_update alone (HIP events), B samples."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "1d-burgers-equation-roms_amd"))
import numpy as np, torch
from burgers_hip import rom, lib

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N, r = 512, int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(0)
    X = np.linspace(0, 100, N)
    mu1 = rng.uniform(4.25, 5.5, B); mu2 = rng.uniform(0.015, 0.03, B)
    U = 1 + 4 * rng.random((B, N))
    Phi = np.linalg.qr(rng.standard_normal((N, r)))[0]
    c = rom._setup(X, U, mu1, mu2, 0.05, 0.0, None)
    dev = c.device
    Ud = torch.as_tensor(U, device=dev); G = torch.empty_like(Ud); rom._mass_rhs(c, Ud, G)
    Phid = torch.as_tensor(Phi, device=dev)
    W = torch.as_tensor(rng.standard_normal((B, N, r)), device=dev) if "--persample" in sys.argv else None
    Ar = torch.zeros((B, r, r), dtype=torch.float64, device=dev); br = torch.zeros((B, r), dtype=torch.float64, device=dev)
    wtu = torch.zeros((B, r), dtype=torch.float64, device=dev); q = torch.as_tensor(U @ Phi, device=dev)
    for proj, name in ((0, "galerkin"), (1, "lspg")):
        for mode in ("U", "lifted") + (("persample",) if W is not None else ()):
            def call():
                if mode == "U": rom.rom_reduce(c, Phid, Ud, G, proj, True, None, Ar, br, wtu)
                elif mode == "lifted": rom.rom_reduce_lifted(c, Phid, q, Ud, G, proj, True, None, Ar, br, wtu)
                else: rom.rom_reduce(c, W, Ud, G, proj, True, None, Ar, br, None)
            for _ in range(3): call()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"reduce {name:9s} {mode:9s} B={B}: {us:8.1f} us/launch  {us/ (B/256):6.2f} us per sample per WG")
    st = rom._IterState(c, r)
    for _ in range(3):
        st.active.fill_(1); st.k.zero_(); st.launched = 0
        st.solve_update(1, Ar, br, wtu, q, 1e-6, 20)
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        st.active.fill_(1); st.k.zero_(); st.launched = 0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); st.solve_update(1, Ar, br, wtu, q, 1e-6, 20); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e3)
    print(f"lu_solve_update n={r} B={B}: {min(t):.1f} us")
main()
