#!/usr/bin/env python3
"""bg_decode_mlp_bf16 (MLP inside the contraction kernel) against the PyTorch bf16 module + bg_decode_modes_bf16, and both
against the float32 tier: relative L2 and the share of result entries that are bitwise equal; kernel-only times."""
import copy, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "1d-burgers-equation-roms_amd")]
import numpy as np, torch
import bench
from burgers_hip import decoder
g = bench.golden("nonintrusive_decoder.npz")
model = bench.decoder_model(g)
rng = np.random.default_rng(5)
B = 1024
mu1, mu2 = rng.uniform(4.25, 5.5, B), rng.uniform(0.015, 0.03, B)
Nt = 501
mk = lambda **kw: decoder.GridDecoder(Nt, g["U_modes"], copy.deepcopy(model), g["mean"], g["std"], **kw)
fused, plain, f32 = mk(dtype=torch.bfloat16), mk(dtype=torch.bfloat16, fused=False), mk()
U, V = fused.predict(mu1[:64], mu2[:64]), plain.predict(mu1[:64], mu2[:64])
F = f32.predict(mu1[:64], mu2[:64])
rl = lambda a, b: float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
print("fused vs module: rel-L2 %.3e, bitwise-equal entries %.4f; vs float32 tier: fused %.3e, module %.3e"
      % (rl(U, V), float((U == V).double().mean()), rl(U, F), rl(V, F)))
m1, m2 = torch.as_tensor(mu1, device="cuda"), torch.as_tensor(mu2, device="cuda")
for name, d in (("bg_decode_mlp_bf16", fused), ("module + bg_decode_modes_bf16", plain)):
    d.predict(m1, m2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = d.predict(m1, m2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {ms:.3f} ms per {B} samples = {B * Nt / ms * 1e3:.3g} columns/s, result stream {B * Nt * 512 * 8 / ms / 1e9:.2f} TB/s")
