"""Non-intrusive POD-ANN decoder (BASELINE config 5, decoder-only variant).

reference: Non-Instrusive/predict_pod_ann.py:73-80  `predict_on_fom_grid`:
    Z = [mu1, mu2, tau],  tau = linspace(0, 1, Nt);  Zs = (Z - mean) / std
    Qhat = MLP(float32(Zs));  Uhat = U_modes @ Qhat.T
No Newton loop: this is one MLP evaluation and one dense contraction over the whole batch,
i.e. plain PyTorch-ROCm GEMMs (fp32 like the reference, or bf16 weights/activations with fp32
accumulate for the throughput tier of config 5).
"""
from __future__ import annotations

import numpy as np
import torch

from . import lib as _lib


def standardize(Z, mean, std):
    std = std.clone()
    std[std == 0] = 1.0
    return (Z - mean) / std


def predict_on_grid(mu1, mu2, Nt, U_modes, model, mean, std, dtype=torch.float32, device=None):
    """Batched decoder: returns (B, N, Nt) float64 on the device, sample b = (mu1[b], mu2[b])."""
    device = _lib.require_device(device)
    mu1 = torch.as_tensor(np.atleast_1d(np.asarray(mu1, dtype=np.float64)), device=device)
    mu2 = torch.as_tensor(np.atleast_1d(np.asarray(mu2, dtype=np.float64)), device=device)
    B = max(mu1.numel(), mu2.numel())
    mu1, mu2 = mu1.expand(B), mu2.expand(B)
    tau = torch.linspace(0.0, 1.0, Nt, dtype=torch.float64, device=device)
    Z = torch.stack([mu1[:, None].expand(B, Nt), mu2[:, None].expand(B, Nt), tau[None, :].expand(B, Nt)], dim=-1)
    mean = torch.as_tensor(np.asarray(mean, dtype=np.float64), device=device).reshape(1, 1, 3)
    std = torch.as_tensor(np.asarray(std, dtype=np.float64), device=device).reshape(1, 1, 3)
    Zs = standardize(Z, mean, std).reshape(B * Nt, 3)
    model = model.to(device=device, dtype=dtype)
    Um = torch.as_tensor(np.asarray(U_modes), device=device)
    with torch.no_grad():
        Q = model(Zs.to(dtype)).reshape(B, Nt, -1)                 # (B, Nt, n)
        # Uhat[b] = U_modes @ Q[b]^T as ONE batched product that lands directly in the (B, N, Nt) result layout
        # (a (N, B*Nt) product followed by permute + contiguous moves the 8-byte result twice more)
        if dtype == torch.float32:
            U = torch.matmul(Um.to(torch.float64), Q.to(torch.float64).transpose(1, 2))     # reference: float64 modes @ float32 output
        else:
            U = torch.matmul(Um.to(dtype), Q.transpose(1, 2)).to(torch.float64)             # bf16 tier: low-precision GEMM, fp32 accumulate
    return U
