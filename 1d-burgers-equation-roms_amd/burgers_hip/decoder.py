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


class GridDecoder:
    """`predict_on_fom_grid` for batches of (mu1, mu2) with everything that does not depend on the batch kept on the
    device: the modes in the compute dtype, the standardised time column, the model.  A call is then a handful of
    launches (the uploads of the 0.65 MB mode matrix and the dtype conversion of the model on every call made the
    chunked first version of the bench 80 % host time)."""

    def __init__(self, Nt, U_modes, model, mean, std, dtype=torch.float32, device=None, fused=True):
        self.device = _lib.require_device(device)
        self.dtype, self.Nt = dtype, int(Nt)
        f64 = dict(dtype=torch.float64, device=self.device)
        mean = torch.as_tensor(np.asarray(mean, dtype=np.float64), **f64).reshape(3)
        std = torch.as_tensor(np.asarray(std, dtype=np.float64), **f64).reshape(3).clone()
        std[std == 0] = 1.0
        self.mean, self.std = mean, std
        tau = torch.linspace(0.0, 1.0, self.Nt, **f64)
        self.z_tau = ((tau - mean[2]) / std[2])                              # (Nt,)
        self.model = model.to(device=self.device, dtype=dtype).eval()
        Um = torch.as_tensor(np.asarray(U_modes), device=self.device) if not torch.is_tensor(U_modes) else U_modes.to(self.device)
        # reference: float64 modes @ float32 MLP output; bf16 tier: low-precision GEMM with fp32 accumulate
        self.Um = Um.to(torch.float64 if dtype == torch.float32 else dtype).contiguous()
        self.plan = self._fused_plan() if fused else None

    def _fused_plan(self):
        """The model as bg_decode_mlp_bf16 wants it (the MLP evaluated inside the contraction kernel), or None when that
        kernel does not apply: not the bf16 tier, not a plain Linear / ELU|ReLU|Tanh stack the recogniser of the
        intrusive path knows (rom._mlp_layers), or beyond the kernel's widths."""
        import ctypes
        import torch.nn as nn
        from . import rom
        N, n = self.Um.shape
        if self.dtype != torch.bfloat16 or self.device.type != "cuda" or N % 32 or n % 32 or n > 256:
            return None
        layers = rom._mlp_layers(self.model)
        kinds = {type(None): _lib.BG_ACT_NONE, nn.ELU: _lib.BG_ACT_ELU, nn.ReLU: _lib.BG_ACT_RELU, nn.Tanh: _lib.BG_ACT_TANH}
        if (layers is None or len(layers) > 8 or layers[0][0].in_features != 3 or layers[-1][0].out_features != n
                or any(type(act) not in kinds for _, act in layers) or max(lin.out_features for lin, _ in layers) > 256):
            return None
        with torch.no_grad():                               # the recognised chain must BE the model
            probe = torch.linspace(-1.5, 1.5, 24, dtype=torch.float32, device=self.device).reshape(8, 3).to(self.dtype)
            y = probe
            for lin, act in layers:
                y = lin(y)
                y = act(y) if act is not None else y
            if not torch.equal(y, self.model(probe)):
                return None
        bf = dict(dtype=torch.bfloat16, device=self.device)
        pad32 = lambda k: -(-k // 32) * 32
        win = [16] + [pad32(lin.out_features) for lin, _ in layers[:-1]]
        wout = [pad32(lin.out_features) for lin, _ in layers]
        Ws, bs = [], []
        for (lin, _), ki, ko in zip(layers, win, wout):
            W = torch.zeros((ko, ki), **bf)
            W[:lin.out_features, :lin.in_features] = lin.weight.detach().to(**bf)
            Ws.append(W.contiguous())
            b = torch.zeros((ko,), **bf)
            if lin.bias is not None:
                b[:lin.out_features] = lin.bias.detach().to(**bf)
            bs.append(b)
        nl = len(layers)
        return dict(keep=(Ws, bs), nl=nl, win=(ctypes.c_int * nl)(*win), wout=(ctypes.c_int * nl)(*wout),
                    W=(ctypes.c_void_p * nl)(*[w.data_ptr() for w in Ws]), bias=(ctypes.c_void_p * nl)(*[b.data_ptr() for b in bs]),
                    acts=(ctypes.c_int * nl)(*[kinds[type(act)] for _, act in layers]),
                    alphas=(ctypes.c_float * nl)(*[float(getattr(act, "alpha", 1.0)) for _, act in layers]))

    def predict(self, mu1, mu2):
        """(B, N, Nt) float64 on the device, sample b = (mu1[b], mu2[b])."""
        f64 = dict(dtype=torch.float64, device=self.device)
        mu1 = torch.as_tensor(mu1, **f64).reshape(-1)
        mu2 = torch.as_tensor(mu2, **f64).reshape(-1)
        B = max(mu1.numel(), mu2.numel())
        z1 = ((mu1 - self.mean[0]) / self.std[0]).expand(B)
        z2 = ((mu2 - self.mean[1]) / self.std[1]).expand(B)
        if self.plan is not None:
            # bf16 tier, recognised MLP: network and contraction in ONE kernel (bg_decode_mlp_bf16) -- no activation and no
            # coefficient crosses HBM; the PyTorch module below is the path for every other model
            p = self.plan
            N, n = self.Um.shape
            z1c, z2c = z1.contiguous(), z2.contiguous()
            out = torch.empty((B, N, self.Nt), dtype=torch.float64, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().bg_decode_mlp_bf16(N, n, B, self.Nt, _lib.ptr(self.Um), _lib.ptr(z1c), _lib.ptr(z2c),
                                                          _lib.ptr(self.z_tau), p["nl"], p["win"], p["wout"], p["W"], p["bias"],
                                                          p["acts"], p["alphas"], _lib.ptr(out), _lib.stream_ptr(self.device)),
                           "bg_decode_mlp_bf16")
            return out
        Zs = torch.stack([z1[:, None].expand(B, self.Nt), z2[:, None].expand(B, self.Nt),
                          self.z_tau[None, :].expand(B, self.Nt)], dim=-1).reshape(B * self.Nt, 3)
        with torch.no_grad():
            Q = self.model(Zs.to(self.dtype)).reshape(B, self.Nt, -1)                  # (B, Nt, n)
            # Uhat[b] = U_modes @ Q[b]^T as ONE batched product that lands directly in the (B, N, Nt) result layout
            # (a (N, B*Nt) product followed by permute + contiguous moves the 8-byte result twice more)
            if self.dtype == torch.float32:
                return torch.matmul(self.Um, Q.to(torch.float64).transpose(1, 2))
            N, n = self.Um.shape
            if self.dtype == torch.bfloat16 and N % 32 == 0 and n % 16 == 0 and n <= 256:
                # bf16 tier: the product is write-bound; bg_decode_modes_bf16 writes the float64 result once instead of a
                # bf16 GEMM result plus a cast pass over it
                Qc = Q.reshape(B * self.Nt, n).contiguous()
                out = torch.empty((B, N, self.Nt), dtype=torch.float64, device=self.device)
                with torch.cuda.device(self.device):
                    _lib.check(_lib.load().bg_decode_modes_bf16(N, n, B, self.Nt, _lib.ptr(self.Um), _lib.ptr(Qc), _lib.ptr(out),
                                                                _lib.stream_ptr(self.device)), "bg_decode_modes_bf16")
                return out
            return torch.matmul(self.Um, Q.transpose(1, 2)).to(torch.float64)


def predict_on_grid(mu1, mu2, Nt, U_modes, model, mean, std, dtype=torch.float32, device=None, fused=True):
    """Batched decoder: returns (B, N, Nt) float64 on the device, sample b = (mu1[b], mu2[b])."""
    mu1 = np.atleast_1d(np.asarray(mu1, dtype=np.float64)) if not torch.is_tensor(mu1) else mu1
    mu2 = np.atleast_1d(np.asarray(mu2, dtype=np.float64)) if not torch.is_tensor(mu2) else mu2
    return GridDecoder(Nt, U_modes, model, mean, std, dtype=dtype, device=device, fused=fused).predict(mu1, mu2)
