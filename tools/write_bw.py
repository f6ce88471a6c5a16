#!/usr/bin/env python3
"""What this box writes to HBM at best: torch fill_ / copy_ of a 2.1 GB float64 tensor (the decoder's result size)."""
import torch
dev = torch.device("cuda", 0)
x = torch.empty((1024, 512, 501), dtype=torch.float64, device=dev)
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
b = x.numel() * 8
ms = t(lambda: x.fill_(1.5)); print(f"fill_  2.1 GB: {ms*1e3:.0f} us = {b/ms/1e9:.2f} TB/s written")
ms = t(lambda: x.zero_()); print(f"zero_  2.1 GB: {ms*1e3:.0f} us = {b/ms/1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy_  2.1 GB: {ms*1e3:.0f} us = {b/ms/1e9:.2f} TB/s written (+ as much read)")
