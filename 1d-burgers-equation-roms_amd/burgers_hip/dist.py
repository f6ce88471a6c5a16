"""Multi-GPU mu-sweep: one process per GPU, samples sharded by rank, no collective in the
time loop.  The only exchange is the all-gather of per-rank snapshot blocks that feeds the
offline SVD (the reference's POD/pod.py:80-84 stacks the same blocks from .npy files).

backend "nccl" is RCCL over xGMI on ROCm; "gloo" runs the same code on CPU tensors (tests).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of `total` samples owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t, rank, world):
    lo, hi = shard_bounds(len(t), rank, world)
    return t[lo:hi]


def all_gather_blocks(local, total, group=None):
    """Gather per-rank blocks local[(hi-lo), ...] into the full [total, ...] tensor on every rank.

    Ranks pad their block to the largest shard so that one all_gather_into_tensor moves
    everything (one large collective; xGMI rings are per-link bound, so fewer, larger is better).
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (total + world - 1) // world
    tail = tuple(local.shape[1:])
    buf = local
    if local.shape[0] != per:
        buf = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    buf = buf.contiguous()
    out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, buf, group=group)
    except (RuntimeError, NotImplementedError):           # backends without the fused form
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        out = torch.cat(parts, dim=0)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        pieces.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(pieces, dim=0) if any(p.shape[0] != per for p in pieces) else out[:total]


def sweep(runner, mu1_all, mu2_all, rank, world):
    """Run `runner(mu1_shard, mu2_shard)` on this rank's contiguous block of the sweep."""
    lo, hi = shard_bounds(len(mu1_all), rank, world)
    return runner(mu1_all[lo:hi], mu2_all[lo:hi])


def max_over_ranks(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
