"""Batched projection-ROM time-steppers (host orchestration over the C ABI).

Each Picard / Gauss-Newton iteration of the whole batch is two HIP launches,
    bg_rom_reduce[_lifted]  (assembly + fp64-MFMA projection, for POD also the lift u = Phi q)
    bg_lu_solve_update      (pivoted r x r solve + reduced-coordinate update + convergence bookkeeping)
plus the family-specific decode / tangent, which are plain dense contractions over the batch
(torch / rocBLAS).  Samples carry an ``active`` flag on the device; converged samples are
skipped by the kernels, so every sample follows exactly the reference's own loop:
  pod_prom_burgers        FEM/fem_burgers.py:709-785
  pod_quadratic_manifold  FEM/fem_burgers.py:1081-1175
  pod_ann_prom            FEM/fem_burgers.py:1177-1251
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import lib as _lib
from .fom import FomResult, _as_dev, check_mesh
from .fom import batch_inputs as _batch_inputs

PROJ = {"galerkin": _lib.BG_PROJ_GALERKIN, "lspg": _lib.BG_PROJ_LSPG}


class SingularReducedSystem(np.linalg.LinAlgError):
    """np.linalg.solve raises LinAlgError('Singular matrix') at the same place (:767)."""


@dataclass
class _Common:
    L: object
    device: torch.device
    X: torch.Tensor
    N: int
    B: int
    mu1: torch.Tensor
    mu2: torch.Tensor
    u0: torch.Tensor
    fdt: torch.Tensor
    hfs: torch.Tensor
    dt: float
    E: float
    mesh_opt: int = 0
    Un: torch.Tensor = None          # state at the start of the time step (library path only)

    def stream(self):
        return _lib.stream_ptr(self.device)


def _setup(X, u0, mu1, mu2, dt, E, device, max_n=None):
    L = _lib.load()
    device = _lib.require_device(device)
    mesh_opt = _lib.mesh_options(check_mesh(X), supg=False)
    Xd = _as_dev(X, device)
    N = Xd.numel()
    max_n = L.bg_fom_max_n() if max_n is None else max_n
    if N > max_n:
        raise NotImplementedError(f"ROM steppers cover N <= {max_n} (got {N})")
    u0d, mu1d, mu2d = _batch_inputs(u0, mu1, mu2, N, device)
    B = mu1d.numel()
    fdt = torch.empty((B, N), dtype=torch.float64, device=device)
    hfs = torch.empty((B, N), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        _lib.check(L.bg_forcing_setup(N, B, _lib.ptr(Xd), _lib.ptr(mu2d), float(dt), mesh_opt, _lib.ptr(fdt),
                                      _lib.ptr(hfs), _lib.stream_ptr(device)), "bg_forcing_setup")
    return _Common(L, device, Xd, N, B, mu1d, mu2d, u0d, fdt, hfs, float(dt), float(E), mesh_opt)


def _mass_rhs(c, Un, out):
    c.Un = Un.clone()
    with torch.cuda.device(c.device):
        _lib.check(c.L.bg_mass_rhs(c.N, c.B, _lib.ptr(c.X), _lib.ptr(Un), _lib.ptr(c.fdt), c.mesh_opt, _lib.ptr(out),
                                   c.stream()), "bg_mass_rhs")
    return out


def rom_reduce(c, W, U, G, proj, supg, active, Ar, br, wtu=None, colmajor=False, w_index=None, extra_opts=0):
    """Ar, br (and optionally W^T u) of every active sample.  W: (N, r) shared or (B, N, r);
    with ``colmajor`` the transposed blocks (r, N) / (B, r, N); with ``w_index`` (B,) int32 a stack
    (C, N, r) of bases of which sample b uses block w_index[b]."""
    r = W.shape[-2] if colmajor else W.shape[-1]
    if r > c.L.bg_rom_max_r() or c.N > c.L.bg_rom_max_n():
        Wl = W if w_index is None else W[w_index.long()]
        return _rom_reduce_library(c, Wl.transpose(-1, -2) if colmajor else Wl, U, proj, supg, active, Ar, br, wtu)
    stride = 0 if W.dim() == 2 else c.N * r
    opts = (1 if supg else 0) | c.mesh_opt | (_lib.BG_OPT_W_COLMAJOR if colmajor else 0) | extra_opts
    if w_index is not None:
        with torch.cuda.device(c.device):
            rc = c.L.bg_rom_reduce_indexed(c.N, c.B, r, proj, _lib.ptr(c.X), _lib.ptr(W), stride, _lib.ptr(w_index),
                                           _lib.ptr(U), _lib.ptr(G), _lib.ptr(c.hfs), _lib.ptr(c.mu1), c.dt, c.E, opts,
                                           _lib.ptr(active) if active is not None else None, _lib.ptr(Ar), _lib.ptr(br),
                                           _lib.ptr(wtu) if wtu is not None else None, c.stream())
        _lib.check(rc, "bg_rom_reduce_indexed")
        return
    with torch.cuda.device(c.device):
        rc = c.L.bg_rom_reduce(c.N, c.B, r, proj, _lib.ptr(c.X), _lib.ptr(W), stride, _lib.ptr(U), _lib.ptr(G),
                               _lib.ptr(c.hfs), _lib.ptr(c.mu1), c.dt, c.E, opts,
                               _lib.ptr(active) if active is not None else None, _lib.ptr(Ar), _lib.ptr(br),
                               _lib.ptr(wtu) if wtu is not None else None, c.stream())
    if rc == _lib.BG_ERR_UNSUPPORTED_R:
        raise NotImplementedError(f"ROM kernels cover r <= {c.L.bg_rom_max_r()} (got {r})")
    _lib.check(rc, "bg_rom_reduce")


def _rom_reduce_library(c, W, U, proj, supg, active, Ar, br, wtu):
    """Same outputs as bg_rom_reduce for sizes beyond the register-resident MFMA kernels
    (r > 47 or N > 512): HIP assembly (bg_fom_assemble), then A W, the projection and W^T u as
    library GEMMs over the batch.  Inactive samples keep their previous outputs."""
    from . import fom as _fom
    lo, di, up, rhs = _fom.fom_assemble(c.X.cpu().numpy(), U, c.Un, c.mu1, c.mu2, c.dt, E=c.E, supg=supg,
                                        device=c.device)
    Wb = W if W.dim() == 3 else W.unsqueeze(0)
    z = torch.zeros_like(Wb[:, :1])
    Y = di.unsqueeze(-1) * Wb + lo.unsqueeze(-1) * torch.cat([z, Wb[:, :-1]], 1) + \
        up.unsqueeze(-1) * torch.cat([Wb[:, 1:], z], 1)                                   # A W, (B, N, r)
    L_ = Wb if proj == _lib.BG_PROJ_GALERKIN else Y
    Ar_new = torch.matmul(L_.transpose(1, 2), Y)
    br_new = -torch.matmul(L_.transpose(1, 2), rhs.unsqueeze(-1)).squeeze(-1)             # R = -rhs
    m = torch.ones((c.B,), dtype=torch.bool, device=c.device) if active is None else active.bool()
    Ar.copy_(torch.where(m[:, None, None], Ar_new, Ar))
    br.copy_(torch.where(m[:, None], br_new, br))
    if wtu is not None:
        wtu.copy_(torch.where(m[:, None], torch.matmul(Wb.transpose(1, 2), U.unsqueeze(-1)).squeeze(-1), wtu))


def lu_solve(A, b, sign=1.0, active=None, x=None, info=None):
    """x = solve(A, sign*b) for a batch of (n, n) systems on the device."""
    L = _lib.load()
    device = _lib.require_device(A.device)
    B, n, _ = A.shape
    x = torch.empty((B, n), dtype=torch.float64, device=device) if x is None else x
    info = torch.zeros((B,), dtype=torch.int32, device=device) if info is None else info
    with torch.cuda.device(device):
        rc = L.bg_lu_solve(n, B, _lib.ptr(A), _lib.ptr(b), float(sign),
                           _lib.ptr(active) if active is not None else None, _lib.ptr(x), _lib.ptr(info),
                           _lib.stream_ptr(device))
    if rc == _lib.BG_ERR_UNSUPPORTED_R:
        raise NotImplementedError("bg_lu_solve covers n <= 64")
    _lib.check(rc, "bg_lu_solve")
    return x, info


def rom_reduce_lifted(c, Phi, q, U, G, proj, supg, active, Ar, br, wtu, extra_opts=0):
    """bg_rom_reduce with u = Phi q formed in-kernel (and stored to U)."""
    r = Phi.shape[1]
    with torch.cuda.device(c.device):
        rc = c.L.bg_rom_reduce_lifted(c.N, c.B, r, proj, _lib.ptr(c.X), _lib.ptr(Phi), _lib.ptr(q), _lib.ptr(U),
                                      _lib.ptr(G), _lib.ptr(c.hfs), _lib.ptr(c.mu1), c.dt, c.E,
                                      (1 if supg else 0) | c.mesh_opt | extra_opts, _lib.ptr(active) if active is not None else None, _lib.ptr(Ar), _lib.ptr(br),
                                      _lib.ptr(wtu) if wtu is not None else None, c.stream())
    if rc == _lib.BG_ERR_UNSUPPORTED_R:
        raise NotImplementedError(f"ROM kernels cover r <= {c.L.bg_rom_max_r()} (got {r})")
    _lib.check(rc, "bg_rom_reduce_lifted")


def rom_lift(c, Phi, q, U, active=None):
    with torch.cuda.device(c.device):
        rc = c.L.bg_rom_lift(c.N, c.B, Phi.shape[1], _lib.ptr(c.X), _lib.ptr(Phi), _lib.ptr(q),
                             _lib.ptr(active) if active is not None else None, _lib.ptr(U), c.stream())
    _lib.check(rc, "bg_rom_lift")


class _IterState:
    """Per-time-step bookkeeping that lives on the device; one 8-byte readback per iteration."""

    def __init__(self, c, n):
        i32 = dict(dtype=torch.int32, device=c.device)
        self.c = c
        self.active = torch.ones((c.B,), **i32)
        self.k = torch.zeros((c.B,), **i32)
        self.flags = torch.zeros((c.B,), **i32)
        self.info = torch.zeros((c.B,), **i32)
        self.counter = torch.zeros((2, _lib.BG_COUNTER_SLOTS * _lib.BG_COUNTER_STRIDE), **i32)   # partial counts: still active | singular
        self.dq = torch.zeros((c.B, n), dtype=torch.float64, device=c.device)

    # Converged samples are skipped ON THE DEVICE (active mask), so the host does not have to
    # look after every batched iteration: it polls after POLL_FIRST iterations of a time step
    # and then every POLL_EVERY; an iteration launched on an all-converged batch is a no-op.
    POLL_FIRST = 4
    POLL_EVERY = 2

    def begin_step(self):
        self.active.fill_(1)
        self.k.zero_()
        self.launched = 0

    def _solve_update_library(self, mode, Ar, br, wtu, q, tol, max_it):
        """bg_lu_solve_update for n > 64: rocSOLVER LU through torch, same update rules."""
        act = self.active.bool()
        dq = _batched_solve(Ar, -br, act)
        qn = (wtu if mode == 1 else q) + dq
        q.copy_(torch.where(act[:, None], qn, q))
        nd, nq = torch.linalg.vector_norm(dq, dim=1), torch.linalg.vector_norm(qn, dim=1)
        self.k += self.active
        if mode == 1:
            err = nd / nq; more = (err > tol) & (self.k < max_it)
        elif mode == 2:
            err = nd / torch.clamp(nq, min=1e-14); more = ~(err < tol) & (self.k < max_it)
        else:
            err = nd / (nq + 1e-14); more = (err > tol) & (self.k < max_it)
        capped = (self.k >= max_it) & (~(err < tol) if mode == 2 else torch.ones_like(more))
        self.flags |= (act & ~torch.isfinite(err)).to(torch.int32) * _lib.BG_FLAG_NONFINITE
        self.flags |= (act & capped).to(torch.int32) * _lib.BG_FLAG_HIT_CAP
        self.active.copy_((act & more).to(torch.int32))
        return int(self.active.sum().item())

    def solve_update(self, mode, Ar, br, wtu, q, tol, max_it):
        """dq = solve(Ar, -br); q, iteration counters and the active mask updated on the device.
        Returns the number of samples that need another iteration (-1: not polled this time)."""
        c = self.c
        self.launched += 1
        self.active_before = self.active.bool()        # samples whose q this call updates
        if q.shape[1] > 64:
            return self._solve_update_library(mode, Ar, br, wtu, q, tol, max_it)
        poll = (self.launched >= max_it or self.launched == self.POLL_FIRST or
                (self.launched > self.POLL_FIRST and (self.launched - self.POLL_FIRST) % self.POLL_EVERY == 0))
        if poll:
            self.counter[0].zero_()         # row 1 (singular systems) keeps accumulating until read
        with torch.cuda.device(c.device):
            rc = c.L.bg_lu_solve_update(q.shape[1], c.B, _lib.ptr(Ar), _lib.ptr(br), mode,
                                        _lib.ptr(wtu) if wtu is not None else None, _lib.ptr(q), _lib.ptr(self.dq),
                                        float(tol), int(max_it), _lib.ptr(self.active), _lib.ptr(self.k),
                                        _lib.ptr(self.flags), _lib.ptr(self.counter), _lib.ptr(self.info), c.stream())
        if rc == _lib.BG_ERR_UNSUPPORTED_R:
            raise NotImplementedError("bg_lu_solve covers n <= 64")
        _lib.check(rc, "bg_lu_solve_update")
        if not poll:
            return -1
        n_active, n_singular = self.counter.cpu().sum(1).tolist()      # 4 KB readback, summed on the host
        if n_singular:
            raise SingularReducedSystem("Singular matrix")
        return n_active


def _alloc_hist(c, nsteps):
    hist = torch.empty((c.B, nsteps + 1, c.N), dtype=torch.float64, device=c.device)
    hist[:, 0] = c.u0
    iters = torch.zeros((c.B, nsteps), dtype=torch.int32, device=c.device)
    flags = torch.zeros((c.B,), dtype=torch.int32, device=c.device)
    return hist, iters, flags


def _workspace(c, r):
    """Reduced system Ar | br, W^T u and the per-step right-hand side G = M u^n + dt F."""
    f64 = dict(dtype=torch.float64, device=c.device)
    return (torch.zeros((c.B, r, r), **f64), torch.zeros((c.B, r), **f64), torch.zeros((c.B, r), **f64),
            torch.empty((c.B, c.N), **f64))


# --------------------------------------------------------------------------- POD
def pod_prom_run_fused(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E=0.0, tol=1e-6, max_it=20, device=None, options=0, balance=True):
    """``pod_prom_burgers`` for a batch with the whole time loop on the device (bg_rom_run): one workgroup per
    sample, no host in the loop.  Covers N <= 512 and r <= bg_rom_run_max_r()."""
    L = _lib.load()
    device = _lib.require_device(device)
    opts = _lib.mesh_options(check_mesh(X), supg=True) | options
    Xd = _as_dev(X, device)
    N = Xd.numel()
    Phid = _as_dev(Phi, device)
    if Phid.dim() != 2 or Phid.shape[0] != N:
        raise ValueError("Phi must have one row per mesh node")
    r = Phid.shape[1]
    u0d, mu1d, mu2d = _batch_inputs(u0, mu1, mu2, N, device)
    B = mu1d.numel()
    hist = torch.empty((B, nsteps + 1, N), dtype=torch.float64, device=device)
    iters = torch.zeros((B, nsteps), dtype=torch.int32, device=device)
    flags = torch.zeros((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    order = sample_order(mu1d, 2 * _cu_count(device)) if balance else None
    with torch.cuda.device(device):
        rc = L.bg_rom_run(N, B, r, int(nsteps), proj, _lib.ptr(Xd), _lib.ptr(Phid), _lib.ptr(u0d), _lib.ptr(mu1d),
                          _lib.ptr(mu2d), float(dt), float(E), float(tol), int(max_it), int(opts), _lib.ptr(hist),
                          _lib.ptr(iters), _lib.ptr(flags), _lib.ptr(info), _lib.ptr(order), _lib.stream_ptr(device))
    _lib.check(rc, "bg_rom_run")
    res = FomResult(hist, iters, flags)
    res.info = info              # checked lazily by the facade (a readback would synchronise)
    return res


def sample_order(mu1d, grid, group=1):
    """The ``order`` argument of the device-side time loops (bg_rom_run, bg_rom_run_wide, bg_quad_rom_run,
    bg_ann_rom_run): which sample each slot of the launch works on.  The samples are independent, so this is a pure
    scheduling decision -- the results are the same bit for bit -- but their cost is not uniform: the iteration count of
    a sample follows its convection parameter mu1 (correlation 0.99 on the bench sweep).  Workgroup k of ``grid``
    persistent workgroups takes the slots k, k + grid, ...; with ``group`` = 4 (bg_quad_rom_run) four consecutive slots
    share a workgroup and every pass lasts until the slowest of the four has converged.  So: sort by mu1, keep
    neighbours together inside a group, and deal the groups out in boustrophedon order (round 0 left to right, round 1
    right to left, ...) so that every workgroup gets the same mix of expensive and cheap ones.  Returns an int32
    device tensor, or None when there is nothing to balance."""
    B = mu1d.numel()
    units = B // group                                   # whole groups; a ragged tail keeps the last slots
    if B <= group or (group == 1 and B <= grid):
        return None
    rank = torch.argsort(mu1d, descending=True)
    u = torch.arange(units, device=mu1d.device)
    rnd, pos = u // grid, u % grid
    size = torch.clamp(units - rnd * grid, max=grid)      # units in this round (the last one may be short)
    slot = rnd * grid + torch.where(rnd % 2 == 0, pos, size - 1 - pos)
    order = torch.empty_like(rank)
    head = units * group
    order[:head].view(units, group)[slot] = rank[:head].view(units, group)
    order[head:] = rank[head:]
    return order.to(torch.int32)


def _cu_count(device):
    return torch.cuda.get_device_properties(device).multi_processor_count


def pod_prom_run_wide(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E=0.0, tol=1e-6, max_it=20, device=None, PhiP=None, options=0,
                      balance=True):
    """``pod_prom_burgers`` for bases of 41 .. 96 modes with the whole time loop on the device (bg_rom_run_wide): the basis
    streams through LDS, the reduced system's accumulators are spread over the four waves of the sample's workgroup.
    Samples whose elimination would have needed a row exchange come back marked and are redone through the library
    path (LU with partial pivoting).  ``PhiP``: the padded basis copy of a previous call (``res.PhiP``) to reuse."""
    L = _lib.load()
    device = _lib.require_device(device)
    opts = _lib.mesh_options(check_mesh(X), supg=True) | options
    Xd = _as_dev(X, device)
    N = Xd.numel()
    Phid = _as_dev(Phi, device)
    if Phid.dim() != 2 or Phid.shape[0] != N:
        raise ValueError("Phi must have one row per mesh node")
    r = Phid.shape[1]
    if PhiP is None:
        NPAD = (N + 63) // 64 * 64
        PhiP = torch.zeros((NPAD + 2, 96), dtype=torch.float64, device=device)
        PhiP[1:N + 1, :r] = Phid
    assert PhiP.numel() == L.bg_rom_run_wide_phi_elems(N)
    u0d, mu1d, mu2d = _batch_inputs(u0, mu1, mu2, N, device)
    B = mu1d.numel()
    hist = torch.empty((B, nsteps + 1, N), dtype=torch.float64, device=device)
    iters = torch.zeros((B, nsteps), dtype=torch.int32, device=device)
    flags = torch.zeros((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    order = sample_order(mu1d, _cu_count(device)) if balance else None
    with torch.cuda.device(device):
        rc = L.bg_rom_run_wide(N, B, r, int(nsteps), proj, _lib.ptr(Xd), _lib.ptr(PhiP), _lib.ptr(u0d), _lib.ptr(mu1d),
                               _lib.ptr(mu2d), float(dt), float(E), float(tol), int(max_it), int(opts), _lib.ptr(hist),
                               _lib.ptr(iters), _lib.ptr(flags), _lib.ptr(info), _lib.ptr(order), _lib.stream_ptr(device))
    _lib.check(rc, "bg_rom_run_wide")
    redo = (info == _lib.BG_INFO_NEEDS_PIVOTING).nonzero().squeeze(1)
    if redo.numel():                                     # rare: np.linalg.solve would have exchanged rows
        rr = _pod_prom_run_library(Xd, u0d[redo], mu1d[redo], mu2d[redo], dt, nsteps, Phid, proj, E, tol, max_it, device)
        hist[redo], iters[redo], flags[redo] = rr.hist, rr.iters, rr.flags
        info[redo] = 0
    res = FomResult(hist, iters, flags)
    res.info = info
    res.redone = int(redo.numel())
    res.PhiP = PhiP
    res._keep = (Xd, u0d, mu1d, mu2d)
    return res


def check_singular(res):
    """np.linalg.solve raises LinAlgError('Singular matrix') at :767; the device loop records it per sample."""
    info = getattr(res, "info", None)
    if info is not None and bool(info.ne(0).any()):
        raise SingularReducedSystem("Singular matrix")
    return res


def pod_prom_run(X, u0, mu1, mu2, dt, nsteps, Phi, projection="Galerkin", E=0.0, tol=1e-6, max_it=20,
                 device=None, fused=True):
    """Batched ``pod_prom_burgers``; ``projection`` is case-sensitive like the reference (:754-764).
    ``fused`` (default): the device-side time loop bg_rom_run where it applies (N <= 512, r <= 40); otherwise, or
    with ``fused=False``, the batched iteration bg_rom_reduce -> bg_lu_solve_update driven from the host."""
    if projection not in ("Galerkin", "LSPG"):
        raise ValueError(f"Projection method '{projection}' is not available. Please use 'Galerkin' or 'LSPG'.")
    proj = PROJ[projection.lower()]
    L = _lib.load()
    r_in, n_in = np.shape(Phi)[1], np.shape(Phi)[0]
    if fused and L.bg_rom_run_max_r() < r_in <= L.bg_rom_run_wide_max_r() and n_in <= 512:
        return check_singular(pod_prom_run_wide(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E, tol, max_it, device))
    if r_in > L.bg_rom_max_r() or n_in > L.bg_rom_max_n():
        return _pod_prom_run_library(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E, tol, max_it, device)
    if fused and r_in <= L.bg_rom_run_max_r():
        return check_singular(pod_prom_run_fused(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E, tol, max_it, device))
    c = _setup(X, u0, mu1, mu2, dt, E, device)
    Phid = _as_dev(Phi, c.device)
    if Phid.shape[0] != c.N:
        raise ValueError("Phi must have one row per mesh node")
    r = Phid.shape[1]
    PhiT = Phid.t().contiguous()
    hist, iters, flags = _alloc_hist(c, nsteps)
    Ar, br, wtu, G = _workspace(c, r)
    st = _IterState(c, r)
    q = torch.zeros((c.B, r), dtype=torch.float64, device=c.device)
    U0 = c.u0.clone()
    for n in range(nsteps):
        _mass_rhs(c, U0, G)
        st.begin_step()
        first = n == 0                      # u0 is not in span(Phi): the very first assembly reads it from HBM
        while True:
            if first:
                rom_reduce(c, Phid, U0, G, proj, True, st.active, Ar, br, wtu)
                first = False
            else:                           # u_k = Phi q formed in-kernel                 (:773)
                rom_reduce_lifted(c, Phid, q, U0, G, proj, True, st.active, Ar, br, wtu)
            if st.solve_update(1, Ar, br, wtu, q, tol, max_it) == 0:     # q = Phi^T U0 + dq (:767-776)
                break
        rom_lift(c, Phid, q, U0)            # U[:, n+1] = Phi q                            (:779)
        iters[:, n] = st.k
        hist[:, n + 1] = U0
    flags |= st.flags
    return FomResult(hist, iters, flags)


def _batched_solve(A, b, considered):
    """x = solve(A, b) over a batch (rocSOLVER LU).  Like numpy (:767), an exactly singular FINITE system of a
    sample that is still iterating raises; a system that already holds NaN/Inf (a diverged sample) does not --
    its solution is simply non-finite, which the caller flags -- and neither does a sample that is masked out.
    The batch goes through in chunks: hipblasDgetrfBatched fails to allocate its workspace beyond about
    n^2 * batch = 1e7 (n = 160: 256 systems pass, 512 do not; tools/probe_solve.py)."""
    B, n, _ = A.shape
    chunk = max(16, int(6.0e6 / (n * n)))
    x = torch.empty_like(b)
    info = torch.empty((B,), dtype=torch.int32, device=A.device)
    for b0 in range(0, B, chunk):
        xs, inf = torch.linalg.solve_ex(A[b0:b0 + chunk], b[b0:b0 + chunk], check_errors=False)
        x[b0:b0 + chunk] = xs
        info[b0:b0 + chunk] = inf
    bad = (info != 0) & considered & torch.isfinite(A).all(dim=2).all(dim=1)
    if bool(bad.any()):
        raise SingularReducedSystem("Singular matrix")
    return x


def _pod_prom_run_library(X, u0, mu1, mu2, dt, nsteps, Phi, proj, E, tol, max_it, device):
    """POD PROM for bases beyond the register-resident MFMA kernels (r > 47 or N > 512; the thesis
    also runs r = 96, 160, 227): bg_fom_assemble (HIP) for A(u), R(u), then the projection and
    the reduced solve as plain library calls over the batch (rocBLAS GEMMs, rocSOLVER LU).
    Same loop and stopping rule as the fused path; slower, but it keeps the API total."""
    from . import fom as _fom
    L = _lib.load()
    c = _setup(X, u0, mu1, mu2, dt, E, device, max_n=L.bg_fom_max_n())
    Xh = c.X.cpu().numpy()
    Phid = _as_dev(Phi, c.device)
    if Phid.shape[0] != c.N:
        raise ValueError("Phi must have one row per mesh node")
    PhiT = Phid.t().contiguous()
    zrow = torch.zeros((1, Phid.shape[1]), dtype=torch.float64, device=c.device)
    PhiT_dn = torch.cat([zrow, Phid[:-1]], 0).t().contiguous()     # column i holds Phi[i-1]
    PhiT_up = torch.cat([Phid[1:], zrow], 0).t().contiguous()      # column i holds Phi[i+1]
    hist, iters, flags = _alloc_hist(c, nsteps)
    U0 = c.u0.clone()
    for n in range(nsteps):
        Un = U0.clone()
        active = torch.ones((c.B,), dtype=torch.bool, device=c.device)
        k = torch.zeros((c.B,), dtype=torch.int32, device=c.device)
        while True:
            lo, di, up, rhs = _fom.fom_assemble(Xh, U0, Un, c.mu1, c.mu2, c.dt, E=c.E, supg=True, device=c.device)
            # (A Phi)^T as (B, r, N): the Galerkin projection is then ONE GEMM over B r rows against Phi
            YT = di.unsqueeze(1) * PhiT + lo.unsqueeze(1) * PhiT_dn + up.unsqueeze(1) * PhiT_up
            if proj == _lib.BG_PROJ_GALERKIN:
                Ar = torch.matmul(YT, Phid).transpose(1, 2)                  # Phi^T A Phi       (:756)
                br = -(rhs @ Phid)                                           # Phi^T R, R = -rhs (:757)
            else:
                Ar = torch.matmul(YT, YT.transpose(1, 2))                    # (A Phi)^T (A Phi) (:761)
                br = -torch.matmul(YT, rhs.unsqueeze(-1)).squeeze(-1)
            dq = _batched_solve(Ar, -br, active)                            # np.linalg.solve (:767)
            q = U0 @ Phid + dq
            U1 = q @ PhiT
            err = torch.linalg.vector_norm(dq, dim=1) / torch.linalg.vector_norm(q, dim=1)
            U0 = torch.where(active[:, None], U1, U0)
            k += active.to(torch.int32)
            flags |= (active & ~torch.isfinite(err)).to(torch.int32) * _lib.BG_FLAG_NONFINITE
            active = active & (err > tol) & (k < max_it)
            if not bool(active.any()):
                break
        flags |= (k >= max_it).to(torch.int32) * _lib.BG_FLAG_HIT_CAP
        iters[:, n] = k
        hist[:, n + 1] = U0
    return FomResult(hist, iters, flags)


# ------------------------------------------------------------ quadratic manifold
def sym_index_tables(n, device):
    """Row-major upper-triangle pair order of get_sym (:263-273) and the (n, n) lookup of
    the pair index, with the factor (1 + delta_ab) of get_dQ_dq (:292-312)."""
    I, J = np.triu_indices(n)
    idx = np.zeros((n, n), dtype=np.int64)
    idx[I, J] = np.arange(len(I))
    idx[J, I] = np.arange(len(I))
    fac = np.ones((n, n)) + np.eye(n)
    return (torch.as_tensor(I, device=device), torch.as_tensor(J, device=device),
            torch.as_tensor(idx, device=device), torch.as_tensor(fac, device=device))


def quad_tangent_tensor(Phid, Hd):
    """H3[i][a][c] = H[i][pair(a, c)] (1 + delta_ac), (N, n, n): tangent = Phi + H3 . q (:1120-1123 with get_dQ_dq
    :292-312 folded in)."""
    n = Phid.shape[1]
    _, _, idx, fac = sym_index_tables(n, Phid.device)
    return (Hd[:, idx] * fac).contiguous()


class QuadFusedPlan:
    """The operand copies bg_quad_rom_run reads, built once per (Phi, H) on the device (include/burgers_hip.h):
    Phi^T zero padded, the accumulator seeds of the tangent tiles, and H3 in the A-operand order of the matrix
    instruction.  None-like (``ok`` False) when the basis is beyond the kernel (N > 512 or n > 40)."""

    def __init__(self, Phi, H, device):
        L = _lib.load()
        self.Phi, self.H = _as_dev(Phi, device), _as_dev(np.ascontiguousarray(H) if isinstance(H, np.ndarray) else H, device)
        N, n = self.Phi.shape
        self.N, self.n = N, n
        if self.H.shape != (N, n * (n + 1) // 2):
            raise ValueError("Phi must be (N, n) and H (N, n(n+1)/2)")
        self.ok = N <= 512 and n <= L.bg_quad_rom_max_n()
        if not self.ok:
            return
        f64 = dict(dtype=torch.float64, device=device)
        NG, NPAD = (N + 3) // 4, (N + 63) // 64 * 64
        self.PhiT = torch.zeros((40, NPAD), **f64)
        self.PhiT[:n, :N] = self.Phi.t()
        Pp = torch.zeros((4 * NG, 40), **f64)
        Pp[:N, :n] = self.Phi
        # accumulator seeds of the tangent tiles: (rg, blk, c, i) -> [rg][c][4 i + blk]
        self.Phif = Pp.reshape(NG, 4, 10, 4).permute(0, 2, 3, 1).contiguous()
        H3p = torch.zeros((4 * NG, 40, 40), **f64)
        H3p[:N, :n, :n] = quad_tangent_tensor(self.Phi, self.H)
        # H3 of a mesh row is symmetric: only its upper 4 x 4 blocks (a <= b, row-major) travel.  Block (a, b), lane 16 k + 4 blk + i
        # = H3[4 rg + blk][4 a + i][4 b + k]; two blocks per 16-byte slot: [rg][slot (28)][lane][2] (block 55 = zero padding)
        Hb = H3p.reshape(NG, 4, 10, 4, 10, 4).permute(0, 2, 4, 5, 1, 3)           # (rg, a, b, k, blk, i)
        A_, B_ = np.triu_indices(10)
        Hu = Hb[:, torch.as_tensor(A_, device=device), torch.as_tensor(B_, device=device)].reshape(NG, 55, 64)
        Hu = torch.cat([Hu, torch.zeros((NG, 1, 64), **f64)], 1)
        self.H3f = Hu.reshape(NG, 28, 2, 64).permute(0, 1, 3, 2).contiguous()
        assert self.H3f.numel() == L.bg_quad_rom_h3f_elems(N) and self.Phif.numel() == L.bg_quad_rom_phif_elems(N)


def quadratic_run_fused(X, u0, mu1, mu2, dt, nsteps, plan, proj, E=0.0, newton_tol=1e-6, newton_itmax=25, device=None,
                        balance=True):
    """``pod_quadratic_manifold`` for a batch with the whole time loop on the device (bg_quad_rom_run): four samples
    per workgroup, no host in the loop; the reduced solve pivots like np.linalg.solve."""
    L = _lib.load()
    device = _lib.require_device(device)
    opts = _lib.mesh_options(check_mesh(X), supg=False)
    Xd = _as_dev(X, device)
    N = Xd.numel()
    if N != plan.N:
        raise ValueError("Phi must be (N, n) and H (N, n(n+1)/2)")
    u0d, mu1d, mu2d = _batch_inputs(u0, mu1, mu2, N, device)
    B = mu1d.numel()
    hist = torch.empty((B, nsteps + 1, N), dtype=torch.float64, device=device)
    iters = torch.zeros((B, nsteps), dtype=torch.int32, device=device)
    flags = torch.zeros((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    order = sample_order(mu1d, _cu_count(device), group=4) if balance else None
    with torch.cuda.device(device):
        rc = L.bg_quad_rom_run(N, B, plan.n, int(nsteps), proj, _lib.ptr(Xd), _lib.ptr(plan.PhiT), _lib.ptr(plan.Phif),
                               _lib.ptr(plan.H3f), _lib.ptr(u0d), _lib.ptr(mu1d), _lib.ptr(mu2d), float(dt), float(E),
                               float(newton_tol), int(newton_itmax), int(opts), _lib.ptr(hist), _lib.ptr(iters),
                               _lib.ptr(flags), _lib.ptr(info), _lib.ptr(order), _lib.stream_ptr(device))
    _lib.check(rc, "bg_quad_rom_run")
    res = FomResult(hist, iters, flags)
    res.info = info              # checked lazily by the caller (a readback would synchronise)
    res._keep = (plan, Xd, u0d, mu1d, mu2d)
    return res


def quadratic_run(X, u0, mu1, mu2, dt, nsteps, Phi, H, projection="LSPG", E=0.0, newton_tol=1e-6,
                  newton_itmax=25, device=None, fused=True, plan=None):
    """Batched ``pod_quadratic_manifold`` (no SUPG term in this variant, :1142).  ``fused`` (default): the device-side
    time loop bg_quad_rom_run where it applies (N <= 512, n <= 40); otherwise, or with ``fused=False``, the batched
    iteration (bg_quad_tangent -> bg_rom_reduce_frag -> bg_lu_solve_update -> decode GEMM) driven from the host.
    ``plan``: a QuadFusedPlan of (Phi, H) to reuse across calls."""
    p = projection.lower()
    if p not in PROJ:
        raise ValueError("projection must be 'Galerkin' or 'LSPG'")
    proj = PROJ[p]
    if fused:
        dev = _lib.require_device(device)
        if plan is None:
            plan = QuadFusedPlan(Phi, H, dev)
        if plan.ok:
            return check_singular(quadratic_run_fused(X, u0, mu1, mu2, dt, nsteps, plan, proj, E, newton_tol, newton_itmax, dev))
    c = _setup(X, u0, mu1, mu2, dt, E, device)
    Phid, Hd = _as_dev(Phi, c.device), _as_dev(np.ascontiguousarray(H) if isinstance(H, np.ndarray) else H, c.device)
    n = Phid.shape[1]
    kk = n * (n + 1) // 2
    if Hd.shape != (c.N, kk) or Phid.shape[0] != c.N:
        raise ValueError("Phi must be (N, n) and H (N, n(n+1)/2)")
    I, J, idx, fac = sym_index_tables(n, c.device)
    PhiT, HT = Phid.t().contiguous(), Hd.t().contiguous()
    # H3[i, a, b] = H[i, pair(a, b)] * (1 + delta_ab):  tangent = Phi + H3 . q   (:1120-1123)
    H3 = (Hd[:, idx] * fac).reshape(c.N * n, n).contiguous()

    # decode u = Phi q + H Q(q) (:1116-1118) as ONE GEMM [q | Q(q)] . [Phi^T; H^T]; the left operand comes from bg_quad_features
    WT = torch.cat([PhiT, HT], 0).contiguous()               # (n + k, N)
    I32, J32 = I.to(torch.int32).contiguous(), J.to(torch.int32).contiguous()
    feat = torch.empty((c.B, n + kk), dtype=torch.float64, device=c.device)

    def decode(q):
        with torch.cuda.device(c.device):
            _lib.check(c.L.bg_quad_features(c.B, n, _lib.ptr(q), _lib.ptr(I32), _lib.ptr(J32), _lib.ptr(feat), c.stream()),
                       "bg_quad_features")
        return feat @ WT

    hist, iters, flags = _alloc_hist(c, nsteps)
    Ar, br, _, G = _workspace(c, n)
    st = _IterState(c, n)
    H3t = H3.t().contiguous()
    Phi_flat = Phid.reshape(1, c.N * n)
    per = int(c.L.bg_rom_frag_elems(c.N, n))
    Wf = torch.zeros((c.B, per), dtype=torch.float64, device=c.device) if per > 0 else None
    if Wf is not None:                                   # zero-padded operands of bg_quad_tangent
        NP = int(c.L.bg_rom_frag_pad(n))
        H3p = torch.nn.functional.pad(H3.reshape(c.N, n, n), (0, NP - n)).contiguous()
        qpad = torch.zeros((c.B, NP), dtype=torch.float64, device=c.device)
    Un = c.u0.clone()
    for m in range(nsteps):
        _mass_rhs(c, Un, G)
        q = (Un @ Phid).contiguous()                         # first guess            (:1129)
        u = decode(q).contiguous()
        st.begin_step()
        while True:
            if Wf is not None:                               # fused HIP tangent -> fragment-major W -> MFMA reduce
                if NP != n:
                    qpad[:, :n] = q                          # (n = 40 needs no padding: q itself is the operand)
                with torch.cuda.device(c.device):
                    _lib.check(c.L.bg_quad_tangent(c.N, c.B, n, _lib.ptr(Phid), _lib.ptr(H3p), _lib.ptr(qpad if NP != n else q),
                                                   _lib.ptr(st.active), _lib.ptr(Wf), c.stream()), "bg_quad_tangent")
                    _lib.check(c.L.bg_rom_reduce_frag(c.N, c.B, n, proj, _lib.ptr(c.X), _lib.ptr(Wf), _lib.ptr(u),
                                                      _lib.ptr(G), _lib.ptr(c.hfs), _lib.ptr(c.mu1), c.dt, c.E, c.mesh_opt,
                                                      _lib.ptr(st.active), _lib.ptr(Ar), _lib.ptr(br), None, c.stream()),
                               "bg_rom_reduce_frag")
            else:                                            # sizes beyond the fused kernels: library GEMM
                T = torch.addmm(Phi_flat, q, H3t).reshape(c.B, c.N, n)
                rom_reduce(c, T, u, G, proj, False, st.active, Ar, br, None)
            left = st.solve_update(2, Ar, br, None, q, newton_tol, newton_itmax)     # q += dq (:1161-1169)
            u = decode(q).contiguous()                       # inactive samples keep their q, hence their u
            if left == 0:
                break
        iters[:, m] = st.k
        hist[:, m + 1] = u
        Un = u
    flags |= st.flags                                        # HIT_CAP = "Newton did not converge" (:1171)
    return FomResult(hist, iters, flags)


# ----------------------------------------------------------------------- POD-ANN
def _mlp_layers(model):
    """Recognise a plain MLP: nn.Sequential of Linear / ELU|ReLU|Tanh, or the reference's POD_ANN
    class (fc1..fcK + self.elu, POD-ANN/pod_ann.py:38-56).  Returns [(Linear, act-or-None)] or None."""
    import torch.nn as nn
    acts = (nn.ELU, nn.ReLU, nn.Tanh)
    if isinstance(model, nn.Sequential):
        mods = list(model)
        out, i = [], 0
        while i < len(mods):
            if not isinstance(mods[i], nn.Linear):
                return None
            act = None
            if i + 1 < len(mods) and isinstance(mods[i + 1], acts):
                act = mods[i + 1]; i += 1
            out.append((mods[i - (1 if act is not None else 0)], act))
            i += 1
        return out if out else None
    fcs = []
    while isinstance(getattr(model, f"fc{len(fcs) + 1}", None), nn.Linear):
        fcs.append(getattr(model, f"fc{len(fcs) + 1}"))
    elu = getattr(model, "elu", None)
    if len(fcs) >= 2 and isinstance(elu, nn.ELU):
        return [(fc, elu if i < len(fcs) - 1 else None) for i, fc in enumerate(fcs)]
    return None


def _mlp_forward_jacobian(layers, x, want_jac=True):
    """Forward pass and forward-mode input-Jacobian of a recognised MLP as batched GEMMs."""
    import torch.nn as nn
    J = None
    for lin, act in layers:
        z = torch.addmm(lin.bias, x, lin.weight.t()) if lin.bias is not None else x @ lin.weight.t()
        if want_jac:
            J = lin.weight.unsqueeze(0).expand(x.shape[0], -1, -1) if J is None else torch.matmul(lin.weight, J)
        if act is None:
            x = z
            continue
        if isinstance(act, nn.ELU):
            x = torch.nn.functional.elu(z, alpha=act.alpha)
            d = torch.where(z > 0, torch.ones_like(z), act.alpha * torch.exp(z))
        elif isinstance(act, nn.ReLU):
            x = torch.relu(z); d = (z > 0).to(z.dtype)
        else:
            x = torch.tanh(z); d = 1.0 - x * x
        if want_jac:
            J = d.unsqueeze(-1) * J
    return x, J


class AnnEvaluator:
    """model(q) and d model / d q for a batch, in the model's dtype.  A recognised MLP runs as a
    few batched GEMMs (checked against the module itself on a probe); anything else falls back
    to torch.func (vmap of jacfwd), the batched stand-in for the reference's per-sample
    torch.autograd.functional.jacobian (:1254-1275)."""

    _live = weakref.WeakSet()         # evaluators that currently own a captured graph

    def __init__(self, model, n, dtype):
        self.model, self.dtype = model, dtype
        self.layers = _mlp_layers(model)
        if self.layers is not None:
            dev = next(model.parameters()).device
            probe = torch.linspace(-1.0, 1.0, 4 * n, device=dev, dtype=dtype).reshape(4, n)
            with torch.no_grad():
                ok = torch.allclose(model(probe), _mlp_forward_jacobian(self.layers, probe, False)[0],
                                    rtol=1e-3 if dtype != torch.float32 else 1e-5, atol=1e-5)
            if not ok:
                self.layers = None

    def forward(self, q):
        with torch.no_grad():
            return self.model(q.to(self.dtype)).to(torch.float64)

    def jacobian(self, q):
        with torch.no_grad():
            x = q.to(self.dtype)
            if self.layers is not None:
                return _mlp_forward_jacobian(self.layers, x)[1].to(torch.float64)
            return ann_jacobian(self.model, x).to(torch.float64)

    # ---- value and Jacobian in one pass (recognised fp32 MLP on the device): the value and the n tangent
    # directions are the 1 + n rows of one matrix per sample, every linear layer is ONE GEMM over B (1 + n)
    # rows, every activation one bg_mlp_act_jvp launch.  Captured in a hipGraph and replayed.
    def bind(self, B, n, device, jt_out, qs_out):
        """jt_out (B, n, nbar) and qs_out (B, nbar): fp64 destinations of dN^T and N(q)."""
        self.jt_out, self.qs_out = jt_out, qs_out
        self._graph = None
        self.fused = self.layers is not None and self.dtype == torch.float32 and device.type == "cuda"
        if not self.fused:
            return
        import torch.nn as nn
        L = _lib.load()
        self.q_in = torch.zeros((B, n), dtype=torch.float64, device=device)
        x0 = torch.zeros((B, 1 + n, n), dtype=torch.float32, device=device)
        x0[:, 1:, :] = torch.eye(n, dtype=torch.float32, device=device)
        kinds = {type(None): _lib.BG_ACT_NONE, nn.ELU: _lib.BG_ACT_ELU, nn.ReLU: _lib.BG_ACT_RELU, nn.Tanh: _lib.BG_ACT_TANH}
        plan = [(lin.weight.detach().t().contiguous(), None if lin.bias is None else lin.bias.detach().contiguous(),
                 kinds[type(act)], float(getattr(act, "alpha", 1.0))) for lin, act in self.layers]

        def run():
            x0[:, 0, :] = self.q_in                                           # fp64 -> fp32 like q.to(float32)
            x = x0
            for wt, bias, kind, alpha in plan:
                z = torch.matmul(x, wt)                                       # (B, 1+n, h): one GEMM
                with torch.cuda.device(device):
                    _lib.check(L.bg_mlp_act_jvp(B, 1 + n, wt.shape[1], _lib.ptr(z), _lib.ptr(bias) if bias is not None else None,
                                                kind, alpha, _lib.stream_ptr(device)), "bg_mlp_act_jvp")
                x = z
            self.qs_out.copy_(x[:, 0, :])
            self.jt_out.copy_(x[:, 1:, :])

        self._run = run
        self._device = device
        import os
        if os.environ.get("BG_ANN_GRAPH", "1") == "0":        # eager launches: the capture is an optimisation only
            return
        # Graph lifetime is explicit ("AnnEvaluator graph lifetime"): no captured graph of an OLDER
        # evaluator is alive -- or still replaying -- while a new capture runs, and none is left to the garbage
        # collector, which may run at any point, also inside someone else's capture.
        for other in list(AnnEvaluator._live):
            if other is not self:
                other.release()
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):                      # warm-up outside capture (lazy inits, GEMM selection)
                run(); run()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            import gc
            was_enabled = gc.isenabled()
            gc.disable()           # no finaliser (of anything holding device memory or a graph) runs inside the capture
            try:
                with torch.cuda.graph(g):
                    run()
            finally:
                if was_enabled:
                    gc.enable()
            self._graph = g
            AnnEvaluator._live.add(self)
        except Exception:                                      # capture is an optimisation only
            self._graph = None

    def release(self):
        """Destroy the captured graph NOW: after the device has finished every replay of it, outside any capture --
        not whenever the collector gets to it."""
        if getattr(self, "_graph", None) is not None:
            torch.cuda.synchronize(self._device)
            self._graph = None
        AnnEvaluator._live.discard(self)

    def eval(self, q):
        """N(q) into qs_out and (dN/dq)^T into jt_out for the batch q (B, n), float64 in and out."""
        if not self.fused:
            self.qs_out.copy_(self.forward(q))
            self.jt_out.copy_(self.jacobian(q).transpose(1, 2))
            return
        self.q_in.copy_(q)
        if self._graph is not None:
            self._graph.replay()
        else:
            self._run()


def ann_jacobian(model, q32):
    """Batched input-Jacobian (B, nbar, n) of ``model`` in fp32; forward mode, since n << nbar.
    Stand-in for the per-sample torch.autograd.functional.jacobian of :1254-1275."""
    from torch.func import jacfwd, vmap
    return vmap(jacfwd(lambda z: model(z.unsqueeze(0)).squeeze(0)))(q32)


class _ClosureTangent:
    """(U_p + U_s dN)^T for a batch, as ONE GEMM: [dN^T | I] (B n x (nbar+n)) . [U_s^T ; U_p^T] ((nbar+n) x N).
    The output is the column-major per-sample tangent bg_rom_reduce reads with BG_OPT_W_COLMAJOR; the batch of
    B small (N x nbar) . (nbar x n) GEMMs it replaces ran at 2 TFLOP/s.  reference: :1224, :1361."""

    def __init__(self, Up, Us, B):
        n, self.nbar = Up.shape[1], Us.shape[1]
        self.rhs = torch.cat([Us.t(), Up.t()], 0).contiguous()              # (nbar + n, N)
        self.lhs = torch.zeros((B, n, self.nbar + n), dtype=torch.float64, device=Up.device)
        self.lhs[:, :, self.nbar:] = torch.eye(n, dtype=torch.float64, device=Up.device)

    @property
    def jt(self):
        """The (B, n, nbar) slot for dN^T."""
        return self.lhs[:, :, :self.nbar]

    def gemm(self):
        return torch.matmul(self.lhs, self.rhs)

    def __call__(self, dN):
        """dN: (B, nbar, n) -> (B, n, N)."""
        self.jt.copy_(dN.transpose(1, 2))
        return self.gemm()


def _ann_fused_plan(model, n, nbar, N, dtype, device):
    """The closure as bg_ann_rom_run wants it, or None when the device-side loop does not apply (not a plain fp32
    MLP the evaluator recognises, or beyond bg_ann_rom_limits)."""
    import ctypes
    import torch.nn as nn
    if dtype != torch.float32 or N > 512:
        return None
    ann = AnnEvaluator(model, n, dtype)
    if ann.layers is None:
        return None
    L = _lib.load()
    lim = [ctypes.c_int() for _ in range(4)]
    L.bg_ann_rom_limits(*[ctypes.byref(v) for v in lim])
    max_n, max_nbar, max_w, max_l = (v.value for v in lim)
    kinds = {type(None): _lib.BG_ACT_NONE, nn.ELU: _lib.BG_ACT_ELU, nn.ReLU: _lib.BG_ACT_RELU, nn.Tanh: _lib.BG_ACT_TANH}
    widths = [ann.layers[0][0].in_features] + [lin.out_features for lin, _ in ann.layers]
    if (n > max_n or nbar > max_nbar or len(ann.layers) > max_l or max(widths[1:]) > max_w or widths[0] != n
            or widths[-1] != nbar or any(type(act) not in kinds for _, act in ann.layers)):
        return None
    f32 = dict(dtype=torch.float32, device=device)
    # W^T zero-padded to [in rounded up to 4][out rounded up to 8]: a thread fetches the weights of 8 outputs of one input
    # as two 16-byte loads, four inputs at a time, with no bounds checks in the kernel
    wts = [torch.nn.functional.pad(lin.weight.detach().to(**f32).t(), (0, -lin.out_features % 8, 0, -lin.in_features % 4)).contiguous()
           for lin, _ in ann.layers]
    biases = [None if lin.bias is None else lin.bias.detach().to(**f32).contiguous() for lin, _ in ann.layers]
    nl = len(wts)
    return dict(
        keep=(wts, biases), nl=nl,
        widths=(ctypes.c_int * (nl + 1))(*widths),
        wt=(ctypes.c_void_p * nl)(*[w.data_ptr() for w in wts]),
        bias=(ctypes.c_void_p * nl)(*[None if b is None else b.data_ptr() for b in biases]),
        acts=(ctypes.c_int * nl)(*[kinds[type(act)] for _, act in ann.layers]),
        alphas=(ctypes.c_float * nl)(*[float(getattr(act, "alpha", 1.0)) for _, act in ann.layers]))


def pod_ann_run_fused(X, u0, mu1, mu2, dt, nsteps, U_p, U_s, model, proj, E=0.0, tol=1e-6, max_it=50, device=None,
                      options=0, plan=None, balance=True):
    """``pod_ann_prom`` for a batch with the whole time loop on the device (bg_ann_rom_run): one workgroup per sample,
    the closure MLP evaluated in-kernel in float32, the reduced solve with partial pivoting.  Returns None when the model
    is outside what that kernel covers."""
    L = _lib.load()
    device = _lib.require_device(device)
    opts = _lib.mesh_options(check_mesh(X), supg=True) | options
    Xd = _as_dev(X, device)
    N = Xd.numel()
    Up, Us = _as_dev(U_p, device), _as_dev(U_s, device)
    if Up.dim() != 2 or Us.dim() != 2 or Up.shape[0] != N or Us.shape[0] != N:
        raise ValueError("U_p and U_s must have one row per mesh node")
    n, nbar = Up.shape[1], Us.shape[1]
    if plan is None:
        plan = _ann_fused_plan(model.to(device=device, dtype=torch.float32).eval(), n, nbar, N, torch.float32, device)
    if plan is None:
        return None
    UT = torch.zeros((-(-(n + nbar) // 8) * 8, N), dtype=torch.float64, device=device)     # [U_p^T; U_s^T; zero rows]
    UT[:n] = Up.t()
    UT[n:n + nbar] = Us.t()
    u0d, mu1d, mu2d = _batch_inputs(u0, mu1, mu2, N, device)
    B = mu1d.numel()
    hist = torch.empty((B, nsteps + 1, N), dtype=torch.float64, device=device)
    iters = torch.zeros((B, nsteps), dtype=torch.int32, device=device)
    flags = torch.zeros((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    order = sample_order(mu1d, 2 * _cu_count(device)) if balance else None
    with torch.cuda.device(device):
        rc = L.bg_ann_rom_run(N, B, n, nbar, int(nsteps), proj, _lib.ptr(Xd), _lib.ptr(UT), _lib.ptr(u0d),
                              _lib.ptr(mu1d), _lib.ptr(mu2d), plan["nl"], plan["widths"], plan["wt"], plan["bias"],
                              plan["acts"], plan["alphas"], float(dt), float(E), float(tol), int(max_it), int(opts),
                              _lib.ptr(hist), _lib.ptr(iters), _lib.ptr(flags), _lib.ptr(info), _lib.ptr(order),
                              _lib.stream_ptr(device))
    _lib.check(rc, "bg_ann_rom_run")
    res = FomResult(hist, iters, flags)
    res.info = info
    res._keep = (plan, UT, Xd, u0d, mu1d, mu2d)      # the launch is asynchronous: its operands live as long as the result
    return res


def pod_ann_run(X, u0, mu1, mu2, dt, nsteps, U_p, U_s, model, projection="LSPG", E=0.0, tol=1e-6, max_it=50,
                device=None, ann_dtype=torch.float32, fused=True):
    """Batched ``pod_ann_prom``.  The MLP and its Jacobian are evaluated in ``ann_dtype`` (the
    reference uses float32, :1219,:1241); everything else is fp64.  ``fused`` (default): the device-side time loop
    bg_ann_rom_run when the closure is a plain float32 MLP within bg_ann_rom_limits; otherwise, or with
    ``fused=False``, the batched iteration driven from the host (MLP layers as GEMMs through PyTorch-ROCm)."""
    p = projection.lower()
    if p not in PROJ:
        raise ValueError("projection must be 'Galerkin' or 'LSPG'")
    proj = PROJ[p]
    if fused and ann_dtype == torch.float32:
        res = pod_ann_run_fused(X, u0, mu1, mu2, dt, nsteps, U_p, U_s, model, proj, E, tol, max_it, device)
        if res is not None:
            return check_singular(res)
    c = _setup(X, u0, mu1, mu2, dt, E, device)
    Up, Us = _as_dev(U_p, c.device), _as_dev(U_s, c.device)
    n = Up.shape[1]
    UpT, UsT = Up.t().contiguous(), Us.t().contiguous()
    tangent = _ClosureTangent(Up, Us, c.B)
    model = model.to(device=c.device, dtype=ann_dtype).eval()
    ann = AnnEvaluator(model, n, ann_dtype)
    qs = torch.zeros((c.B, Us.shape[1]), dtype=torch.float64, device=c.device)
    ann.bind(c.B, n, c.device, tangent.jt, qs)
    hist, iters, flags = _alloc_hist(c, nsteps)
    Ar, br, _, G = _workspace(c, n)
    st = _IterState(c, n)
    U0 = c.u0.clone()
    try:
        for nt in range(nsteps):
            _mass_rhs(c, U0, G)
            qp = (U0 @ Up).contiguous()                                         # (:1197)
            st.begin_step()
            ann.eval(qp)                                                        # dN at the first guess (:1219)
            while True:
                rom_reduce(c, tangent.gemm(), U0, G, proj, True, st.active, Ar, br, None, colmajor=True)   # (:1224)
                left = st.solve_update(3, Ar, br, None, qp, tol, max_it)        # q_p += dq           (:1237-1244)
                ann.eval(qp)                            # q_s = N(q_p) for the decode (:1241) and dN for the next pass
                U0 = (qp @ UpT + qs @ UsT).contiguous()                         # (:1242)
                if left == 0:
                    break
            iters[:, nt] = st.k
            hist[:, nt + 1] = U0
        flags |= st.flags
    finally:
        ann.release()               # also on an exception (SingularReducedSystem): never leave the graph to the collector
    return FomResult(hist, iters, flags)


# ----------------------------------------------------------------------- POD-RBF
class RbfClosure:
    """Scaled RBF closure q_s = unscale(k(|x - x_i|) @ W), x = scale(q_p), and its full-chain Jacobian, batched
    over samples (FEM/fem_burgers.py:160-260).  bg_rbf_eval produces the kernel values phi (B, Ns) and the
    gradient factors d phi / d q_p already transposed, GT (B, n, Ns), in one pass; value and Jacobian are then
    ONE GEMM each against the output-scaled weights: q_s = phi Wd + (dy/2 + y_min), (dq_s/dq_p)^T = GT Wd."""

    def __init__(self, X_train, W, eps, kernel, x_min, x_max, y_min, y_max, device):
        if kernel not in ("gaussian", "imq"):
            raise ValueError("kernel must be 'gaussian' or 'imq'.")
        f = lambda a: _as_dev(np.asarray(a, dtype=np.float64), device)
        self.L = _lib.load()
        self.device = device
        Xt, Wm = f(X_train), f(W)
        self.eps = float(eps)
        self.kind = _lib.BG_RBF_GAUSSIAN if kernel == "gaussian" else _lib.BG_RBF_IMQ
        self.x_min, y_min = f(x_min), f(y_min)
        self.dx = f(x_max) - self.x_min
        self.dx[self.dx < 1e-15] = 1.0
        dy = f(y_max) - y_min
        dy[dy < 1e-15] = 1.0
        if Wm.shape != (Xt.shape[0], dy.numel()) or Xt.shape[1] != self.dx.numel():
            raise ValueError("X_train must be (Ns, n) and W (Ns, nbar)")
        self.Ns, self.n = Xt.shape
        self.XtT = Xt.t().contiguous()                                 # (n, Ns): centre index fastest
        self.Wd = (Wm * (0.5 * dy)).contiguous()                       # output scaling folded into the weights
        self.bias = (0.5 * dy + y_min).contiguous()
        self._buf = {}

    def _eval(self, qp, want_gt):
        B = qp.shape[0]
        if B not in self._buf:
            f64 = dict(dtype=torch.float64, device=self.device)
            self._buf[B] = (torch.empty((B, self.Ns), **f64), torch.empty((B, self.n, self.Ns), **f64))
        phi, GT = self._buf[B]
        qp = qp.contiguous()
        with torch.cuda.device(self.device):
            rc = self.L.bg_rbf_eval(B, self.n, self.Ns, self.kind, self.eps, _lib.ptr(qp), _lib.ptr(self.x_min),
                                    _lib.ptr(self.dx), _lib.ptr(self.XtT), _lib.ptr(phi),
                                    _lib.ptr(GT) if want_gt else None, _lib.stream_ptr(self.device))
        _lib.check(rc, "bg_rbf_eval")
        return phi, GT

    def value(self, qp):
        phi, _ = self._eval(qp, False)
        return torch.addmm(self.bias, phi, self.Wd)                    # (B, nbar)

    def jacobian_t(self, qp):
        _, GT = self._eval(qp, True)
        return torch.matmul(GT, self.Wd)                               # (B, n, nbar): one GEMM over B n rows

    def jacobian(self, qp):
        return self.jacobian_t(qp).transpose(1, 2)                     # (B, nbar, n) view


def pod_rbf_run(X, u0, mu1, mu2, dt, nsteps, U_p, U_s, X_train, W, epsilon, x_min, x_max, y_min, y_max,
                projection="LSPG", kernel="gaussian", E=0.0, tol_newton=1e-6, max_newton=30, device=None):
    """Batched ``pod_rbf_prom`` (FEM/fem_burgers.py:1278-1398)."""
    p = projection.lower()
    if p not in PROJ:
        raise ValueError("projection must be 'LSPG' or 'Galerkin'.")
    proj = PROJ[p]
    c = _setup(X, u0, mu1, mu2, dt, E, device)
    rbf = RbfClosure(X_train, W, epsilon, kernel, x_min, x_max, y_min, y_max, c.device)
    Up, Us = _as_dev(U_p, c.device), _as_dev(U_s, c.device)
    n = Up.shape[1]
    UpT, UsT = Up.t().contiguous(), Us.t().contiguous()
    tangent = _ClosureTangent(Up, Us, c.B)
    hist, iters, flags = _alloc_hist(c, nsteps)
    Ar, br, _, G = _workspace(c, n)
    st = _IterState(c, n)
    U0 = c.u0.clone()
    q = torch.zeros((c.B, n), dtype=torch.float64, device=c.device)
    for nt in range(nsteps):
        _mass_rhs(c, U0, G)
        st.begin_step()
        while True:
            qp = (U0 @ Up).contiguous()                                     # q_p = U_p^T U0        (:1352)
            tangent.jt.copy_(rbf.jacobian_t(qp))
            rom_reduce(c, tangent.gemm(), U0, G, proj, True, st.active, Ar, br, None, colmajor=True)   # (:1361)
            left = st.solve_update(1, Ar, br, qp, q, tol_newton, max_newton)   # q_new = q_p + dq, err = |dq|/|q_new|
            act = st.active_before
            U1 = q @ UpT + rbf.value(q) @ UsT                               # (:1378-1381)
            U0 = torch.where(act[:, None], U1, U0).contiguous()
            if left == 0:
                break
        iters[:, nt] = st.k
        hist[:, nt + 1] = U0
    flags |= st.flags
    return FomResult(hist, iters, flags)


# --------------------------------------------------------------------- local POD
def local_prom_run(X, u0, mu1, mu2, dt, nsteps, centers, local_bases, U_global, num_global_modes,
                   projection="Galerkin", E=0.0, tol=1e-6, max_it=20, device=None):
    """Batched ``local_prom_burgers`` (FEM/fem_burgers.py:979-1079): each sample picks, once per time
    step, the local basis of the cluster whose centre is nearest to ``U_global[:, :m]^T u^n``
    (= ``kmeans.predict``), then iterates like ``pod_prom_burgers`` in that basis.  The bases are
    zero-padded to a common width and travel as per-sample W; padded reduced unknowns get a unit
    diagonal so that their correction is exactly zero."""
    if projection not in ("Galerkin", "LSPG"):
        raise ValueError(f"Projection method '{projection}' is not available. Please use 'Galerkin' or 'LSPG'.")
    proj = PROJ[projection.lower()]
    c = _setup(X, u0, mu1, mu2, dt, E, device)
    ids = sorted(local_bases.keys())
    widths = [int(np.shape(local_bases[i])[1]) for i in ids]
    rmax = max(widths)          # beyond 47 modes rom_reduce / solve_update take their library paths
    stack = torch.zeros((len(ids), c.N, rmax), dtype=torch.float64, device=c.device)
    for s_, (i, w_) in enumerate(zip(ids, widths)):
        stack[s_, :, :w_] = _as_dev(local_bases[i], c.device)
    slot_of = torch.full((max(ids) + 1,), -1, dtype=torch.long, device=c.device)
    slot_of[torch.as_tensor(ids, device=c.device)] = torch.arange(len(ids), device=c.device)
    width_t = torch.as_tensor(widths, device=c.device)
    cen = _as_dev(centers, c.device)                                        # (n_clusters, m)
    Ug = _as_dev(U_global, c.device)[:, :num_global_modes].contiguous()
    col = torch.arange(rmax, device=c.device)
    stackT = stack.transpose(1, 2).contiguous()                              # (C, rmax, N)
    rows = torch.arange(c.B, device=c.device)
    hist, iters, flags = _alloc_hist(c, nsteps)
    Ar, br, wtu, G = _workspace(c, rmax)
    st = _IterState(c, rmax)
    q = torch.zeros((c.B, rmax), dtype=torch.float64, device=c.device)
    U0 = c.u0.clone()
    for n in range(nsteps):
        _mass_rhs(c, U0, G)
        qg = U0 @ Ug                                                          # (:1011)
        cid = torch.cdist(qg, cen, compute_mode="donot_use_mm_for_euclid_dist").argmin(dim=1)   # kmeans.predict (:1012)
        slot = slot_of[cid]
        if bool((slot < 0).any()):
            raise KeyError("a predicted cluster has no local basis")
        slot32 = slot.to(torch.int32)                                         # basis of each sample (:1013)
        pad = (col[None, :] >= width_t[slot][:, None]).to(torch.float64)      # 1 on padded reduced unknowns
        st.begin_step()
        while True:
            rom_reduce(c, stack, U0, G, proj, True, st.active, Ar, br, wtu, w_index=slot32)
            Ar.diagonal(dim1=1, dim2=2).add_(pad * st.active[:, None].to(torch.float64))
            left = st.solve_update(1, Ar, br, wtu, q, tol, max_it)           # q = Phi^T U0 + dq
            act = st.active_before
            # U1 = Phi q: one GEMM per cluster over the whole batch against the SHARED bases, then each sample
            # picks its cluster's row (a bmm against the gathered per-sample copies streams B N r doubles)
            U1 = torch.matmul(q, stackT)[slot, rows]                         # (C, B, N) -> (B, N)
            U0 = torch.where(act[:, None], U1, U0).contiguous()
            if left == 0:
                break
        iters[:, n] = st.k
        hist[:, n + 1] = U0
    flags |= st.flags
    return FomResult(hist, iters, flags)
