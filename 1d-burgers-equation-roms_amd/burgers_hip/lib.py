"""ctypes binding of libburgers_hip.so (the C ABI declared in include/burgers_hip.h).

There is no CPU fallback: if the shared object is missing, or a call is made without a
HIP device, this module raises.  torch is imported first so that the process uses one
HIP runtime (torch's) for device memory, streams and our kernels alike.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch  # noqa: F401  (must precede the CDLL load: one libamdhip64 per process)

from . import build as _build

c_double_p = ctypes.c_void_p      # raw device pointers travel as integers
c_int_p = ctypes.c_void_p

BG_OK = 0
BG_ERR_BAD_ARG = -1
BG_ERR_UNSUPPORTED_N = -2
BG_ERR_NONUNIFORM = -3
BG_ERR_LAUNCH = -4
BG_ERR_UNSUPPORTED_R = -5
BG_ERR_PROJECTION = -6
BG_ERR_WORKSPACE = -7
BG_PROJ_GALERKIN, BG_PROJ_LSPG = 0, 1
BG_FLAG_HIT_CAP, BG_FLAG_NONFINITE = 1, 2
BG_OPT_SUPG, BG_OPT_NONUNIFORM, BG_OPT_W_COLMAJOR, BG_OPT_MFMA_16X16, BG_OPT_FORCE_PIVOTED, BG_OPT_NO_TANGENT_REUSE = 1, 2, 4, 8, 16, 32
BG_OPT_FOM_WIDE, BG_OPT_FOM_WAVE = 64, 128
BG_ACT_NONE, BG_ACT_ELU, BG_ACT_RELU, BG_ACT_TANH = 0, 1, 2, 3
BG_COUNTER_SLOTS, BG_COUNTER_STRIDE = 16, 32
BG_RBF_GAUSSIAN, BG_RBF_IMQ = 0, 1
BG_INFO_NEEDS_PIVOTING = -1

_SIGNATURES = {
    # name: (restype, argtypes)
    "bg_abi_version": (ctypes.c_int, []),
    "bg_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "bg_last_hip_error": (ctypes.c_int, []),
    "bg_fom_max_n": (ctypes.c_int, []),
    "bg_fom_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                  c_double_p, c_double_p, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, ctypes.c_int, ctypes.c_int, c_double_p, c_int_p,
                                  c_int_p, ctypes.c_void_p]),
    "bg_fom_run_traced": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                         c_double_p, c_double_p, ctypes.c_double, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, c_double_p, c_int_p,
                                         c_int_p, c_double_p, ctypes.c_void_p]),
    "bg_fom_assemble": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                       c_double_p, c_double_p, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p,
                                       ctypes.c_void_p]),
    "bg_tridiag_solve": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                        c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_transpose_batched": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                            c_double_p, ctypes.c_void_p]),
    "bg_fd_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                 c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_double_p, c_int_p,
                                 c_int_p, ctypes.c_void_p]),
    "bg_rom_max_n": (ctypes.c_int, []),
    "bg_rom_max_r": (ctypes.c_int, []),
    "bg_forcing_setup": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_double,
                                        ctypes.c_int, c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_mass_rhs": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                   c_double_p, ctypes.c_void_p]),
    "bg_rom_reduce": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                     c_double_p, ctypes.c_longlong, c_double_p, c_double_p, c_double_p,
                                     c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_int_p,
                                     c_double_p, c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_rom_reduce_indexed": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                             c_double_p, ctypes.c_longlong, c_int_p, c_double_p, c_double_p, c_double_p,
                                             c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_int_p,
                                             c_double_p, c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_rom_reduce_lifted": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                            c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                            c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_int_p,
                                            c_double_p, c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_rom_run_max_r": (ctypes.c_int, []),
    "bg_rom_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                  c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_int, ctypes.c_int, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, ctypes.c_void_p]),
    "bg_quad_rom_max_n": (ctypes.c_int, []),
    "bg_quad_rom_h3f_elems": (ctypes.c_longlong, [ctypes.c_int]),
    "bg_quad_rom_phif_elems": (ctypes.c_longlong, [ctypes.c_int]),
    "bg_quad_rom_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                       c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_int, ctypes.c_int, c_double_p, c_int_p, c_int_p, c_int_p,
                                       c_int_p, ctypes.c_void_p]),
    "bg_rom_run_wide_max_r": (ctypes.c_int, []),
    "bg_rom_run_wide_phi_elems": (ctypes.c_longlong, [ctypes.c_int]),
    "bg_rom_run_wide": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                                       c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_int, ctypes.c_int, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, ctypes.c_void_p]),
    "bg_rom_lift": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                   c_int_p, c_double_p, ctypes.c_void_p]),
    "bg_quad_features": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_int_p, c_int_p, c_double_p, ctypes.c_void_p]),
    "bg_rom_frag_elems": (ctypes.c_longlong, [ctypes.c_int, ctypes.c_int]),
    "bg_rom_frag_pad": (ctypes.c_int, [ctypes.c_int]),
    "bg_quad_tangent": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_double_p,
                                       c_int_p, c_double_p, ctypes.c_void_p]),
    "bg_rom_reduce_frag": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_double_p,
                                          c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_int, c_int_p, c_double_p,
                                          c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_lu_solve_update": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int,
                                          c_double_p, c_double_p, c_double_p, ctypes.c_double, ctypes.c_int,
                                          c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, ctypes.c_void_p]),
    "bg_lu_solve": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_double,
                                   c_int_p, c_double_p, c_int_p, ctypes.c_void_p]),
    "bg_jacobi_sweep": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, c_int_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double, c_int_p, ctypes.c_void_p]),
    "bg_rbf_eval": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p,
                                   c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_void_p]),
    "bg_mlp_act_jvp": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "bg_decode_modes_bf16": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_void_p, c_double_p, ctypes.c_void_p]),
    # win / wout / acts (int[]), W / bias (void*[]) and alphas (float[]) are HOST arrays built by the caller
    "bg_decode_mlp_bf16": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          c_double_p, c_double_p, c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_float), c_double_p, ctypes.c_void_p]),
    "bg_ann_rom_limits": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)] * 4),
    # widths / acts (int[]), wt / bias (void*[]) and alphas (float[]) are HOST arrays built by the caller
    "bg_ann_rom_run": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_float), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_int, ctypes.c_int, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, ctypes.c_void_p]),
}

_lib = None


class BurgersHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = _lib.bg_strerror(code).decode() if _lib is not None else str(code)
        extra = f" (hipError {_lib.bg_last_hip_error()})" if code == BG_ERR_LAUNCH and _lib else ""
        super().__init__(f"{where}: {msg}{extra} [code {code}]")


def library_path():
    # BG_LIB_PATH: load an experimental build of the same ABI (kernel A/B timing); default in-tree
    return os.environ.get("BG_LIB_PATH") or _build.LIB_PATH


def load(build_if_missing=False):
    """Load the shared object and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        if build_if_missing:
            _build.build_library()
        else:
            raise ImportError(
                f"{path} is missing: build it with `python __graft_entry__.py build` "
                "(there is no CPU fallback for the HIP path)")
    L = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if L.bg_abi_version() != 1:
        raise ImportError(f"ABI mismatch: library reports {L.bg_abi_version()}, binding expects 1")
    _lib = L
    return L


def declared_symbols():
    return sorted(_SIGNATURES)


def check(code, where):
    if code != BG_OK:
        raise BurgersHipError(code, where)


def require_device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the Burgers HIP kernels need an MI355X (no CPU fallback)")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:      # 'cuda' -> 'cuda:<current>': tensors report an indexed device
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def ptr(t):
    return ctypes.c_void_p(None if t is None else t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


PINNED_MIN_BYTES = 1 << 20


def to_host(t):
    """Device tensor -> numpy array.  Results of a sweep are gigabytes, and a copy into fresh
    pageable memory runs far below the PCIe rate, so anything over 1 MiB goes through torch's
    caching pinned-host allocator: the returned ndarray is a view of the pinned block, which
    returns to that cache when the array is dropped.  BG_PINNED_RESULTS=0 keeps pageable memory."""
    t = t.detach()
    if (t.is_cuda and t.numel() * t.element_size() >= PINNED_MIN_BYTES
            and os.environ.get("BG_PINNED_RESULTS", "1") != "0"):
        try:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        except RuntimeError:            # pinning refused (ulimit, fragmentation): pageable copy
            return t.cpu().numpy()
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host.numpy()
    return t.cpu().numpy()


def mesh_options(X, supg=True):
    """Option bits for the assembly-carrying entry points: SUPG on/off, uniform/non-uniform mesh."""
    return (BG_OPT_SUPG if supg else 0) | (0 if mesh_is_uniform(X) else BG_OPT_NONUNIFORM)


def mesh_is_uniform(X, rtol=1e-9):
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 1 or len(X) < 2:
        return False
    d = np.diff(X)
    h = (X[-1] - X[0]) / (len(X) - 1)
    return bool(h > 0 and np.all(np.abs(d - h) <= rtol * abs(h)))
