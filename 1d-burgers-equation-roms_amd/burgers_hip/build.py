"""In-tree build of libburgers_hip.so (hipcc, gfx950 only).

The library is built next to this package (``1d-burgers-equation-roms_amd/libburgers_hip.so``)
so that it travels with the source tree; nothing is installed or JIT-cached elsewhere.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.path.join(PKG_ROOT, "libburgers_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    repo = os.path.dirname(PKG_ROOT)
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
        sorted(glob.glob(os.path.join(repo, "include", "*.h")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libburgers_hip.so cannot be built")
    return exe


def build_library(force=False, extra_flags=(), verbose=False):
    """Compile every csrc/*.hip into one shared object for gfx950."""
    if not force and not is_stale():
        return LIB_PATH
    objs = []
    tmpdir = os.path.join(PKG_ROOT, "build")
    os.makedirs(tmpdir, exist_ok=True)
    common = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
              "-Wall", "-Wno-unused-function", *extra_flags]
    procs = []
    for src in sources():
        obj = os.path.join(tmpdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = common + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
