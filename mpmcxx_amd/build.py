"""Build the HIP extension IN-TREE: mpmcxx_amd/libmpmc_energy.so (gfx950 only).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract of the kernels (csrc/pair_math.h): the minimum-image distance must round like the reference.

Every source is compiled to its own object under mpmcxx_amd/.obj/ (in parallel, rebuilt only when it or a header
changed), then linked: a one-kernel edit costs one translation unit, not the whole library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, ".obj")
LIB = os.path.join(HERE, "libmpmc_energy.so")
SOURCES = ["kernels.hip", "kernels_sym.hip", "kernels_panel.hip", "kernels_pair.hip", "erfc_table.cpp", "kernels_delta.hip", "kernels_gs.hip", "kernels_dense.hip", "context.cpp", "evaluate.cpp",
           "trial.cpp", "pi.cpp", "comm.cpp", "gibbs.cpp"]
HEADERS = ["kernels.h", "context.h", "pair_math.h", "erfcx_coeffs.h", "device_math.h", "erfc_table.inc", os.path.join("..", "..", "include", "mpmc_energy.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-ldl", "-lpthread"]
FLAGS = CFLAGS + LDFLAGS  # (kept for tools that print the build line)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def _deps():
    return [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, s) for s in SOURCES] + _deps())


def _compile(cc: str, src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in [path] + _deps()):
            return obj
    tmp = f"{obj}.tmp.{os.getpid()}"
    cmd = [cc] + CFLAGS + ["-x", "hip", "-c", path, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    # several ranks of one job may arrive here at once (torch.distributed.run starts N processes): one of them builds, into a
    # temporary file that is renamed over the library when complete; the others wait on the lock and find the library fresh
    import fcntl

    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return LIB
            cc = hipcc()
            os.makedirs(OBJ, exist_ok=True)
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                objs = list(ex.map(lambda s: _compile(cc, s, force, verbose), SOURCES))
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [cc] + objs + LDFLAGS + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    import sys

    print(build_library(force="--force" in sys.argv, verbose=True))
