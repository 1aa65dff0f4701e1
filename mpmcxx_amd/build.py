"""Build the HIP extension IN-TREE: mpmcxx_amd/libmpmc_energy.so (gfx950 only).

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  -ffp-contract=off is part of the numerical
contract of the kernels (csrc/pair_math.h): the minimum-image distance must round like the reference.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpmc_energy.so")
SOURCES = ["kernels.hip", "kernels_sym.hip", "kernels_delta.hip", "kernels_gs.hip", "kernels_dense.hip", "context.cpp", "evaluate.cpp", "trial.cpp", "pi.cpp"]
HEADERS = ["kernels.h", "context.h", "pair_math.h", "erfcx_coeffs.h", "device_math.h", os.path.join("..", "..", "include", "mpmc_energy.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


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
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
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
    print(build_library(force=True, verbose=True))
