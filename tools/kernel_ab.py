#!/usr/bin/env python3
"""Same-process A/B of kernel variants on ONE bead of the benchmark box (10 000 polarizable atoms).

usage: python tools/kernel_ab.py "label:key=v,key=v" ...        (an empty list = defaults; key=v: energy.configure(key, v), i.e.
       mpmc_debug_configure -- e.g. "fused:pair_kernel=1" "sweep:pair_kernel=2"); every variant runs on ONE stream (side_stream=0)
       unless it says otherwise, so that the HIP-event brackets see each kernel alone on the GPU

Each variant gets a fresh context created under its settings, 2 warm-up evaluations, then `reps` profiled evaluations; prints HIP-event ms per launch of every
kernel class, the evaluation wall time and the total energy (must be identical to ~1e-12 between variants).
The list is run twice (A B A B) so that drift of the box shows up.
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

reps = int(os.environ.get("KAB_REPS", "5"))
natoms = int(os.environ.get("KAB_NATOMS", "10000"))
if os.environ.get("KAB_WATER") == "1":  # natoms / 3 rigid three-site polarizable molecules (sigma- and epsilon-less, partly non-polarizable H sites)
    from mpmcxx_amd import gen_box, pqr

    wd = tempfile.mkdtemp()
    nm = natoms // 3
    L = 14.0 * (nm / 64.0) ** (1.0 / 3.0)
    gen_box.write_pqr(os.path.join(wd, "w.pqr"), gen_box.molecular_box(nm, L, 5, extra_neutral=False))
    gen_box.write_input(os.path.join(wd, "w.in"), "w.pqr", gen_box.cubic(L), dict(gen_box.POLAR_OPTS))
    atoms, basis, opts = pqr.load_case(os.path.join(wd, "w.in"))
else:
    atoms, basis, opts = bench.build_case(natoms, tempfile.mkdtemp())
specs = sys.argv[1:] or ["default:"]
for rnd in (1, 2):
    for spec in specs:
        label, _, envs = spec.partition(":")
        S = energy.System(atoms, basis, opts)
        S.configure("side_stream", 0)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            S.configure(k.lstrip("@"), float(v))
        for _ in range(2):
            e = S.energy()
        S.set_profiling(True)
        S.timings(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            e = S.energy()
        wall = (time.perf_counter() - t0) / reps
        t = S.timings(reset=True)
        S.close()
        cls = "  ".join(f"{k} {v['ms'] / max(v['launches'], 1):.4f}x{v['launches'] // reps}" for k, v in t.items() if v["launches"])
        print(f"r{rnd} {label:>12s}: eval {wall * 1e3:.3f} ms  E {e:.12e} | {cls}", flush=True)
