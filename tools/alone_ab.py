#!/usr/bin/env python3
"""One system at a time (what System::mc sees): wall time per evaluation of BASELINE configs[2] / configs[3] under a list of
mpmc_debug_configure variants, interleaved three times.  usage: python tools/alone_ab.py "label:key=v,key=v" ..."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpmcxx_amd import energy, gen_box, pqr  # noqa: E402

specs = sys.argv[1:] or ["default:"]
wd = tempfile.mkdtemp()
for name, reps in (("ion10k_es", 300), ("ion10k_polar", 60)):
    inp, _ = gen_box.materialize(name, wd)
    atoms, basis, opts = pqr.load_case(inp)
    systems = {}
    for spec in specs:
        label, _, envs = spec.partition(":")
        kvs = [kv.partition("=") for kv in filter(None, envs.split(","))]
        at_create = {}  # switches that are read when the context (its side stream) is made: process-wide default around the constructor
        for k, _, v in kvs:
            if k in at_create:
                energy.configure(k, float(v))
        S = energy.System(atoms, basis, opts)
        for k, dflt in at_create.items():
            energy.configure(k, dflt)
        for k, _, v in kvs:
            if k not in at_create:
                S.configure(k, float(v))
        S.energy()
        S.energy()
        systems[label] = S
    for rnd in range(3):
        row = []
        for label, S in systems.items():
            t0 = time.perf_counter()
            for _ in range(reps):
                S.energy()
            row.append(f"{label} {(time.perf_counter() - t0) / reps * 1e6:.1f} us")
        print(f"{name} r{rnd}: " + "   ".join(row), flush=True)
    for S in systems.values():
        S.close()
