#!/usr/bin/env python3
"""Latency of ONE evaluation of the small fixtures (launch-bound regime): ms per mpmc_energy() call, back to back.
usage: python tools/small_rates.py [name ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

for name in (sys.argv[1:] or ["lj64", "lj1000", "ion64_es", "ion216_polar", "ion216_precision", "water64_polar", "ion1000_polar"]):
    atoms, basis, opts = util.load_fixture(name)
    S = energy.System(atoms, basis, opts)
    e = S.energy()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        S.energy()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:18s} n={S.n:5d}: {dt * 1e6:8.1f} us per evaluation  E = {e:.12e}  iterations {S.observables['polar_iterations']}", flush=True)
    S.close()
