#!/usr/bin/env python3
"""Monte Carlo trial-move rate: single-molecule displacements through mpmc_trial_* versus full evaluations.
usage: python tools/trial_bench.py [natoms] [n_trials]   (non-polarizable LJ + Ewald box of the bench generator)"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
atoms, basis, opts = bench.build_case(n, tempfile.mkdtemp())
opts = dict(opts)
opts.update(polarization=0, polar_iterative=0)
S = energy.System(atoms, basis, opts)
e = S.energy()
t0 = time.perf_counter()
for _ in range(5):
    S.energy()
t_full = (time.perf_counter() - t0) / 5
rng = np.random.default_rng(1)
pos = atoms["pos"].copy()
acc = 0
t0 = time.perf_counter()
for _ in range(trials):
    i = int(rng.integers(n))
    trial = pos[i:i + 1] + rng.normal(scale=0.15, size=(1, 3))
    et = S.trial_energy(i, trial)
    if et < e or rng.random() < np.exp(-(et - e) / 300.0):
        S.accept()
        pos[i] = trial[0]
        e = et
        acc += 1
    else:
        S.reject()
t_trial = (time.perf_counter() - t0) / trials
e_full = S.energy()
print(f"natoms {n}: full evaluation {t_full * 1e3:.3f} ms ({1 / t_full:.0f}/s); trial move {t_trial * 1e6:.1f} us ({1 / t_trial:.0f}/s), "
      f"{acc}/{trials} accepted; drift after {acc} accepted moves: {abs(e - e_full) / abs(e_full):.2e} relative")
S.close()
