#!/usr/bin/env python3
"""Energy evaluations per second of BASELINE configs[1..3] on one GPU: one system evaluated back to back,
and 32 copies with jittered positions in flight (energy.pi_potential_local).  usage: python tools/config_rates.py"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mpmcxx_amd import gen_box  # noqa: E402
from mpmcxx_amd import energy, pqr  # noqa: E402

wd = tempfile.mkdtemp()
for label, name in (("configs[1]  1 000-atom LJ box (rd_only)", "lj1000"), ("configs[2] 10 000-atom LJ + Ewald box", "ion10k_es"),
                    ("configs[3] 10 000-atom LJ + Ewald + Thole box", "ion10k_polar")):
    inp, _ = gen_box.materialize(name, wd)
    atoms, basis, opts = pqr.load_case(inp)
    S = energy.System(atoms, basis, opts)
    e = S.energy()
    reps = 400 if name == "lj1000" else (100 if name == "ion10k_es" else 50)
    t0 = time.perf_counter()
    for _ in range(reps):
        S.energy()
    one = (time.perf_counter() - t0) / reps
    S.close()
    beads = []
    for b in range(32):
        a = dict(atoms)
        a["pos"] = atoms["pos"] + np.random.default_rng([17, b]).normal(scale=0.05, size=atoms["pos"].shape)
        beads.append(energy.System(a, basis, opts))
    energy.pi_potential_local(beads)
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        energy.pi_potential_local(beads)
    many = (time.perf_counter() - t0) / (steps * 32)
    for s in beads:
        s.close()
    print(f"{label}: E = {e:.10e} K; one system at a time {one * 1e3:.3f} ms ({1 / one:.0f} evals/s); 32 systems in flight {many * 1e3:.3f} ms each "
          f"({1 / many:.0f} evals/s)", flush=True)
# configs[3] taken literally: the dense 3N x 3N matrix on the fp64 matrix cores (--solver dense), one system
inp, _ = gen_box.materialize("ion10k_polar", wd)
atoms, basis, opts = pqr.load_case(inp)
S = energy.System(atoms, basis, dict(opts, solver="dense"))
e = S.energy()
S.energy()
t0 = time.perf_counter()
for _ in range(5):
    S.energy()
one = (time.perf_counter() - t0) / 5
S.close()
print(f"configs[3] literal (dense 3N x 3N A matrix, MFMA contraction): E = {e:.10e} K; one system at a time {one * 1e3:.3f} ms ({1 / one:.0f} evals/s)", flush=True)
