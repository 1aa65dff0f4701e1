#!/usr/bin/env python3
"""Where the HOST spends a path-integral step of the benchmark (32 beads of the 10 000-atom polarizable box on one GPU): seconds in the
enqueue loop (mpmc_energy_async per bead) and in the waits (mpmc_energy_wait per bead), per step.  usage: python tools/host_step_profile.py [beads] [steps]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
beads = [energy.System(dict(atoms, pos=bench.bead_positions(atoms["pos"], b)), basis, opts) for b in range(P)]
for _ in range(3):
    energy.pi_potential_local(beads)
t_enq = t_wait = 0.0
first_wait = 0.0
per_enq = []
t_all = time.perf_counter()
for _ in range(steps):
    t0 = time.perf_counter()
    for s in beads:
        ta = time.perf_counter()
        s.energy_async()
        per_enq.append(time.perf_counter() - ta)
    t1 = time.perf_counter()
    for k, s in enumerate(beads):
        tb = time.perf_counter()
        s.energy_wait()
        if k == 0:
            first_wait += time.perf_counter() - tb
    t2 = time.perf_counter()
    t_enq += t1 - t0
    t_wait += t2 - t1
wall = time.perf_counter() - t_all
per_enq.sort()
print(f"{P} beads, {steps} steps: {wall / steps * 1e3:.2f} ms per step ({P * steps / wall:.1f} evals/s); enqueue loop {t_enq / steps * 1e3:.2f} ms "
      f"({t_enq / steps / P * 1e6:.0f} us per bead; median {per_enq[len(per_enq) // 2] * 1e6:.0f}, max {per_enq[-1] * 1e6:.0f}), waits {t_wait / steps * 1e3:.2f} ms "
      f"(first bead {first_wait / steps * 1e3:.2f} ms)")
c = energy.lib()
for s in beads:
    s.close()
