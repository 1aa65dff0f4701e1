#!/usr/bin/env python3
"""First evaluations of freshly created polarizable contexts from several host threads at once (the reference's path-integral loop calls
energy() from P OpenMP threads): every context's first result must equal its own second evaluation bit for bit.
usage: python tools/first_eval_stress.py [threads] [rounds]      (MPMC_ENERGY_LIB selects another build)"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import util
from mpmcxx_amd import energy

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 25
atoms, basis, opts = util.load_fixture("ion1000_polar")
energy.lib()
bad = []
lock = threading.Lock()


def worker(t):
    rng = np.random.default_rng(t)
    for r in range(rounds):
        a = dict(atoms)
        a["pos"] = atoms["pos"] + rng.normal(scale=0.02, size=atoms["pos"].shape)
        S = energy.System(a, basis, opts)
        e1 = S.energy(); p1 = S.observables["polarization_energy"]
        e2 = S.energy(); p2 = S.observables["polarization_energy"]
        if e1 != e2 or p1 != p2:
            with lock:
                bad.append((t, r, p1, p2))
        S.close()


ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print(f"{threads} threads x {rounds} fresh contexts: {len(bad)} first evaluations differ from the second", bad[:3])
sys.exit(1 if bad else 0)
