#!/usr/bin/env python3
"""Fuzz of the context's state handling: one live context per random system goes through a random sequence of
  small position updates, partial and full bulk updates, trial moves (accepted / rejected), cell rescaling, molecule removal and
  insertion (growth past the capacity hint), option changes (polarization on / off, solver)
and after EVERY operation its energy must equal that of a fresh context built from the host-side truth (1e-10: same arithmetic, other
atom order), and at the end the oracle's (1e-9).
usage: python tools/fuzz_state.py [first_seed] [count] [threads]
threads > 1: the seeds are spread over that many host threads working at the same time, each on its own contexts (the oracle check at
the end of a sequence is left out there: the comparison with fresh contexts is the test of cross-context interference)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_random as T
import util
from mpmcxx_amd import energy
from oracle import OracleSystem

import threading

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bad = 0
done = 0
lock = threading.Lock()
t0 = time.time()


def mol_ranges(ids):
    starts = [0] + [i for i in range(1, len(ids)) if ids[i] != ids[i - 1]] + [len(ids)]
    return [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]


def run_seed(seed):
    global bad, done
    rng = np.random.default_rng(90000 + seed)
    n = int(rng.choice([40, 130, 300, 520, 700]))
    atoms, basis = T.random_system(rng, n, str(rng.choice(["cubic", "ortho", "triclinic"])))
    atoms = {k: np.array(v) for k, v in atoms.items()}
    opts = T.random_options(rng)
    opts.update(polar_precision=0.0, polar_gs=0)
    log = []
    try:
        S = energy.System(atoms, basis, opts, max_atoms=len(atoms["charge"]))
        S.energy()
        for step in range(10):
            op = str(rng.choice(["small", "bulk_part", "bulk_all", "trial", "cell", "remove", "insert", "options"]))
            mols = mol_ranges(atoms["mol_id"])
            if op == "small":
                a, b = mols[rng.integers(len(mols))]
                atoms["pos"][a:b] += rng.normal(scale=0.2, size=(b - a, 3))
                S.update_positions(a, atoms["pos"][a:b])
            elif op == "bulk_part" and len(atoms["charge"]) > 300:
                a = int(rng.integers(0, len(atoms["charge"]) - 280)); b = a + 270
                atoms["pos"][a:b] += rng.normal(scale=0.05, size=(b - a, 3))
                S.update_positions(a, atoms["pos"][a:b])
            elif op == "bulk_all":
                atoms["pos"] += rng.normal(scale=float(rng.choice([0.03, 2.5])), size=atoms["pos"].shape)
                S.update_positions(0, atoms["pos"])
            elif op == "trial":
                S.energy()  # trial moves start from an evaluated, accepted configuration
                a, b = mols[rng.integers(len(mols))]
                trial = atoms["pos"][a:b] + rng.normal(scale=0.3, size=(b - a, 3))
                S.trial_energy(a, trial)
                if rng.random() < 0.5:
                    S.accept(); atoms["pos"][a:b] = trial
                else:
                    S.reject()
            elif op == "cell":
                s = float(rng.uniform(0.97, 1.04))
                basis = basis * s
                atoms["pos"] = atoms["pos"] * s
                S.set_box(basis)
                S.update_positions(0, atoms["pos"])
            elif op == "remove" and len(mols) > 3:
                a, b = mols[rng.integers(len(mols))]
                keep = np.ones(len(atoms["charge"]), bool); keep[a:b] = False
                atoms = {k: v[keep].copy() for k, v in atoms.items()}
                S.set_atoms(atoms)
            elif op == "insert":
                a, b = mols[rng.integers(len(mols))]
                add = {k: v[a:b].copy() for k, v in atoms.items()}
                add["pos"] = add["pos"] + rng.uniform(-0.5, 0.5, size=3) @ basis
                add["mol_id"] = np.full(b - a, atoms["mol_id"].max() + 1, dtype=np.int32)
                atoms = {k: np.concatenate([atoms[k], add[k]]) for k in atoms}
                S.set_atoms(atoms)
            elif op == "options":
                if opts["polarization"]:
                    opts.update(polarization=0, polar_iterative=0)
                else:
                    opts.update(polarization=1, polar_iterative=1, polar_ewald=int(rng.random() < 0.7), polar_damp=2.1304, polar_max_iter=int(rng.integers(1, 6)),
                                solver=str(rng.choice(["auto", "compact", "matrix_free", "dense"])))
                S.set_options(opts)
            else:
                continue
            log.append(op)
            e = S.energy()
            F = energy.System(atoms, basis, opts)
            ef = F.energy()
            ok = (not np.isfinite(ef) and not np.isfinite(e)) or abs(e - ef) <= 1e-10 * abs(ef) + 1e-9
            for k in ("rd_energy", "coulombic_energy", "polarization_energy"):
                ok = ok and (abs(S.observables[k] - F.observables[k]) <= 1e-10 * max(abs(F.observables[k]), 1e-3 * abs(ef)) + 1e-9 or not np.isfinite(ef))
            ok = ok and int(S.observables["n_lj_in_cutoff"]) == int(F.observables["n_lj_in_cutoff"])
            F.close()
            assert ok, (step, op, e, ef)
        if threads == 1:
            ref = OracleSystem(atoms, basis, opts).energy()
            if np.isfinite(ref["energy"]):
                assert util.close(S.energy(), ref["energy"]), ("final vs oracle", S.observables["energy"], ref["energy"])
        S.close()
    except Exception as e:  # noqa: BLE001
        with lock:
            bad += 1
        print(f"FAIL seed {seed} n {n} ops {log}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    with lock:
        done += 1
        if done % 10 == 0:
            print(f"  ... {done} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)


def worker(k):
    for seed in range(first + k, first + count, threads):
        run_seed(seed)


energy.lib()
ths = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
[t.start() for t in ths]
[t.join() for t in ths]
print(f"fuzz_state: {count} cases from seed {first} on {threads} thread(s): {bad} failures")
sys.exit(1 if bad else 0)
