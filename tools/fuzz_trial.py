#!/usr/bin/env python3
"""Fuzz of the trial-move (delta energy) path: random systems (LJ / LJ + Ewald / polarizable with every Thole option the random-system
generator draws, molecules of 1-4 sites, frozen and chargeless sites, all cell shapes; sizes up to 700 atoms so that tile classes, panels
and the touched-tile store update are active), random sequences of molecule moves with accept / reject; every trial energy against a
fresh context on the same configuration (1e-11; polarization energy 1e-10), the final accumulated totals against the oracle (1e-9).
usage: python tools/fuzz_trial.py [first_seed] [count] [polar]     ("polar": polarizable systems only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_random as T
import util
from mpmcxx_amd import energy
from oracle import OracleSystem
if "sweep" in sys.argv:  # force the fast pair sweep (kernels_pair.hip) onto these small tables, where the default is k_pair_fused
    sys.argv.remove("sweep")
    from mpmcxx_amd import energy as _E
    _E.configure("pair_kernel", 2)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
polar_only = len(sys.argv) > 3 and sys.argv[3] == "polar"
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(70000 + seed)
    n = int(rng.choice([2, 5, 40, 64, 65, 130, 200, 300, 450, 700]))
    cell = str(rng.choice(["cubic", "ortho", "triclinic"]))
    atoms, basis = T.random_system(rng, n, cell)
    opts = T.random_options(rng)
    if polar_only and not opts["polarization"]:
        continue
    if not polar_only:
        opts.update(polarization=0, polar_iterative=0, polar_ewald=0, rd_only=int(rng.random() < 0.3))
        r_ext = rng.random()  # adjacent physics of the delta kernels (round 3): Wolf electrostatics, Feynman-Hibbs corrections
        if r_ext < 0.2 and not opts["rd_only"]:
            opts.update(wolf=1)
        elif r_ext < 0.4:
            opts.update(feynman_hibbs=1, feynman_hibbs_order=int(rng.choice([2, 4])), temperature=float(rng.uniform(20, 150)))
    elif rng.random() < 0.5:
        opts["solver"] = str(rng.choice(["auto", "compact", "matrix_free"]))
    ids = atoms["mol_id"]
    starts = [0] + [i for i in range(1, len(ids)) if ids[i] != ids[i - 1]] + [len(ids)]
    mols = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]
    try:
        S = energy.System(atoms, basis, opts)
        e_acc = S.energy()
        if not np.isfinite(e_acc):
            S.close(); continue
        pos = atoms["pos"].copy()
        for step in range(8):
            a, b = mols[rng.integers(len(mols))]
            trial = pos[a:b] + rng.normal(scale=0.3, size=(b - a, 3)) + (basis[rng.integers(3)] if rng.random() < 0.1 else 0.0)
            e_trial = S.trial_energy(a, trial)
            full = pos.copy(); full[a:b] = trial
            F = energy.System(dict(atoms, pos=full), basis, opts)
            e_full = F.energy()
            if not (np.isfinite(e_full) and np.isfinite(e_trial)):
                F.close(); S.reject(); continue
            for k in ("energy", "rd_energy", "coulombic_energy", "es_real", "es_recip", "lj_pairs", "polarization_energy"):
                x, y = S.trial_observables[k], F.observables[k]
                tol = 1e-10 if k in ("polarization_energy", "energy") and opts["polarization"] else 1e-11
                assert abs(x - y) <= tol * max(abs(y), abs(F.observables["energy"]) * 1e-3) + 1e-9, (step, k, x, y)
            if opts["polarization"]:
                assert S.trial_observables["polar_iterations"] == F.observables["polar_iterations"], (step, "iterations")
            assert S.trial_observables["n_lj_in_cutoff"] == F.observables["n_lj_in_cutoff"], (step, "n_lj")
            if not opts["rd_only"]:
                assert S.trial_observables["n_es_in_cutoff"] == F.observables["n_es_in_cutoff"], (step, "n_es")
            if not opts["polarization"]:
                assert not S.last_trial_was_full(), (step, "a full evaluation instead of delta energies")
            F.close()
            if rng.random() < 0.5:
                S.accept(); pos = full; e_acc = e_trial
            else:
                S.reject()
        ref = OracleSystem(dict(atoms, pos=pos), basis, opts).energy()
        if np.isfinite(ref["energy"]):  # (the oracle's Wolf total carries no real / reciprocal split: the totals are compared)
            # (1e-9 relative; energies that are zero to rounding -- two atoms beyond every cutoff: 1e-29 against 1e-34 -- get an absolute floor)
            e_now = S.energy()
            assert abs(e_now - ref["energy"]) <= 1e-9 * abs(ref["energy"]) + 1e-15 and abs(e_acc - ref["energy"]) <= 1e-9 * abs(ref["energy"]) + 1e-15, ("final", e_acc, e_now, ref["energy"])
        S.close()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"FAIL seed {seed} n {n} {cell} rd_only {opts['rd_only']}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    if (seed - first + 1) % 20 == 0:
        print(f"  ... {seed - first + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_trial: {count} cases from seed {first}: {bad} failures")
sys.exit(1 if bad else 0)
