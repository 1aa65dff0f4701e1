#!/usr/bin/env python3
"""Fuzz of the HIP path against the oracle: random systems (1-450 atoms, molecules of 1-4 sites, frozen / chargeless / non-polarizable /
sigma <= 0 sites, cubic / orthorhombic / triclinic cells, positions outside the cell) x random options (LJ, Ewald, Thole fixed /
precision / Gauss-Seidel, Wolf, Feynman-Hibbs, solver), the acceptance criteria of tests/test_gpu_random.py.
usage: python tools/fuzz.py [first_seed] [count]      prints one line per failure and a summary"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_random as T
if "fused" in sys.argv:  # the dipole update riding the panel launch (fused_update = 1; round 5, off by default): last-arriving workgroup per tile
    sys.argv.remove("fused")
    from mpmcxx_amd import energy as _E4
    _E4.configure("fused_update", 1)
if "split" in sys.argv:  # ... and its two-waves-per-tile-pair form (pair_split = 1; off by default since round 4)
    sys.argv.remove("split")
    from mpmcxx_amd import energy as _E2
    _E2.configure("pair_split", 1)
if "sweep" in sys.argv:  # force the fast pair sweep (kernels_pair.hip) onto these small tables, where the default is k_pair_fused
    sys.argv.remove("sweep")
    from mpmcxx_amd import energy as _E
    _E.configure("pair_kernel", 2)

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(50000 + seed)
    n = int(rng.choice([1, 2, 5, 30, 64, 65, 100, 128, 191, 192, 193, 256, 300, 450]))
    cell = str(rng.choice(["cubic", "ortho", "triclinic"]))
    atoms, basis = T.random_system(rng, n, cell)
    opts = T.random_options(rng)
    r = rng.random()
    if opts["polarization"] and r < 0.2:
        opts.update(polar_gs=1, polar_max_iter=int(rng.integers(1, 5)))
    elif r < 0.3 and not opts["polarization"]:
        opts.update(wolf=1)
    elif r < 0.4:
        opts.update(feynman_hibbs=1, feynman_hibbs_order=int(rng.choice([2, 4])), temperature=float(rng.uniform(20, 150)))
    if opts["polarization"] and not opts.get("polar_gs"):
        opts["solver"] = str(rng.choice(["auto", "compact", "matrix_free", "dense"]))
    try:
        T.check(atoms, basis, opts, f"seed {seed} n {n} {cell}", wolf=bool(opts.get("wolf")))
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"FAIL seed {seed} n {n} {cell} {opts}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    if (seed - first + 1) % 25 == 0:
        print(f"  ... {seed - first + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {count} cases from seed {first}: {bad} failures")
sys.exit(1 if bad else 0)
