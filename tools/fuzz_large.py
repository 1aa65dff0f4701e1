#!/usr/bin/env python3
"""tools/fuzz.py at sizes where the spatial order, the tile-pair classes (beyond cutoff / beyond the damping range) and the uniform-image
fast path are all active: 700-2600 atoms, random cells and options.  usage: python tools/fuzz_large.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_gpu_random as T
if "grid" in sys.argv:  # the aligned-grid spatial order of large tables (round 4) forced onto these sizes: 4 x 2 columns, serpentine
    sys.argv.remove("grid")
    from mpmcxx_amd import energy as _E3
    _E3.configure("sort_nx", 4)
    _E3.configure("sort_ny", 2)
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
count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(60000 + seed)
    n = int(rng.choice([700, 1100, 1600, 2600]))
    cell = str(rng.choice(["cubic", "ortho", "ortho", "triclinic"]))
    atoms, basis = T.random_system(rng, n, cell)
    if rng.random() < 0.5:  # a stretched cell: many tile pairs beyond the cutoff (half the SHORTEST lattice vector)
        basis = basis * np.array([1.0, 1.0, 2.2])[:, None] if cell != "triclinic" else basis
        atoms["pos"] = atoms["pos"] * (np.array([1.0, 1.0, 2.2]) if cell != "triclinic" else 1.0)
    opts = T.random_options(rng)
    if opts["polarization"]:
        opts.update(polar_precision=0.0, polar_max_iter=int(rng.integers(1, 5)), solver=str(rng.choice(["auto", "compact", "matrix_free"])))
    try:
        T.check(atoms, basis, opts, f"seed {seed} n {n} {cell}")
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"FAIL seed {seed} n {n} {cell} {opts}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"  ... {seed - first + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz_large: {count} cases from seed {first}: {bad} failures")
sys.exit(1 if bad else 0)
