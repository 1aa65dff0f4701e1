#!/usr/bin/env python3
"""Where four waves per tile pair stop paying in the pair sweep: full evaluations of the polarizable ion box of bench.py at several sizes with
MPMC_PAIR_WAVES=1 and =4 (the switch is read when a context is created), same process, same box; energies of the two forms compared."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [216, 512, 1000, 2000, 3000, 4000, 5000, 7000, 10000]
tmp = tempfile.mkdtemp()
print("atoms  tile_pairs   W=1 us/eval   W=4 us/eval   rel.diff of the energies")
for n in sizes:
    atoms, basis, opts = bench.build_case(n, tmp)
    res = {}
    for w in (1, 4):
        os.environ["MPMC_PAIR_WAVES"] = str(w)
        S = energy.System(atoms, basis, opts)
        for _ in range(5):
            e = S.energy()
        reps = 40 if n <= 4000 else 15
        t = time.perf_counter()
        for _ in range(reps):
            S.energy()
        res[w] = ((time.perf_counter() - t) / reps * 1e6, e)
        S.close()
    nt = (len(atoms["pos"]) + 63) // 64
    print(f"{len(atoms['pos']):5d}  {nt * (nt + 1) // 2:9d}   {res[1][0]:10.1f}   {res[4][0]:10.1f}   {abs(res[1][1] - res[4][1]) / abs(res[1][1]):.2e}", flush=True)
