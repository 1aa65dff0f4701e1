#!/usr/bin/env python3
"""Two settings of a measurement switch (mpmc_debug_configure) over system sizes: full evaluations of the polarizable ion box of
bench.py, one at a time, same process, same box, the settings interleaved (three rounds); energies of the two forms compared.
   python tools/pair_waves_sweep.py [--key NAME=a,b] [sizes...]      default: pair_kernel=1,2 (k_pair_fused against the fast sweep);
   pair_waves=1,4 (with pair_kernel=1 as a second --key) one against four waves per tile pair; side_stream=0,1 places the size below
   which the side stream is not forked."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

args = sys.argv[1:]
name, values = "pair_kernel", ["1", "2"]
fixed = []
while args and args[0] == "--key":
    k, v = args[1].split("=")
    if "," in v:
        name, values = k, v.split(",")
    else:
        fixed.append((k, float(v)))
    args = args[2:]
for k, v in fixed:
    energy.configure(k, v)
sizes = [int(x) for x in args] or [216, 512, 1000, 2000, 3000, 4000, 5000, 7000, 10000]
tmp = tempfile.mkdtemp()
print(f"atoms  tile_pairs   {name}={values[0]} us/eval (3 rounds)      {name}={values[1]} us/eval (3 rounds)      rel.diff of the energies")
for n in sizes:
    atoms, basis, opts = bench.build_case(n, tmp)
    res = {v: [] for v in values}
    en = {}
    for rnd in range(3):
        for v in values:
            energy.configure(name, float(v))
            S = energy.System(atoms, basis, opts)
            for _ in range(5):
                en[v] = S.energy()
            reps = 200 if n <= 1000 else (60 if n <= 4000 else 20)
            t = time.perf_counter()
            for _ in range(reps):
                S.energy()
            res[v].append((time.perf_counter() - t) / reps * 1e6)
            S.close()
    nt = (len(atoms["pos"]) + 63) // 64
    a, b = values
    print(f"{len(atoms['pos']):5d}  {nt * (nt + 1) // 2:9d}   " + " / ".join(f"{x:7.1f}" for x in res[a]) + "      " + " / ".join(f"{x:7.1f}" for x in res[b])
          + f"      {abs(en[a] - en[b]) / abs(en[a]):.2e}", flush=True)
