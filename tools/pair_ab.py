#!/usr/bin/env python3
"""Same-box A/B of pair-sweep variants by the clock bench.py's roofline uses: `reps` launches back to back between ONE pair of HIP events
on the kernel's stream, kernel alone on the GPU (mpmc_debug_time_pair).  Boxes: BASELINE configs[2] (10 000-atom LJ + Ewald: sweep without
field and store) and configs[3] (polarizable: field + tensor store).
usage: python tools/pair_ab.py "label:key=v,key=v" ...      (keys of mpmc_debug_configure; the list is run three times, interleaved)"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpmcxx_amd import energy, gen_box, pqr  # noqa: E402

specs = sys.argv[1:] or ["default:"]
reps = int(os.environ.get("PAB_REPS", "60"))
wd = tempfile.mkdtemp()
for name in ("ion10k_es", "ion10k_polar"):
    inp, _ = gen_box.materialize(name, wd)
    atoms, basis, opts = pqr.load_case(inp)
    systems = {}
    for spec in specs:
        label, _, envs = spec.partition(":")
        S = energy.System(atoms, basis, opts)
        S.configure("side_stream", 0)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            S.configure(k, float(v))
        e = S.energy()
        S.energy()
        systems[label] = (S, e)
    for rnd in range(int(os.environ.get("PAB_ROUNDS", "3"))):
        row = []
        for label, (S, e) in systems.items():
            S.energy()
            row.append(f"{label} {S.time_kernel('pair', reps) * 1e3:.1f} us" + (f" (panel {S.time_kernel('panel', reps) * 1e3:.1f})" if name == "ion10k_polar" else ""))
        print(f"{name} r{rnd}: " + "   ".join(row), flush=True)
    print(f"{name} energies: " + "  ".join(f"{label} {e:.13e}" for label, (S, e) in systems.items()), flush=True)
    for S, _ in systems.values():
        S.close()
