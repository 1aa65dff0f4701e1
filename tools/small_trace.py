#!/usr/bin/env python3
"""Launch list of ONE steady-state evaluation of a small polarizable fixture (default ion216_polar), for rocprofv3 --kernel-trace:
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/small_trace.py [fixture] [evals]
then  python3 tools/small_trace.py --report $OUT   prints, for the last evaluation, every kernel with its duration and the gap before it."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # evaluations are separated by the pair sweep
    starts = [i for i, r in enumerate(rows) if "k_pair_fused" in r["Kernel_Name"] or "k_tile_bounds" in r["Kernel_Name"]]
    firsts = [i for k, i in enumerate(starts) if k == 0 or rows[starts[k - 1]]["Kernel_Name"] == rows[i]["Kernel_Name"] or True]
    tb = [i for i, r in enumerate(rows) if "k_tile_bounds" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "k_pair_fused" in r["Kernel_Name"]]
    a, b = tb[-2], tb[-1]
    ev = rows[a:b]
    t0 = int(ev[0]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    print(f"{len(ev)} launches in the evaluation before the last one; span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
    for r in ev:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        name = r["Kernel_Name"].replace("mpmc::", "").split("(")[0][:58]
        print(f"  +{(s - t0) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  run {(e - s) / 1e3:6.1f}  {name}")
        prev_end = e
    print(f"kernel time {busy / 1e3:.1f} us")
    sys.exit(0)

import time  # noqa: E402

import util  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ion216_polar"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if name in util.LARGE:  # (the 10 000-atom boxes are regenerated, not committed)
    import tempfile

    atoms, basis, opts = util.load_generated(name, tempfile.mkdtemp())
else:
    atoms, basis, opts = util.load_fixture(name)
S = energy.System(atoms, basis, opts)
for arg in sys.argv[3:]:  # "cfg:key=v,key=v": measurement switches of this context
    if arg.startswith("cfg:"):
        for kv in filter(None, arg[4:].split(",")):
            k, _, v = kv.partition("=")
            S.configure(k, float(v))
if len(sys.argv) > 3 and sys.argv[3] == "volume":  # us per volume move (NPT / Gibbs): set_box with scaled positions + full evaluation
    import numpy as np

    pos0 = atoms["pos"].copy()
    S.energy()
    t_set = t_en = 0.0
    for it in range(n + 10):
        if it == 10:
            t_set = t_en = 0.0
        f = 1.0 + 0.002 * ((it % 2) * 2 - 1)
        t0 = time.perf_counter()
        S.set_box(basis * f)
        S.update_positions(0, pos0 * f)
        t1 = time.perf_counter()
        S.energy()
        t2 = time.perf_counter()
        t_set += t1 - t0
        t_en += t2 - t1
    print(f"{name}: {len(pos0)} atoms, volume move: set_box + positions {t_set / n * 1e6:.1f} us + energy {t_en / n * 1e6:.1f} us (python loop)")
    S.close()
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "resize":  # us per N-changing move (uVT / Gibbs insert or remove): set_atoms + full evaluation
    import numpy as np

    ids = atoms["mol_id"]
    last = int(np.nonzero(ids == ids[-1])[0][0])
    fewer = {k: (v[:last].copy() if isinstance(v, np.ndarray) and len(v) == len(ids) else v) for k, v in atoms.items()}
    S.energy()
    t_set = t_en = 0.0
    for it in range(n + 10):
        if it == 10:
            t_set = t_en = 0.0
        a = fewer if it % 2 == 0 else atoms
        t0 = time.perf_counter()
        S.set_atoms(a)
        t1 = time.perf_counter()
        S.energy()
        t2 = time.perf_counter()
        t_set += t1 - t0
        t_en += t2 - t1
    print(f"{name}: {len(ids)} atoms, remove / insert the last molecule ({len(ids) - last} atoms): set_atoms {t_set / n * 1e6:.1f} us + energy {t_en / n * 1e6:.1f} us (python loop)")
    S.close()
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "trial":  # us per trial move: trial_energy + reject, and trial_energy + accept
    import numpy as np

    rng = np.random.default_rng(1)
    ids = atoms["mol_id"]
    starts = [0] + [i for i in range(1, len(ids)) if ids[i] != ids[i - 1]] + [len(ids)]
    pos = atoms["pos"].copy()
    S.energy()
    out = []
    for accept in (False, True):
        moves = []
        for _ in range(n + 20):
            k = int(rng.integers(len(starts) - 1))
            moves.append((starts[k], starts[k + 1], rng.normal(scale=0.05, size=(starts[k + 1] - starts[k], 3))))
        t = 0.0
        for it, (a, b, d) in enumerate(moves):
            if it == 20:
                t = time.perf_counter()
            S.trial_energy(a, pos[a:b] + d)
            if accept:
                S.accept()
                pos[a:b] += d
            else:
                S.reject()
        out.append((time.perf_counter() - t) / n * 1e6)
    print(f"{name}: {len(pos)} atoms, trial + reject {out[0]:.1f} us, trial + accept {out[1]:.1f} us (python loop)")
    S.close()
    sys.exit(0)
for _ in range(20):
    S.energy()
t = time.perf_counter()
for _ in range(n):
    S.energy()
dt = (time.perf_counter() - t) / n
print(f"{name}: {len(atoms['pos'])} atoms, {dt * 1e6:.1f} us per evaluation (python loop), iterations {S.observables['polar_iterations']}")
S.close()
