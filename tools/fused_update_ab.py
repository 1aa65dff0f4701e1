#!/usr/bin/env python3
"""The dipole update riding the panel launch (fused_update) against the launch of its own, ONE bead of the 10 000-atom box, one stream:
ms per panel launch back to back (mpmc_debug_time_panel), wall time per evaluation, HIP-event ms of the kernel classes.
fused_update = 2 is the producer side alone (write-through stores, waits, arrival atomics; no update -- results invalid).
usage: python tools/fused_update_ab.py [rounds]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
variants = [("separate, table order", 0, 0), ("separate, descending", 0, 1), ("arrivals only, descending", 2, 1), ("fused, table order", 1, 0), ("fused, descending", 1, 1)]
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for label, fused, rev in variants:
        S = energy.System(atoms, basis, opts)
        S.configure("side_stream", 0)
        S.configure("fused_update", fused)
        S.configure("panel_reverse", rev)
        for _ in range(3):
            e = S.energy()
        t0 = time.perf_counter()
        for _ in range(10):
            S.energy()
        wall = (time.perf_counter() - t0) / 10
        S.set_profiling(True)
        S.timings(reset=True)
        for _ in range(3):
            S.energy()
        t = S.timings(reset=True)
        S.set_profiling(False)
        S.energy()
        runs = sorted(S.time_kernel("panel", 100) for _ in range(3))
        S.close()
        per_eval = {k: v["ms"] / 3 for k, v in t.items() if v["launches"]}
        print(f"r{rnd} {label:>26s}: panel launch {runs[1] * 1e3:7.2f} us back to back | evaluation {wall * 1e3:.3f} ms | per evaluation: dipole_iter "
              f"{per_eval.get('dipole_iter', 0) * 1e3:.1f} us + reduce {per_eval.get('reduce', 0) * 1e3:.1f} us | E {e:.10e}", flush=True)
