#!/usr/bin/env python3
"""The same 8000 polarizable atoms in a skewed cell (tests' `ion8000_triclinic`) and in an orthorhombic cell of the same edge lengths:
ms per evaluation (one bead at a time) and HIP-event ms per launch of the kernel classes -- what the general-cell path costs.
usage: python tools/triclinic_rate.py"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpmcxx_amd import energy, gen_box, pqr  # noqa: E402

tri = [[79.8, 0.0, 0.0], [9.0, 77.0, 0.0], [-6.0, 11.0, 75.0]]
ortho = [[79.8, 0.0, 0.0], [0.0, 77.0, 0.0], [0.0, 0.0, 75.0]]
for label, basis, grid in (("orthorhombic", ortho, -1), ("triclinic", tri, -1), ("orthorhombic, bisection order", ortho, 0), ("triclinic, bisection order", tri, 0)):
    energy.configure("sort_grid", grid)  # (-1: the aligned-grid spatial order of large tables, round 4; 0: the nested bisection of rounds 1-3)
    wd = tempfile.mkdtemp()
    gen_box.write_pqr(os.path.join(wd, "b.pqr"), gen_box.lattice_box_cell(8000, basis, 22))
    gen_box.write_input(os.path.join(wd, "b.in"), "b.pqr", basis, dict(gen_box.POLAR_OPTS))
    atoms, b, opts = pqr.load_case(os.path.join(wd, "b.in"))
    S = energy.System(atoms, b, opts)
    S.configure("side_stream", 0)
    for _ in range(2):
        e = S.energy()
    S.set_profiling(True)
    S.timings(reset=True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        e = S.energy()
    wall = (time.perf_counter() - t0) / reps
    t = S.timings(reset=True)
    st = S.pair_stats()
    cls = "  ".join(f"{k} {v['ms'] / max(v['launches'], 1):.4f}x{v['launches'] // reps}" for k, v in t.items() if v["launches"])
    print(f"{label:>30s}: eval {wall * 1e3:.3f} ms  E {e:.10e} | {cls} | tile pairs {st.get('tile_pairs')} stored {st.get('tile_pairs_stored')} far {st.get('tile_pairs_far')} "
          f"non-uniform dims x pairs far {st.get('nonuniform_dims_x_pairs_far')}", flush=True)
    S.close()
