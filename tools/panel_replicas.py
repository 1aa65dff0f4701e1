#!/usr/bin/env python3
"""mpmc_debug_time_panel with the panel kernel's grid repeated in y: what one launch for several beads would cost per bead (measurement
only; the repeated systems write the same values into the same slots).  usage: python tools/panel_replicas.py"""
import sys, tempfile, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
S = energy.System(atoms, basis, opts)
S.configure("side_stream", 0)
for _ in range(3): S.energy()
for rep in (1, 2, 4, 8, 16, 32, 1):
    S.configure("panel_replicas", rep)
    ms = S.time_kernel("panel", 20)
    print(f"replicas {rep:2d}: {ms*1e3:8.1f} us per launch = {ms*1e3/rep:6.1f} us per system", flush=True)
