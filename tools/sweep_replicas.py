#!/usr/bin/env python3
"""What the pair sweep costs per tile-pair table when the launch is long enough for its tail not to matter: the grid repeated R times
in y (same work, same outputs), ms per launch / R.  The difference to R = 1 is the price of filling and draining the chip once.
usage: python tools/sweep_replicas.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpmcxx_amd import energy, gen_box, pqr  # noqa: E402

wd = tempfile.mkdtemp()
for name in ("ion10k_es", "ion10k_polar"):
    inp, _ = gen_box.materialize(name, wd)
    atoms, basis, opts = pqr.load_case(inp)
    S = energy.System(atoms, basis, opts)
    S.configure("side_stream", 0)
    S.energy()
    S.energy()
    for rnd in range(2):
        row = []
        for R in (1, 2, 4, 8):
            S.configure("panel_replicas", R)
            row.append(f"R={R}: {S.time_kernel('pair', 30) * 1e3 / R:.1f} us")
        print(f"{name} r{rnd}: " + "   ".join(row), flush=True)
    S.configure("panel_replicas", 1)
    S.close()
