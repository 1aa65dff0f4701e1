#!/usr/bin/env python3
"""CPU only: how far the REFERENCE's own accumulation order (System.Energy.cpp:1011: every pair's `rd + lrc` added onto one accumulator
in list order) drifts from the exactly rounded sum of the same fp64 terms, as a function of the number of atoms.  The HIP path sums
trees and takes the pair LRC in its O(N) moment form, so it sits next to the exact value; the gap a parity test sees on `rd_energy` /
`lrc_pair` at large N is this drift.  usage: python tools/lrc_drift.py [N ...]   (boxes: bench.build_case's density, polarization off)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
from oracle import OracleSystem  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1000, 5000, 10000, 20000, 40000]
print("natoms  pairs  rel|rd_list - rd_exact|  rel|lrc_pair_list - lrc_pair_exact|  rel|lj_pairs_list - lj_pairs_exact|  seconds")
for n in sizes:
    atoms, basis, opts = bench.build_case(n, tempfile.mkdtemp())
    opts = dict(opts, polarization=0)
    t0 = time.time()
    x = OracleSystem(atoms, basis, opts).lj_exact()
    rel = lambda a, b: abs(a - b) / abs(b)
    print(f"{n:6d}  {n * (n - 1) // 2:11d}  {rel(x['rd_list_order'], x['rd_exact']):.2e}  {rel(x['lrc_pair_list_order'], x['lrc_pair_exact']):.2e}  "
          f"{rel(x['lj_pairs_list_order'], x['lj_pairs_exact']):.2e}  {time.time() - t0:.0f}", flush=True)
