#!/usr/bin/env python3
"""EXPERIMENT (round 4): spatial orders on an aligned grid (sort_nx x sort_ny strips, serpentine, z inside) against the nested count-based
bisection: non-uniform dimensions per pair, stored tile pairs, the hot kernels alone and the evaluation, on the benchmark box."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from mpmcxx_amd import energy
atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
variants = [(0, 0), (4, 4), (4, 6), (6, 6), (6, 4), (4, 8), (6, 8), (2, 8), (8, 8)]
for rnd in range(2):
    for nx, ny in variants:
        energy.configure("sort_nx", nx); energy.configure("sort_ny", ny)
        try:
            S = energy.System(atoms, basis, opts)
        finally:
            energy.configure("sort_nx", 0); energy.configure("sort_ny", 0)
        S.configure("side_stream", 0)
        e = S.energy(); S.energy()
        ps, ts = S.pair_stats(), S.tile_stats()
        t0 = time.perf_counter()
        for _ in range(20): S.energy()
        ev = (time.perf_counter() - t0) / 20
        pan = sorted(S.time_kernel("panel", 60) for _ in range(3))[1]
        par = sorted(S.time_kernel("pair", 30) for _ in range(3))[1]
        nu_far = ps["nonuniform_dims_x_pairs_far"] / max(ps["pairs_far"], 1); nu_st = ps["nonuniform_dims_x_pairs_stored"] / max(ps["pairs_stored"], 1)
        nu_sw = ps["nonuniform_dims_x_pairs_swept"] / max(ps["pairs_swept"], 1)
        print(f"r{rnd} grid {nx}x{ny}: E {e:.10e}  nonuniform dims/pair far {nu_far:.2f} stored {nu_st:.2f} swept {nu_sw:.2f}  tile pairs stored {ts['thole_stored']} beyond {ts['beyond_cutoff']}"
              f"  panel {pan*1e3:.1f} us  pair {par*1e3:.1f} us  eval(one stream) {ev*1e6:.0f} us", flush=True)
        S.close()
