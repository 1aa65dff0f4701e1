#!/usr/bin/env python3
"""Timeline of ONE launch of the panel Jacobi kernel (k_dipole_iter_panel) on the 10 000-atom box: per workgroup start / end time stamps
(wall_clock64, 100 MHz) and the CU it ran on.  Prints how full the chip was over the launch and how the last workgroups end.
usage: python tools/panel_trace.py"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mpmcxx_amd import energy  # noqa: E402

energy.configure("trace_panel", 1)
energy.configure("side_stream", 0)

atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
S = energy.System(atoms, basis, opts)
for _ in range(3):
    S.energy()
L = energy.lib()
L.mpmc_debug_panel_trace.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
buf = np.zeros((20000, 4), dtype=np.int64)
n = L.mpmc_debug_panel_trace(S.handle, buf.ctypes.data_as(C.POINTER(C.c_longlong)), 20000)
L.mpmc_debug_panel_table.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
tab = np.zeros((20000, 4), dtype=np.int32)
nt_ = L.mpmc_debug_panel_table(S.handle, tab.ctypes.data_as(C.POINTER(C.c_int)), 20000)
assert nt_ == n
t = buf[:n]
keep = t[:, 1] > 0
tab = tab[:n][keep]
t = t[keep]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # microseconds
hw, xcc = t[:, 2], t[:, 3]
cu = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)  # XCC, SE, CU (gfx9 HW_ID: cu_id bits 11:8, sh 12, se 15:13)
print(f"{len(t)} workgroups, launch spans {end.max():.1f} us (first start {start.min():.2f}, last start {start.max():.1f}); {len(np.unique(cu))} distinct CU ids")
dur = end - start
print(f"workgroup duration: mean {dur.mean():.1f} us, p10 {np.percentile(dur, 10):.1f}, median {np.median(dur):.1f}, p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}")
# by kind of entry: stored / far, members, non-uniform dimensions
kind_far, members, nonuni, diag_ = (tab[:, 2] & 8) != 0, np.where(tab[:, 1] >= 0, 2, 1), 3 - np.array([bin(int(v) & 7).count("1") for v in tab[:, 2]]), (tab[:, 2] & 16) != 0
print(" kind     members  non-uniform dims   workgroups   mean duration (us)   per tile pair (us)   share of workgroup time")
tot = dur.sum()
for far in (False, True):
    for mem in (1, 2):
        for nu in range(4):
            sel = (kind_far == far) & (members == mem) & (nonuni == nu) & ~diag_
            if sel.sum():
                print(f"  {'far   ' if far else 'stored'}   {mem}        {nu}                  {sel.sum():6d}       {dur[sel].mean():7.2f}              {dur[sel].mean() / mem:7.2f}          {dur[sel].sum() / tot:6.3f}")
sel = diag_
print(f"  diagonal (stored, 32 steps)           {sel.sum():6d}       {dur[sel].mean():7.2f}              {dur[sel].mean():7.2f}          {dur[sel].sum() / tot:6.3f}")
for far in (False, True):
    sel = (kind_far == far) & ~diag_
    print(f"  all {'far' if far else 'stored'}: {int((members[sel]).sum())} tile pairs, {dur[sel].sum() / members[sel].sum():.2f} us of workgroup time per tile pair")
grid = np.linspace(0, end.max(), 41)
print(" time(us)  workgroups resident  CUs with >= 1 workgroup")
for a, b in zip(grid[:-1], grid[1:]):
    mid = 0.5 * (a + b)
    live = (start <= mid) & (end > mid)
    print(f"  {mid:6.1f}    {live.sum():6d}              {len(np.unique(cu[live])):4d}")
per_cu_end = {}
for c_, e_ in zip(cu, end):
    per_cu_end[c_] = max(per_cu_end.get(c_, 0.0), e_)
ends = np.array(sorted(per_cu_end.values()))
print(f"per-CU finishing time: min {ends.min():.1f}, median {np.median(ends):.1f}, max {ends.max():.1f} us; mean idle at the end {(ends.max() - ends).mean():.1f} us")
S.close()
