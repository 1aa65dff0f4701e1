"""One polar_gs evaluation of the 10 000-atom box (for rocprofv3 --kernel-trace --stats: python3 tools/gs_trace.py)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from mpmcxx_amd import energy
atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
o = dict(opts); o["polar_gs"] = 1
S = energy.System(atoms, basis, o)
S.energy()
t0 = time.perf_counter()
for _ in range(3):
    e = S.energy()
print(f"gauss_seidel: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per evaluation, pol {S.observables['polarization_energy']:.12e}")
S.close()
