"""Jacobi vs Gauss-Seidel dipole solver on the 10 000-atom benchmark box: wall time per evaluation (python tools/solver_time.py)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from mpmcxx_amd import energy
atoms, basis, opts = bench.build_case(10000, tempfile.mkdtemp())
for gs in (0, 1):
    o = dict(opts); o["polar_gs"] = gs
    S = energy.System(atoms, basis, o)
    e = S.energy(); 
    t0 = time.perf_counter(); e = S.energy(); dt = time.perf_counter() - t0
    print(f"polar_gs {gs}: {dt*1e3:.2f} ms per evaluation, E {e:.12e}, pol {S.observables['polarization_energy']:.12e}, iters {S.observables['polar_iterations']}")
    S.close()
