"""One-off scale check (not a test): energies of 10k / 20k-atom polarizable boxes against the CPU oracle, component by component,
and lattice-translation invariance at 20k / 40k atoms (40k: the tensor store exceeds its budget, matrix-free solver)."""
import os, sys, time, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench
from mpmcxx_amd import energy
from oracle import OracleSystem
for n in (10000, 20000, 40000):
    atoms, basis, opts = bench.build_case(n, tempfile.mkdtemp())
    S = energy.System(atoms, basis, opts); t0 = time.perf_counter(); e = S.energy(); S.energy(); dt = (time.perf_counter() - t0) / 2
    r = dict(S.observables); tot, ten = S.memory_usage(); S.close()
    a2 = dict(atoms); a2["pos"] = atoms["pos"] + np.array([3 * basis[0, 0], -2 * basis[1, 1], basis[2, 2]])
    T = energy.System(a2, basis, opts); e2 = T.energy(); r2 = dict(T.observables); T.close()
    print(f"n={n}: {dt*1e3:.1f} ms/eval, store {ten/2**30:.2f} GiB, E={e:.12e}; lattice-translated: rel diff {abs(e2-e)/abs(e):.1e}, pol {abs(r2['polarization_energy']-r['polarization_energy'])/abs(r['polarization_energy']):.1e}", flush=True)
    if n <= 20000:
        t0 = time.perf_counter(); ref = OracleSystem(atoms, basis, opts).energy(want_atoms=False)
        keys = ["energy", "rd_energy", "lj_pairs", "lrc_pair", "lrc_self", "coulombic_energy", "es_real", "es_recip", "es_self", "polarization_energy"]
        print(f"   oracle {time.perf_counter()-t0:.0f} s; relative differences: " + ", ".join(f"{k} {abs(ref[k]-r[k])/max(abs(ref[k]),1e-300):.1e}" for k in keys), flush=True)
