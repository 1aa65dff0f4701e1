import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import util
from mpmcxx_amd import energy
for name in util.SMALL:
    atoms, basis, opts = util.load_fixture(name)
    if not opts["polarization"]:
        continue
    g = util.golden(name)
    S = energy.System(atoms, basis, opts)
    S.energy()
    r = S.observables
    mu, E, F = S.dipoles()
    mr = util.max_rel(mu.reshape(-1), g["mu"])
    gr = g["dipole_rrms"]
    print(f"{name:28s} rrms gpu {r['dipole_rrms']:.15e} ref {gr:.15e} rel diff {abs(r['dipole_rrms']-gr)/max(abs(gr),1e-300):.2e}  max_rel(mu) {mr:.2e}  mu/rrms bound {mr/max(gr,1e-300):.2e} iters {r['polar_iterations']}")
    S.close()
