import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import util
from mpmcxx_amd import energy
import tempfile
for name, nb, steps in (("ion1000_polar", 16, 300), ("water64_polar", 32, 300), ("ion10k_polar", 8, 60)):  # (10 000 atoms: two streams, polled waits, fused tail)
    atoms, basis, opts = util.load_generated(name, tempfile.mkdtemp()) if name in util.LARGE else util.load_fixture(name)
    beads = []
    for b in range(nb):
        a = dict(atoms); a["pos"] = atoms["pos"] + np.random.default_rng(b).normal(scale=0.03, size=atoms["pos"].shape)
        beads.append(energy.System(a, basis, opts))
    sums0, per0, _ = energy.pi_potential_local(beads)
    ref = [(p["energy"], p["polarization_energy"], p["rd_energy"], p["coulombic_energy"], p["n_lj_in_cutoff"]) for p in per0]
    bad = 0
    for s in range(steps):
        sums, per, failed = energy.pi_potential_local(beads)
        cur = [(p["energy"], p["polarization_energy"], p["rd_energy"], p["coulombic_energy"], p["n_lj_in_cutoff"]) for p in per]
        if cur != ref or failed:
            bad += 1
    print(name, nb, "beads x", steps, "steps: mismatching steps", bad, flush=True)
    for b in beads: b.close()
