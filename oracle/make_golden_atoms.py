#!/usr/bin/env python3
"""Full per-atom vectors (E0, mu, E_ind of EVERY atom) of the large polarizable boxes from the reference's own object code, stored as
compressed float64 arrays: tests/golden/NAME_atoms.npz.  The JSON goldens of these boxes (oracle/make_golden.py) hold the energies and a
64-atom sample; this closes the gap the round-2 review named (per-atom parity at 10 000 atoms on a sample only).
Fixtures are DATA: inputs regenerated deterministically by mpmcxx_amd/gen_box.py, outputs of oracle/_ref/ref_harness --dump-atoms.
usage: python oracle/make_golden_atoms.py [names...]     (build container only: needs oracle/_ref)"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import GOLDEN, run_harness  # noqa: E402
from mpmcxx_amd import gen_box  # noqa: E402

for name in sys.argv[1:] or ["ion10k_polar", "ion10k_polar_bead0", "ion8000_triclinic"]:
    wd = tempfile.mkdtemp(prefix="golden_atoms_")
    inp, _ = gen_box.materialize(name, wd)
    res = run_harness(inp, ["--dump-atoms"])
    n = res["natoms"]
    out = {k: np.asarray(res[k], dtype=np.float64).reshape(n, 3) for k in ("ef_static", "mu", "ef_induced")}
    np.savez_compressed(os.path.join(GOLDEN, f"{name}_atoms.npz"), total=np.float64(res["total"]), polar=np.float64(res["polar"]), **out)
    print(f"{name}: n={n} polar={res['polar']!r} -> {name}_atoms.npz ({os.path.getsize(os.path.join(GOLDEN, name + '_atoms.npz'))} bytes)")
    shutil.rmtree(wd, ignore_errors=True)
