"""CPU: the exact-sum mode of the oracle (orc_lj_exact) pinned independently -- math.fsum over the pair terms of the 1000-atom LJ box
computed with numpy -- and its list-order columns held to orc_lj bit for bit.  (The mode is the measuring stick of
tests/test_gpu_parity_margin.py; tools/lrc_drift.py prints the drift it measures as a function of N.)"""
import math

import numpy as np

import util
from oracle import OracleSystem


def test_exact_sums_of_the_lj1000_box():
    atoms, basis, opts = util.load_fixture("lj1000")
    O = OracleSystem(atoms, basis, opts)
    ref = O.energy(want_atoms=False)
    x = O.lj_exact()
    # list-order columns: the reference's accumulation, identical to orc_lj (and so to the golden)
    assert x["rd_list_order"] == ref["rd_energy"] and x["lj_pairs_list_order"] == ref["lj_pairs"] and x["lrc_pair_list_order"] == ref["lrc_pair"]
    assert util.close(ref["rd_energy"], util.golden("lj1000")["rd"], 1e-15)
    # independent exact sum of the same terms (cubic cell, single-site atoms: d - L rint(d / L))
    pos, L = atoms["pos"], basis[0, 0]
    sig, eps = atoms["sigma"], atoms["epsilon"]
    terms, lrcs = [], []
    cut = 0.5 * L
    vol = L ** 3
    for i in range(pos.shape[0] - 1):
        d = pos[i] - pos[i + 1:]
        d = d - L * np.rint(d * (1.0 / L))
        r = np.sqrt((d * d).sum(axis=1))
        s = 0.5 * (sig[i] + sig[i + 1:])
        e = np.sqrt(eps[i] * eps[i + 1:])
        sor = s / r
        s6 = sor * sor * sor
        s6 = s6 * s6
        t = 4.0 * e * (s6 * s6 - s6)
        terms.extend(t[r - 1e-12 < cut].tolist())
        sc = s / cut
        sc3 = sc * sc * sc
        lrcs.extend((((16.0 / 3.0) * math.pi * e * s ** 3) * ((1.0 / 3.0) * sc3 ** 3 - sc3) / vol).tolist())
    assert abs(math.fsum(terms) - x["lj_pairs_exact"]) <= 1e-14 * abs(x["lj_pairs_exact"])
    assert abs(math.fsum(lrcs) - x["lrc_pair_exact"]) <= 1e-14 * abs(x["lrc_pair_exact"])
    assert abs(x["rd_exact"] - (x["lj_pairs_exact"] + x["lrc_pair_exact"] + x["lrc_self_exact"])) <= 1e-15 * abs(x["rd_exact"])
    # what the list order costs already at 1000 atoms (5e-13), three orders above the exact mode's own accuracy
    assert 1e-14 < abs(x["rd_list_order"] - x["rd_exact"]) / abs(x["rd_exact"]) < 1e-11
