"""CPU: the oracle (oracle/mpmc_oracle.c) against the golden vectors produced by the reference's own object code.

The oracle restates the reference's loops in the reference's own order, so on these fixtures it is not merely
within tolerance -- it is bit-identical; both facts are asserted (tolerance for portability of libm, equality
reported as a stricter check where it holds)."""
import numpy as np
import pytest

import util
from oracle import OracleSystem, pbc_update, pi_aggregate


@pytest.mark.parametrize("name", util.SMALL)
def test_oracle_energy_matches_reference(name):
    g = util.golden(name)
    atoms, basis, opts = util.load_fixture(name)
    S = OracleSystem(atoms, basis, opts)
    assert S.n == g["natoms"]
    # PeriodicBoundary restatement
    assert S.cutoff == g["cutoff"] and S.volume == g["volume"]
    assert np.array_equal(S.recip.reshape(-1), np.array(g["reciprocal_basis"]))
    assert S.s.ewald_alpha == g["ewald_alpha"] and S.s.polar_ewald_alpha == g["polar_ewald_alpha"]
    r = S.energy()
    rd_only = bool(opts["rd_only"])
    util.assert_energies(r, g, rd_only, tol=1e-13, label=name, wolf=bool(opts.get("wolf")))
    util.assert_counts(r, g, rd_only or bool(opts.get("wolf")), label=name)
    assert r["polar_iterations"] == int(g["polar_iterations"])
    assert r["iterator_failed"] == g["iterator_failed"]
    assert util.close(r["dipole_rrms"], g["dipole_rrms"], 1e-12)
    if opts["polarization"]:
        assert util.max_rel(r["ef_static"].reshape(-1), g["ef_static"]) < 1e-13
        assert util.max_rel(r["mu"].reshape(-1), g["mu"]) < 1e-13
        assert util.max_rel(r["ef_induced"].reshape(-1), g["ef_induced"]) < 1e-12
        for spot in g.get("amatrix", []):
            blk = S.amatrix_block(spot["i"], spot["j"])
            assert util.max_rel(blk, spot["block"]) < 1e-14, (name, spot["i"], spot["j"])


def test_oracle_is_bit_identical_on_the_anchor_box():
    # SURVEY.md §4 anchors (216-atom ionic box): every digit the survey quotes
    g = util.golden("ion216_polar")
    assert g["rd"] == -107838.41890820718 and g["es"] == -34951.684114394302 and g["polar"] == -785.31050811556076
    atoms, basis, opts = util.load_fixture("ion216_polar")
    r = OracleSystem(atoms, basis, opts).energy()
    assert r["rd_energy"] == g["rd"] and r["coulombic_energy"] == g["es"] and r["polarization_energy"] == g["polar"]
    assert r["energy"] == g["total"]


def test_argon_dimer_analytic():
    # Ar2 at exactly 4 A, eps 119.8 K, sigma 3.405 A (pi001 geometry): U_LJ = -112.95570892593686 K
    atoms, basis, opts = util.load_fixture("ar2")
    r = OracleSystem(atoms, basis, opts).energy()
    s = 3.405 / 4.0
    assert abs(r["lj_pairs"] - 4 * 119.8 * (s ** 12 - s ** 6)) < 1e-12
    assert abs(r["rd_energy"] - (-112.95570892593686)) < 1e-11


def test_minimum_image_triclinic_roundtrip():
    atoms, basis, opts = util.load_fixture("ion216_triclinic")
    S = OracleSystem(atoms, basis, opts)
    R = S.recip
    for (i, j) in [(0, 1), (0, 215), (17, 101), (100, 7)]:
        rimg, d, r = S.minimum_image(i, j)
        raw = atoms["pos"][i] - atoms["pos"][j]
        shift = raw - d  # must be an integer combination of lattice vectors
        frac = shift @ R
        assert np.allclose(frac, np.rint(frac), atol=1e-9)
        assert rimg <= r + 1e-12


def test_pi_aggregate_ordered_mean():
    rng = np.random.default_rng(0)
    rd, es, pol = rng.normal(size=8) * 1e5, rng.normal(size=8) * 1e4, rng.normal(size=8) * 1e2
    v, obs = pi_aggregate(rd, es, pol)
    acc = 0.0
    for x in rd:
        acc += x
    assert obs[0] == acc / 8
    assert v == obs[0] + obs[1] + obs[3] + obs[2]
