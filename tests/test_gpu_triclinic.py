"""GPU (MI355X): non-orthorhombic cells get the same machinery as orthorhombic ones (round 3): tile-pair classes from rigorous lower
bounds in fractional coordinates (far-field and beyond-cutoff tile pairs), tile-pair-wide periodic images, and the panel form of the
Jacobi contraction -- against reference-made goldens (tests/golden/ion1000_triclinic, ion8000_triclinic: oracle/make_golden.py) and
against the same evaluation with the classes switched off."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy

pytestmark = pytest.mark.gpu


def test_8000_atom_triclinic_box_matches_reference_and_uses_the_classes(tmp_path):
    g = util.golden("ion8000_triclinic")
    atoms, basis, opts = util.load_generated("ion8000_triclinic", tmp_path)
    S = energy.System(atoms, basis, opts)
    S.energy()
    r = S.observables
    util.assert_counts(r, g, False, label="ion8000_triclinic")
    util.assert_energies(r, g, False, label="ion8000_triclinic")
    assert r["polar_iterations"] == int(g["polar_iterations"])
    mu, E, F = S.dipoles()
    st = g["sample_stride"]
    assert util.max_rel(E[::st].reshape(-1), g["ef_static_sample"]) < util.REL_TOL
    assert util.max_rel(mu[::st].reshape(-1), g["mu_sample"]) < util.REL_TOL
    assert util.max_rel(F[::st].reshape(-1), g["ef_induced_sample"]) < util.REL_TOL
    # the classes are at work in this skewed cell: tile pairs beyond the damping range are not stored, the stored ones share one periodic image
    ps = S.pair_stats()
    assert ps["tile_pairs_far"] > 0.3 * ps["tile_pairs"], ps
    # (3 non-uniform directions per pair = no common image index at all.  In a skewed cell a common index is per LATTICE direction; at 5 tiles
    # per cell edge a far tile pair always straddles a half-cell boundary in some direction, but not in all three)
    assert ps["nonuniform_dims_x_pairs_stored"] < 3 * ps["pairs_stored"], ps
    assert ps["nonuniform_dims_x_pairs_far"] < 2.5 * ps["pairs_far"], ps
    e_cls, mu_cls = r["energy"], mu.copy()
    S.close()
    # the same box with every tile pair "near" (all tensors stored, no image shortcut): same numbers to rounding
    energy.configure("tile_classes", 0)
    try:
        T = energy.System(atoms, basis, opts)
    finally:
        energy.configure("tile_classes", 1)
    T.energy()
    assert util.close(T.observables["energy"], e_cls, 1e-11)
    assert util.max_rel(T.dipoles()[0], mu_cls) < 1e-10
    assert T.pair_stats()["tile_pairs_far"] == 0
    T.close()


@pytest.mark.parametrize("solver", ["compact", "matrix_free"])
def test_1000_atom_triclinic_box_every_solver(solver):
    g = util.golden("ion1000_triclinic")
    atoms, basis, opts = util.load_fixture("ion1000_triclinic")
    S = energy.System(atoms, basis, dict(opts, solver=solver))
    S.energy()
    util.assert_counts(S.observables, g, False, label="ion1000_triclinic")
    util.assert_energies(S.observables, g, False, label="ion1000_triclinic")
    mu, E, F = S.dipoles()
    assert util.max_rel(mu.reshape(-1), g["mu"]) < util.REL_TOL
    assert util.max_rel(E.reshape(-1), g["ef_static"]) < util.REL_TOL
    S.close()


def test_lattice_translations_in_a_triclinic_cell_change_nothing():
    """moving atoms by lattice vectors of the skewed cell: same pair counts, same energies (the classes follow the raw coordinates)."""
    atoms, basis, opts = util.load_fixture("ion1000_triclinic")
    S = energy.System(atoms, basis, opts)
    S.energy()
    r0 = dict(S.observables)
    S.close()
    rng = np.random.default_rng(2)
    shift = rng.integers(-2, 3, size=atoms["pos"].shape) @ np.asarray(basis)
    T = energy.System(dict(atoms, pos=atoms["pos"] + shift), basis, opts)
    T.energy()
    r1 = T.observables
    assert r1["n_lj_in_cutoff"] == r0["n_lj_in_cutoff"] and r1["n_es_in_cutoff"] == r0["n_es_in_cutoff"]
    for k in ("rd_energy", "es_real", "es_recip", "polarization_energy"):
        assert util.close(r1[k], r0[k], 1e-9), k
    T.close()
