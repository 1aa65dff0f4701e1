"""Inputs that stress the geometry bookkeeping of the HIP path against the oracle: unwrapped coordinates, anisotropic cells, boxes far
from the origin, coordinates on a coarse grid (hundreds of pairs exactly on the half-box tie of the minimum image, where the reference's
answer depends on rint's tie rule applied to the RAW displacement), and a cell smaller than one atom tile."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy
from oracle import OracleSystem

pytestmark = pytest.mark.gpu


def build(case):
    atoms, basis, opts = util.load_fixture("ion1000_polar")
    rng = np.random.default_rng(3)
    a = dict(atoms)
    if case == "unwrapped":
        a["pos"] = atoms["pos"] + rng.integers(-3, 4, size=atoms["pos"].shape) * basis[0, 0]
    elif case == "anisotropic":
        basis = np.diag([40.0, 25.0, 70.0])
        a["pos"] = atoms["pos"] * np.array([1.0, 25 / 40, 70 / 40])
    elif case == "far_from_origin":
        a["pos"] = atoms["pos"] + np.array([1234.5, -987.25, 55.125])
    elif case == "integer_grid_ties":
        a["pos"] = np.round(atoms["pos"], 0)
    elif case == "cell_smaller_than_a_tile":
        basis = np.diag([12.0, 12.0, 12.0])
        a = {k: v[:200] for k, v in atoms.items()}
        a["pos"] = rng.uniform(-6, 6, size=(200, 3))
        a["mol_id"] = np.arange(200, dtype=np.int32)
    return a, basis, opts


@pytest.mark.parametrize("solver", ["compact", "matrix_free"])
@pytest.mark.parametrize("case", ["unwrapped", "anisotropic", "far_from_origin", "integer_grid_ties", "cell_smaller_than_a_tile"])
def test_geometry_edge_cases(case, solver):
    a, basis, opts = build(case)
    ref = OracleSystem(a, basis, opts).energy()
    S = energy.System(a, basis, dict(opts, solver=solver))
    S.energy()
    r = S.observables
    for k in ("rd_energy", "coulombic_energy", "polarization_energy", "energy"):
        assert abs(r[k] - ref[k]) <= 1e-9 * max(abs(ref[k]), 1e-3 * abs(ref["energy"])), (case, k, r[k], ref[k])
    assert int(r["n_lj_in_cutoff"]) == int(ref["n_lj_in_cutoff"]) and int(r["n_es_in_cutoff"]) == int(ref["n_es_in_cutoff"])
    assert r["polar_iterations"] == ref["polar_iterations"]
    mu = S.dipoles()[0]
    assert np.abs(mu - ref["mu"]).max() <= 1e-9 * np.abs(ref["mu"]).max() + 1e-13
    S.close()
