"""GPU (MI355X): the fast pair sweep (csrc/kernels_pair.hip: erfc table in LDS, four tile pairs per workgroup behind one j-tile, uniform
periodic images, three grades of exclusion logic) against the reference goldens, the oracle and the generic kernel it replaces.  By default
the library uses it for tables of more than kSweepMinPairs = 2048 tile pairs (the 10 000-atom boxes of test_gpu_parity / test_gpu_config5);
here it is forced onto small boxes, where the fixtures and the oracle are.  Tolerance 1e-9 per component, pair counts bit-exact."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy
from test_gpu_random import check, random_options, random_system

pytestmark = pytest.mark.gpu


@pytest.fixture
def force_sweep():
    energy.configure("pair_kernel", 2)
    yield
    energy.configure("pair_kernel", 0)


# Ewald electrostatics in any cell: the sweep's domain (ion216_alpha sets two different alphas: the generic kernel keeps it); the skewed
# cells take the reference's full image arithmetic per pair, or one common image per tile pair where k_classify found one
SWEEP_FIXTURES = ["ion64_es", "ion216_polar", "ion216_frozen", "ion216_precision", "ion216_gamma", "water64_polar", "ion1000_polar",
                  "ion216_framework", "ion216_triclinic", "ion1000_triclinic"]


@pytest.mark.parametrize("name", SWEEP_FIXTURES)
def test_sweep_matches_reference_golden(name, force_sweep):
    g = util.golden(name)
    atoms, basis, opts = util.load_fixture(name)
    S = energy.System(atoms, basis, opts)
    e = S.energy()
    assert S.last_pair_kernel() == "sweep", name
    r = S.observables
    util.assert_counts(r, g, False, label=name)
    util.assert_energies(r, g, False, label=name)
    assert util.close(e, g["total"])
    assert r["polar_iterations"] == int(g["polar_iterations"])
    if opts["polarization"]:
        mu, E, F = S.dipoles()
        assert util.max_rel(E.reshape(-1), g["ef_static"]) < util.REL_TOL
        assert util.max_rel(mu.reshape(-1), g["mu"]) < util.REL_TOL
        assert util.max_rel(F.reshape(-1), g["ef_induced"]) < util.REL_TOL
    S.close()


@pytest.mark.parametrize("name", ["ion216_alpha", "ion216_polar_nopbc", "ion216_wolf", "water64_fh2", "lj64"])
def test_outside_its_domain_the_generic_kernel_runs(name, force_sweep):
    """two Ewald alphas, no-PBC field, Wolf, Feynman-Hibbs, LJ only: not the sweep's -- and still the reference's numbers."""
    g = util.golden(name)
    atoms, basis, opts = util.load_fixture(name)
    S = energy.System(atoms, basis, opts)
    S.energy()
    assert S.last_pair_kernel() == "fused", name
    util.assert_energies(S.observables, g, bool(opts["rd_only"]), label=name, wolf=bool(opts.get("wolf")))
    S.close()


@pytest.mark.parametrize("name", ["ion1000_polar", "water64_polar", "ion216_framework", "ion1000_triclinic"])
def test_sweep_and_generic_kernel_agree_to_1e12(name):
    atoms, basis, opts = util.load_fixture(name)
    res = {}
    for mode, label in ((1, "fused"), (2, "sweep")):
        energy.configure("pair_kernel", mode)
        try:
            S = energy.System(atoms, basis, opts)
        finally:
            energy.configure("pair_kernel", 0)
        S.energy()
        assert S.last_pair_kernel() == label
        res[label] = (dict(S.observables), S.dipoles())
        S.close()
    a, b = res["fused"][0], res["sweep"][0]
    for k in ("lj_pairs", "es_real", "polarization_energy", "energy"):
        assert abs(a[k] - b[k]) <= 1e-12 * max(abs(a[k]), 1e-3 * abs(a["energy"])), (name, k, a[k], b[k])
    for k in ("n_lj_in_cutoff", "n_es_in_cutoff"):
        assert int(a[k]) == int(b[k]), (name, k)
    for va, vb in zip(res["fused"][1], res["sweep"][1]):
        assert np.abs(va - vb).max() <= 1e-11 * np.abs(va).max() + 1e-15, name


def plain_system(rng, n_target, cell, molecules):
    """every atom charged, with epsilon and sigma > 0, none frozen: all tile pairs are the sweep's; `molecules`: 1-4 atoms per molecule
    (exclusions + the erf form of the field inside molecules) or one atom each."""
    atoms, basis = random_system(rng, n_target, cell)
    n = len(atoms["charge"])
    atoms["charge"] = rng.uniform(0.1, 0.9, size=n) * rng.choice([-1.0, 1.0], size=n) * 408.7816
    atoms["epsilon"] = rng.uniform(5.0, 150.0, size=n)
    atoms["sigma"] = rng.uniform(2.0, 3.4, size=n)
    atoms["frozen"] = np.zeros(n, dtype=np.int32)
    atoms["has_disp"] = np.zeros(n, dtype=np.int32)
    if not molecules:
        atoms["mol_id"] = np.arange(n, dtype=np.int32)
    return atoms, basis


@pytest.mark.parametrize("seed", range(16))
def test_plain_random_systems_match_oracle(seed, force_sweep):
    rng = np.random.default_rng(4000 + seed)
    n = [63, 64, 65, 127, 129, 200, 321, 500][seed % 8]  # ragged sizes: padding slots in the last tile
    cell = ["cubic", "ortho"][seed % 2]
    atoms, basis = plain_system(rng, n, cell, molecules=(seed % 4 < 2))
    opts = random_options(rng)
    opts.update(rd_only=0, ewald_alpha=None, polar_ewald_alpha=None)
    if opts["polarization"]:
        opts["polar_ewald"] = 1
    S_probe = energy.System(atoms, basis, opts)
    S_probe.energy()
    assert S_probe.last_pair_kernel() == "sweep"
    S_probe.close()
    check(atoms, basis, opts, f"plain seed {seed} n {n} {cell} {opts}")


@pytest.mark.parametrize("seed", range(12))
def test_mixed_random_systems_match_oracle(seed, force_sweep):
    """frozen / chargeless / sigma- and epsilon-less atoms in some tiles: those tile pairs take the sweep's masked grade (MODE 2), the plain
    ones its lean grades, and the tile pairs with a sigma < 0 or dispersion-coefficient atom go through the generic kernel on its list --
    all in the same evaluation."""
    rng = np.random.default_rng(5000 + seed)
    n = [129, 200, 321, 450][seed % 4]
    cell = ["cubic", "ortho"][seed % 2]
    atoms, basis = random_system(rng, n, cell)
    # make the first ~half of the atom list plain (atoms are spatially sorted inside the library, so tiles end up mixed or plain)
    m = len(atoms["charge"]) // 2
    atoms["charge"][:m] = np.where(atoms["charge"][:m] == 0.0, 0.3 * 408.7816, atoms["charge"][:m])
    atoms["epsilon"][:m] = np.where(atoms["epsilon"][:m] == 0.0, 50.0, atoms["epsilon"][:m])
    atoms["sigma"][:m] = np.where(atoms["sigma"][:m] <= 0.0, 3.0, atoms["sigma"][:m])
    atoms["frozen"][:m] = 0
    atoms["has_disp"][:m] = 0
    opts = random_options(rng)
    opts.update(rd_only=0, ewald_alpha=None, polar_ewald_alpha=None)
    if opts["polarization"]:
        opts["polar_ewald"] = 1
    check(atoms, basis, opts, f"mixed seed {seed} n {n} {cell} {opts}")


def test_run_to_run_determinism_of_the_sweep(force_sweep):
    atoms, basis, opts = util.load_fixture("ion1000_polar")
    S = energy.System(atoms, basis, opts)
    e = [S.energy() for _ in range(4)]
    assert S.last_pair_kernel() == "sweep"
    assert e[0] == e[1] == e[2] == e[3]
    S.close()


def _pairs_on_the_cutoff(offset=0.0):
    """ion1000_polar with 24 atoms moved so that 24 pairs sit AT the cutoff: r = r_c (1 + delta) for delta from 0 to +-1e-9, i.e. inside the
    band in which the sweep's fused geometry defers to the reference's form (kernels_pair.hip: sweep_step), and just outside it."""
    atoms, basis, opts = util.load_fixture("ion1000_polar")
    pos = atoms["pos"].copy()
    rc = 0.5 * basis[0, 0]
    rng = np.random.default_rng(42)
    deltas = [0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-15, -1e-15, 1e-14, -1e-14, 1e-13, -1e-13, 6e-13, -6e-13, 1e-12, -1e-12, 1e-11, -1e-11,
              1e-10, -1e-10, 5e-10, -5e-10, 9e-10, 2e-9, -2e-9]
    for k, d in enumerate(deltas):
        i, j = 40 * k, 40 * k + 17
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        pos[j] = pos[i] + u * rc * (1.0 + d)
    return dict(atoms, pos=pos + offset), basis, opts


@pytest.mark.parametrize("offset", [0.0, 777.125, 3.0e6])
def test_pairs_on_the_cutoff_are_counted_like_the_reference(offset, force_sweep):
    """pair inclusion is bit-exact with the fused geometry too: a step with a pair inside the 1e-9 band around the cutoff thresholds redoes
    its geometry in the reference's operation order; coordinates far from the origin (3e6 A: the band becomes "everything")."""
    from oracle import OracleSystem

    atoms, basis, opts = _pairs_on_the_cutoff(offset)
    ref = OracleSystem(atoms, basis, opts).energy()
    res = {}
    for fast in (1, 0):
        energy.configure("fast_geometry", fast)
        try:
            S = energy.System(atoms, basis, opts)
        finally:
            energy.configure("fast_geometry", 1)
        S.energy()
        assert S.last_pair_kernel() == "sweep"
        res[fast] = dict(S.observables)
        S.close()
    for fast in (1, 0):
        r = res[fast]
        assert int(r["n_lj_in_cutoff"]) == int(ref["n_lj_in_cutoff"]) and int(r["n_es_in_cutoff"]) == int(ref["n_es_in_cutoff"]), (fast, offset)
        tol = 1e-9 if offset < 1e6 else 1e-7  # (3e6 A from the origin the coordinates themselves carry 5e-10 A)
        for k in ("rd_energy", "coulombic_energy", "polarization_energy", "energy"):
            assert abs(r[k] - ref[k]) <= tol * max(abs(ref[k]), 1e-3 * abs(ref["energy"])), (fast, offset, k, r[k], ref[k])
    for k in ("lj_pairs", "es_real", "polarization_energy"):
        assert abs(res[1][k] - res[0][k]) <= 1e-13 * max(abs(res[0][k]), 1e-3 * abs(res[0]["energy"])), (offset, k)


@pytest.mark.parametrize("name", ["ion1000_polar", "water64_polar", "ion216_framework", "ion1000_triclinic", "ion216_frozen"])
def test_two_waves_per_tile_pair_form(name, force_sweep):
    """pair_split = 1 (round 4; off by default: it shortens a lone launch's tail and costs 0.7 % with 32 beads in flight): two waves share
    a tile pair, half the steps each, and meet in LDS -- the reference's numbers, and the one-wave form's to rounding."""
    g = util.golden(name)
    atoms, basis, opts = util.load_fixture(name)
    res = {}
    for split in (1, 0):
        energy.configure("pair_split", split)
        try:
            S = energy.System(atoms, basis, opts)
        finally:
            energy.configure("pair_split", -1)
        S.energy()
        assert S.last_pair_kernel() == "sweep"
        res[split] = (dict(S.observables), S.dipoles())
        S.close()
    util.assert_counts(res[1][0], g, False, label=name)
    util.assert_energies(res[1][0], g, False, label=name)
    for k in ("lj_pairs", "es_real", "polarization_energy", "energy"):
        assert abs(res[1][0][k] - res[0][0][k]) <= 1e-12 * max(abs(res[0][0][k]), 1e-3 * abs(res[0][0]["energy"])), (name, k)
    for va, vb in zip(res[1][1], res[0][1]):
        assert np.abs(va - vb).max() <= 1e-11 * np.abs(vb).max() + 1e-15, name
