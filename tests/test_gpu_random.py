"""GPU (MI355X): randomized parity of the HIP path against the oracle on small systems that exercise the edge cases
of the domain: ragged sizes around the 64-atom tile (1, 2, 63, 64, 65, 129 ...), multi-atom molecules (exclusions,
intramolecular term), frozen atoms, zero / negative sigma, zero epsilon, zero charges ("es_excluded" inter-molecular
pairs), zero polarizabilities, dispersion flags, orthorhombic and triclinic cells, every option combination of the
path, and the solver fallbacks.  Tolerance 1e-9 relative per component (floored at 1e-12 of the largest component,
for components that vanish by cancellation); pair counts bit-exact."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy

pytestmark = pytest.mark.gpu

E2R = 408.7816


def random_system(rng, n_target, cell):
    lmin = max(9.0, (n_target * 45.0) ** (1.0 / 3.0))  # keeps the density physical
    if cell == "cubic":
        L = rng.uniform(lmin, lmin + 8.0)
        basis = np.diag([L, L, L])
    elif cell == "ortho":
        basis = np.diag(rng.uniform(lmin, lmin + 10.0, size=3))
    else:
        L = rng.uniform(lmin + 2.0, lmin + 10.0)
        basis = np.array([[L, 0, 0], [rng.uniform(-3, 3), L * rng.uniform(0.9, 1.1), 0], [rng.uniform(-3, 3), rng.uniform(-3, 3), L * rng.uniform(0.9, 1.1)]])
    inv = np.linalg.inv(basis)
    pos, q, al, ep, sg, mol, fr, disp, mass = [], [], [], [], [], [], [], [], []
    m = 0
    tries = 0
    while len(pos) < n_target:
        tries += 1
        assert tries < 200000
        size = int(rng.choice([1, 1, 1, 2, 3, 4]))
        size = min(size, n_target - len(pos))
        frac = rng.uniform(-0.7, 1.7, size=3)  # also outside the cell: positions are unwrapped in the reference
        center = frac @ basis
        if pos:  # minimum-image distance to everything placed so far
            d = np.array(pos) - center
            f = d @ inv
            d = d - np.rint(f) @ basis
            if np.min(np.linalg.norm(d, axis=1)) < 3.2:
                continue
        frozen = rng.random() < 0.15
        for k in range(size):
            pos.append(center + (rng.normal(scale=0.45, size=3) if k else 0.0))
            q.append(0.0 if rng.random() < 0.25 else rng.uniform(-0.9, 0.9) * E2R)
            al.append(0.0 if rng.random() < 0.3 else rng.uniform(0.2, 1.5))
            e = 0.0 if rng.random() < 0.15 else rng.uniform(5.0, 150.0)
            s = rng.uniform(2.0, 3.4)
            r = rng.random()
            if r < 0.1:
                s = 0.0
            elif r < 0.15:
                s = -s
            ep.append(e)
            sg.append(s)
            mol.append(m)
            fr.append(1 if frozen else 0)
            disp.append(1 if rng.random() < 0.05 else 0)
            mass.append(rng.uniform(1.0, 40.0))
        m += 1
    atoms = {"pos": np.array(pos), "charge": np.array(q), "polarizability": np.array(al), "epsilon": np.array(ep), "sigma": np.array(sg),
             "mol_id": np.array(mol, dtype=np.int32), "frozen": np.array(fr, dtype=np.int32), "has_disp": np.array(disp, dtype=np.int32),
             "mass": np.array(mass)}
    return atoms, basis


def random_options(rng):
    o = {"rd_only": 0, "rd_lrc": int(rng.random() < 0.8), "polarization": 0, "polar_iterative": 0, "polar_ewald": 0, "polar_max_iter": 10,
         "polar_gs": 0, "polar_rrms": 0, "ewald_kmax": int(rng.choice([3, 5, 7])), "polar_precision": 0.0, "polar_gamma": 1.0, "polar_damp": 0.0,
         "damp_type": "exponential", "ewald_alpha": None, "polar_ewald_alpha": None}
    mode = rng.choice(["lj", "es", "polar_ewald", "polar_nopbc", "polar_ewald", "polar_precision"])
    if mode == "lj":
        o["rd_only"] = 1
    elif mode != "es":
        o.update(polarization=1, polar_iterative=1, polar_damp=float(rng.uniform(1.5, 2.6)), polar_max_iter=int(rng.integers(1, 6)),
                 polar_ewald=int(mode != "polar_nopbc"), polar_gamma=float(rng.choice([1.0, 1.0, 1.05])), polar_rrms=int(rng.random() < 0.3))
        if mode == "polar_precision":
            o.update(polar_precision=float(rng.choice([1e-3, 1e-5])), polar_max_iter=10)
    if rng.random() < 0.3:
        o["ewald_alpha"] = float(rng.uniform(0.2, 0.45))
    if rng.random() < 0.2:
        o["polar_ewald_alpha"] = float(rng.uniform(0.2, 0.45))
    return o


def check(atoms, basis, opts, label, wolf=False):
    from oracle import OracleSystem

    ref = OracleSystem(atoms, basis, opts).energy()
    S = energy.System(atoms, basis, opts)
    S.energy()
    r = S.observables
    if not np.isfinite(ref["energy"]):
        assert not np.isfinite(r["energy"]), label
        S.close()
        return
    keys = ["energy", "rd_energy", "coulombic_energy", "polarization_energy", "es_real", "es_recip", "es_self", "lj_pairs", "lrc_pair", "lrc_self"]
    if wolf:  # coulombic_wolf has no real / reciprocal / self split (the oracle reports the total only)
        keys = [k for k in keys if not k.startswith("es_")]
    scale = max(abs(ref[k]) for k in keys)
    for k in keys:
        tol = 1e-9 * max(abs(ref[k]), 1e-3 * scale) + 1e-12  # absolute floor: a lone atom's energies are pure rounding noise
        assert abs(r[k] - ref[k]) <= tol, (label, k, r[k], ref[k])
    for k in ["n_pairs", "n_intra", "n_rd_excluded", "n_es_excluded", "n_frozen", "n_lj_in_cutoff"]:
        assert int(r[k]) == int(ref[k]), (label, k, r[k], ref[k])
    if not opts["rd_only"] and not wolf:  # (the oracle does not count the pairs of coulombic_wolf)
        assert int(r["n_es_in_cutoff"]) == int(ref["n_es_in_cutoff"]), label
    if opts["polarization"] and not opts["rd_only"]:
        assert r["polar_iterations"] == ref["polar_iterations"], (label, r["polar_iterations"], ref["polar_iterations"])
        assert r["iterator_failed"] == ref["iterator_failed"], label
        mu, E, F = S.dipoles()
        assert np.abs(E - ref["ef_static"]).max() <= 1e-9 * np.abs(ref["ef_static"]).max() + 1e-12, label
        if not ref["iterator_failed"]:
            assert np.abs(mu - ref["mu"]).max() <= 1e-8 * np.abs(ref["mu"]).max() + 1e-12, label
        assert abs(r["dipole_rrms"] - ref["dipole_rrms"]) <= 1e-6 * abs(ref["dipole_rrms"]) + 1e-14, label
    S.close()


SIZES = [1, 2, 3, 17, 63, 64, 65, 127, 129, 200, 321]


@pytest.mark.parametrize("seed", range(24))
def test_random_systems_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    n = SIZES[seed % len(SIZES)]
    cell = ["cubic", "ortho", "triclinic"][seed % 3]
    atoms, basis = random_system(rng, n, cell)
    opts = random_options(rng)
    check(atoms, basis, opts, f"seed {seed} n {n} {cell} {opts}")


@pytest.mark.parametrize("solver", ["compact", "matrix_free"])
def test_solver_variants_agree(solver):
    rng = np.random.default_rng(7)
    atoms, basis = random_system(rng, 300, "ortho")
    opts = random_options(np.random.default_rng(2))
    opts.update(rd_only=0, polarization=1, polar_iterative=1, polar_ewald=1, polar_damp=2.1304, polar_max_iter=6, polar_precision=0.0, solver=solver)
    check(atoms, basis, opts, f"solver {solver}")


def test_auto_solver_falls_back_when_the_store_does_not_fit():
    atoms, basis, opts = util.load_fixture("ion216_polar")
    S = energy.System(atoms, basis, opts)
    S.configure("tensor_budget_mb", 0)  # nothing fits: AUTO must recompute tensors, still on the GPU
    S.energy()
    assert S.memory_usage()[1] == 0
    g = util.golden("ion216_polar")
    util.assert_energies(S.observables, g, False)
    S.close()


def test_coincident_atoms_give_a_non_finite_energy():
    """the reference divides by rimg == 0 (System.Energy.cpp:965): the MC driver rejects the move on a non-finite energy"""
    atoms, basis, opts = util.load_fixture("lj64")
    a = dict(atoms)
    a["pos"] = atoms["pos"].copy()
    a["pos"][5] = a["pos"][9]
    S = energy.System(a, basis, opts)
    assert not np.isfinite(S.energy())
    S.close()


def test_input_validation():
    atoms, basis, opts = util.load_fixture("lj64")
    bad = dict(atoms)
    bad["pos"] = atoms["pos"].copy()
    bad["pos"][3, 1] = np.nan
    with pytest.raises(energy.MpmcError) as ei:
        energy.System(bad, basis, opts)
    assert ei.value.code == 6001  # invalid_datum
    bad = dict(atoms)
    bad["mol_id"] = atoms["mol_id"].copy()
    bad["mol_id"][10] = bad["mol_id"][2]  # molecule id reappears later: not a contiguous molecule
    with pytest.raises(energy.MpmcError) as ei:
        energy.System(bad, basis, opts)
    assert ei.value.code == 6001
    with pytest.raises(energy.MpmcError) as ei:
        energy.System(atoms, np.zeros((3, 3)), opts)
    assert ei.value.code == 6004  # invalid_box_dimensions


def test_context_grows_past_its_capacity_hint():
    """max_atoms is a hint: a context created for 64 atoms takes 128 (insertions in the uVT / Gibbs ensembles), keeps its box and
    options, and gives the energy a fresh context gives."""
    atoms, basis, opts = util.load_fixture("lj64")
    S = energy.System(atoms, basis, opts, max_atoms=64)
    e64 = S.energy()
    big = {k: np.concatenate([v, v]) for k, v in atoms.items()}
    big["pos"] = np.concatenate([atoms["pos"], atoms["pos"] + 1.7])
    big["mol_id"] = np.arange(128, dtype=np.int32)
    S.set_atoms(big)
    e128 = S.energy()
    F = energy.System(big, basis, opts)
    assert e128 == F.energy() and S.observables["n_lj_in_cutoff"] == F.observables["n_lj_in_cutoff"]
    S.set_atoms(atoms)  # and back down
    assert S.energy() == e64
    S.close()
    F.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_systems_gauss_seidel(seed):
    """polar_gs: in-place sweeps in atom order (the spatial sort is off for this solver); ragged sizes, mixed molecules, frozen and
    non-polarizable sites, fixed-count and precision-terminated solves."""
    rng = np.random.default_rng(7000 + seed)
    n = [3, 64, 65, 130, 200, 321, 129, 17][seed]
    cell = ["cubic", "ortho", "triclinic"][seed % 3]
    atoms, basis = random_system(rng, n, cell)
    opts = random_options(rng)
    opts.update(rd_only=0, polarization=1, polar_iterative=1, polar_gs=1, polar_damp=float(rng.uniform(1.5, 2.6)), polar_ewald=int(seed % 2 == 0),
                polar_max_iter=int(rng.integers(1, 5)), polar_rrms=int(seed % 3 == 0))
    if seed >= 5:
        opts.update(polar_precision=1e-4, polar_max_iter=10)
    check(atoms, basis, opts, f"gs seed {seed} n {n} {cell} {opts}")


@pytest.mark.parametrize("seed", range(4))
def test_random_mid_size_systems(seed):
    """700-1500 atoms: many tiles, mixed tile-pair classes (cutoff / damping range / uniform image), molecules spanning tile borders."""
    rng = np.random.default_rng(9000 + seed)
    n = [700, 1100, 1500, 900][seed]
    cell = ["ortho", "cubic", "triclinic", "ortho"][seed]
    atoms, basis = random_system(rng, n, cell)
    opts = random_options(rng)
    opts.update(rd_only=0, polarization=1, polar_iterative=1, polar_damp=float(rng.uniform(1.5, 2.6)), polar_ewald=int(seed != 1),
                polar_max_iter=int(rng.integers(2, 5)), polar_gs=0)
    check(atoms, basis, opts, f"mid seed {seed} n {n} {cell} {opts}")


@pytest.mark.parametrize("seed", range(4))
def test_one_big_frozen_framework_molecule_with_mobile_sorbates(seed):
    """The usual MPMC input: ONE frozen molecule of several hundred charged polarizable sites (a framework: every framework pair is
    intramolecular AND frozen, the molecule spans many 64-atom tiles) plus mobile molecules of 1-4 sites; also a big MOBILE rigid molecule
    (seed 3) whose intramolecular pairs cross tile boundaries."""
    rng = np.random.default_rng(8100 + seed)
    n_frame = [300, 517, 130, 0][seed]
    atoms, basis = random_system(rng, n_frame + 150, ["cubic", "ortho", "triclinic", "cubic"][seed])
    n = len(atoms["charge"])
    order = np.arange(n)
    if n_frame:  # the first n_frame atoms become one frozen molecule; keep molecules contiguous and ids dense
        atoms["mol_id"][:n_frame] = 0
        atoms["frozen"][:n_frame] = 1
        atoms["mol_id"][n_frame:] = 1 + np.unique(atoms["mol_id"][n_frame:], return_inverse=True)[1]
        # a molecule that straddled the cut keeps its frozen flag per atom; make the tail molecules movable
        atoms["frozen"][n_frame:] = 0
    else:  # one big mobile molecule of 150 sites
        atoms["mol_id"][:150] = 0
        atoms["frozen"][:150] = 0
        atoms["mol_id"][150:] = 1 + np.unique(atoms["mol_id"][150:], return_inverse=True)[1]
    atoms["mol_id"] = atoms["mol_id"].astype(np.int32)
    for mode in ("es", "polar_ewald", "polar_nopbc"):
        opts = random_options(rng)
        opts.update(rd_only=0, polarization=int(mode != "es"), polar_iterative=int(mode != "es"), polar_ewald=int(mode == "polar_ewald"),
                    polar_damp=2.1304, polar_max_iter=6, polar_precision=0.0)
        check(atoms, basis, opts, (seed, mode))


@pytest.mark.parametrize("kmax", [1, 2, 9, 15, 16, 21])
def test_reciprocal_space_cutoffs_on_both_sides_of_the_phase_table_limit(kmax):
    """ewald_kmax up to 15 uses the factorised phase tables in LDS, above that one sincos per (k, atom): both against the oracle, with and
    without the polarization field (K = 27 ... 19 000 k-vectors)."""
    rng = np.random.default_rng(9000 + kmax)
    atoms, basis = random_system(rng, 150, "ortho" if kmax % 2 else "triclinic")
    for polar in (0, 1):
        opts = random_options(rng)
        opts.update(rd_only=0, ewald_kmax=kmax, polarization=polar, polar_iterative=polar, polar_ewald=polar, polar_damp=2.1304, polar_max_iter=4,
                    polar_precision=0.0)
        check(atoms, basis, opts, (kmax, polar))


@pytest.mark.parametrize("n", [2, 5, 30, 64, 100, 128, 300])
def test_dense_solver_on_small_and_ragged_systems(n):
    """fewer tiles than the 16 row chunks of the dense matrix-vector product (its partial slots once overran the slot buffer and
    landed in the matrix behind it: right after one iteration, wrong from the second); found by tools/fuzz.py"""
    rng = np.random.default_rng(4000 + n)
    atoms, basis = random_system(rng, n, ["cubic", "ortho", "triclinic"][n % 3])
    for it, precision in ((1, 0.0), (4, 0.0), (10, 1e-5)):
        opts = random_options(rng)
        opts.update(rd_only=0, polarization=1, polar_iterative=1, polar_ewald=int(n % 2), polar_damp=2.1304, polar_max_iter=it, polar_precision=precision,
                    polar_gs=0, solver="dense")
        check(atoms, basis, opts, (n, it, precision))
