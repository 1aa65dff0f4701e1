"""GPU (MI355X): BASELINE configs[4] -- the 32-bead path-integral ensemble of the 10 000-atom polarizable box -- at FULL size on one GPU.

tests/golden/ion10k_polar_bead{0,1}.json hold the reference's own energies (oracle/_ref/ref_harness, build container) of images 0 and 1 of
exactly the ensemble bench.py evaluates (mpmcxx_amd.gen_box.bead_positions: base box + Gaussian displacement on the 6-decimal grid of a PQR
file, written through the PQR path).  Here: all 32 images are built on the one GPU, evaluated through mpmc_pi_potential_local, images 0 / 1
compared with those goldens at 1e-9 (counts bit-exact, a 64-atom sample of E0 / mu / E_ind), every image's stand-alone evaluation
with its in-ensemble one (bit for bit), and the ordered mean with the oracle's restatement of PI_calculate_potential (pi_aggregate)."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy, gen_box

pytestmark = pytest.mark.gpu
P = 32


@pytest.fixture(scope="module")
def ensemble(tmp_path_factory):
    atoms, basis, opts = util.load_generated("ion10k_polar", tmp_path_factory.mktemp("c5"))
    beads = [energy.System(dict(atoms, pos=gen_box.bead_positions(atoms["pos"], b)), basis, opts) for b in range(P)]
    sums, per, failed = energy.pi_potential_local(beads)
    yield atoms, basis, opts, beads, sums, per, failed
    for s in beads:
        s.close()


def test_bead_fixture_is_the_bench_ensemble(tmp_path):
    """the golden's box and bench.py's bead 0 are the same doubles: PQR text -> double is the identity on the 6-decimal grid"""
    import bench

    atoms, basis, opts = util.load_generated("ion10k_polar", tmp_path)
    a0, _, _ = util.load_generated("ion10k_polar_bead0", tmp_path)
    assert np.array_equal(a0["pos"], bench.bead_positions(atoms["pos"], 0))
    assert np.array_equal(a0["pos"], gen_box.bead_positions(atoms["pos"], 0))
    assert np.abs(a0["pos"] - atoms["pos"]).max() > 0.1  # really displaced


@pytest.mark.parametrize("b", [0, 1])
def test_images_0_and_1_match_the_reference(ensemble, b):
    atoms, basis, opts, beads, sums, per, failed = ensemble
    g = util.golden(f"ion10k_polar_bead{b}")
    util.assert_counts(per[b], g, False, label=f"bead{b}")
    util.assert_energies(per[b], g, False, label=f"bead{b}")
    assert per[b]["polar_iterations"] == int(g["polar_iterations"]) and not failed
    mu, E, F = beads[b].dipoles()
    st = g["sample_stride"]
    assert util.max_rel(E[::st].reshape(-1), g["ef_static_sample"]) < util.REL_TOL
    assert util.max_rel(mu[::st].reshape(-1), g["mu_sample"]) < util.REL_TOL
    assert util.max_rel(F[::st].reshape(-1), g["ef_induced_sample"]) < util.REL_TOL


def test_ordered_mean_over_32_images(ensemble):
    from oracle import pi_aggregate

    atoms, basis, opts, beads, sums, per, failed = ensemble
    v, obs = energy.pi_finish(sums, P)
    v_ref, obs_ref = pi_aggregate([p["rd_energy"] for p in per], [p["coulombic_energy"] for p in per], [p["polarization_energy"] for p in per])
    assert v == v_ref and np.array_equal(obs, obs_ref)
    # the two reference-pinned images inside the sums: replacing them by the golden values moves the mean by < 1e-9 relative
    g0, g1 = util.golden("ion10k_polar_bead0"), util.golden("ion10k_polar_bead1")
    swapped = sum(p["energy"] for p in per[2:]) + g0["total"] + g1["total"]
    assert util.close(swapped / P, v, 1e-10)
    # every image is a distinct configuration, and in-ensemble evaluation equals a stand-alone one bit for bit
    assert len({p["energy"] for p in per}) == P
    for b in (0, 7, 31):
        assert beads[b].energy() == per[b]["energy"]


def test_unperturbed_box_sample_of_per_atom_vectors(tmp_path):
    """BASELINE configs[3]: besides the energies (test_gpu_parity) a 64-atom sample of E0 / mu / E_ind at 10 000 atoms"""
    g = util.golden("ion10k_polar")
    atoms, basis, opts = util.load_generated("ion10k_polar", tmp_path)
    S = energy.System(atoms, basis, opts)
    S.energy()
    mu, E, F = S.dipoles()
    st = g["sample_stride"]
    assert util.max_rel(E[::st].reshape(-1), g["ef_static_sample"]) < util.REL_TOL
    assert util.max_rel(mu[::st].reshape(-1), g["mu_sample"]) < util.REL_TOL
    assert util.max_rel(F[::st].reshape(-1), g["ef_induced_sample"]) < util.REL_TOL
    S.close()


def test_dense_mfma_solver_at_full_size(tmp_path):
    """BASELINE configs[3] taken literally: the 3N x 3N matrix (6.8 GiB) in device memory, contraction on v_mfma_f64_16x16x4_f64"""
    g = util.golden("ion10k_polar")
    atoms, basis, opts = util.load_generated("ion10k_polar", tmp_path)
    S = energy.System(atoms, basis, dict(opts, solver="dense"))
    S.energy()
    r = S.observables
    util.assert_counts(r, g, False, label="dense10k")
    util.assert_energies(r, g, False, label="dense10k")
    mu, E, F = S.dipoles()
    st = g["sample_stride"]
    assert util.max_rel(mu[::st].reshape(-1), g["mu_sample"]) < util.REL_TOL
    total, tensor = S.memory_usage()
    assert total > 6 * 2 ** 30  # the dense matrix really is resident
    S.close()


@pytest.mark.parametrize("name", ["ion10k_polar", "ion10k_polar_bead0", "ion8000_triclinic"])
def test_every_atom_of_the_large_boxes(name, tmp_path):
    """round 3: E0 / mu / E_ind of EVERY atom of the large polarizable boxes against the reference's own vectors (tests/golden/NAME_atoms.npz,
    oracle/make_golden_atoms.py), not only the 64-atom sample of the JSON goldens."""
    import os

    g = np.load(os.path.join(util.GOLDEN, f"{name}_atoms.npz"))
    atoms, basis, opts = util.load_generated(name, tmp_path)
    S = energy.System(atoms, basis, opts)
    e = S.energy()
    assert util.close(e, float(g["total"])) and util.close(S.observables["polarization_energy"], float(g["polar"]))
    mu, E, F = S.dipoles()
    assert mu.shape == g["mu"].shape
    assert util.max_rel(E, g["ef_static"]) < util.REL_TOL
    assert util.max_rel(mu, g["mu"]) < util.REL_TOL
    assert util.max_rel(F, g["ef_induced"]) < util.REL_TOL
    # ... and atom by atom, relative to the atom's own dipole where that is not tiny
    scale = np.abs(g["mu"]).max()
    big = np.linalg.norm(g["mu"], axis=1) > 1e-3 * scale
    rel = np.linalg.norm(mu - g["mu"], axis=1)[big] / np.linalg.norm(g["mu"], axis=1)[big]
    assert rel.max() < 1e-8, rel.max()
    S.close()
