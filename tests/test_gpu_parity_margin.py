"""GPU (MI355X): where the parity margin of the headline comes from, pinned instead of left to luck (round-3 review, weak 1b / 1c).

(1) `rd_energy` at 10 000 atoms.  The reference adds every pair's `rd + lrc` onto one accumulator in list order
    (System.Energy.cpp:1011): 5e7 additions of terms a fraction of an ulp of the running -5e6 K sum wide, a rounding DRIFT that grows
    with N.  The HIP path sums trees and takes the pair LRC in its O(N) moment form.  `orc_lj_exact` (oracle, CPU) sums the reference's
    own fp64 pair terms exactly (long double + Neumaier): the HIP path must sit within 1e-12 of THAT; the reference's own distance from
    it is recorded (and bounded, so that a change of the fixture that moves it is noticed).
(2) `kTholeFarX` = 30 (csrc/kernels.h): beyond lambda r = 30 a tile pair's tensors lose their exponential damping, which still differs
    from 1 by 4.7e-10 there.  Pairs are placed AT the boundary, on both sides of it, and every dipole is held to the oracle at 1e-9."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy
from oracle import OracleSystem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return abs(a - b) / abs(b)


def test_rd_energy_against_exact_sums_at_10k(tmp_path):
    atoms, basis, opts = util.load_generated("ion10k_es", tmp_path)
    S = energy.System(atoms, basis, opts)
    S.energy()
    r = dict(S.observables)
    S.close()
    x = OracleSystem(atoms, basis, opts).lj_exact()
    # the HIP path against the exactly rounded sums of the reference's own pair terms
    assert rel(r["lj_pairs"], x["lj_pairs_exact"]) <= 1e-12, (r["lj_pairs"], x["lj_pairs_exact"])
    assert rel(r["lrc_pair"], x["lrc_pair_exact"]) <= 1e-12, (r["lrc_pair"], x["lrc_pair_exact"])
    assert rel(r["lrc_self"], x["lrc_self_exact"]) <= 1e-12
    assert rel(r["rd_energy"], x["rd_exact"]) <= 1e-12, (r["rd_energy"], x["rd_exact"])
    # the reference's list-order accumulation against the same exact sums: this, not the HIP path, is the 1.6e-10 of bench.py's
    # cpu_baseline.parity_rel_err (5.8e-10 on the harness's lrc_pair column); recorded here so that nobody has to re-derive it
    drift_rd, drift_lrc = rel(x["rd_list_order"], x["rd_exact"]), rel(x["lrc_pair_list_order"], x["lrc_pair_exact"])
    print(f"reference list-order drift at 10 000 atoms: rd {drift_rd:.2e}, lrc_pair {drift_lrc:.2e}, lj_pairs {rel(x['lj_pairs_list_order'], x['lj_pairs_exact']):.2e}")
    assert 1e-11 < drift_rd < 1e-9 and 1e-10 < drift_lrc < 1e-9
    # and the contract itself (1e-9 against the reference as it is) holds with the margin the drift leaves
    assert rel(r["rd_energy"], x["rd_list_order"]) < 1e-9


def two_clusters(gap):
    """two tiles of 64 polarizable ions: tile 0 in the cube [0, 6]^3, tile 1 the same cube moved by 6 + gap along x, so that the tiles'
    bounding boxes are `gap` apart and the 16 atoms of the facing faces form 16 pairs at exactly that distance (128 atoms: the library
    keeps the caller's order, a tile is 64 consecutive atoms)."""
    g = np.array([0.0, 2.0, 4.0, 6.0])
    cube = np.array([[x, y, z] for x in g for y in g for z in g])  # 64 sites, x major: x = 6 is the face towards tile 1
    cube_b = cube.copy()
    cube_b[:, 0] = 6.0 + gap + (6.0 - cube[:, 0])  # mirrored: its x = 6 face sits at 6 + gap
    pos = np.concatenate([cube, cube_b]) - np.array([10.0, 3.0, 3.0])
    n = pos.shape[0]
    atoms = {"pos": pos, "charge": 408.7816 * 0.1 * np.where(np.arange(n) % 2 == 0, 1.0, -1.0), "polarizability": np.full(n, 1.6411),
             "epsilon": np.full(n, 119.8), "sigma": np.full(n, 3.405), "mol_id": np.arange(n, dtype=np.int32), "frozen": np.zeros(n, dtype=np.int32),
             "mass": np.full(n, 39.948)}
    return atoms, np.diag([120.0, 120.0, 120.0])


@pytest.mark.parametrize("solver", ["compact", "matrix_free"])
@pytest.mark.parametrize("side", [-1e-6, +1e-6, -1e-9, +1e-9, 0.0])
def test_pairs_at_the_thole_far_boundary(side, solver):
    lam = 2.1304
    atoms, basis = two_clusters(30.0 / lam + side)
    opts = {"polarization": 1, "polar_damp": lam, "damp_type": "exponential", "polar_iterative": 1, "polar_max_iter": 10, "polar_ewald": 1,
            "ewald_kmax": 7, "solver": solver}
    ref = OracleSystem(atoms, basis, opts).energy()
    S = energy.System(atoms, basis, opts)
    S.energy()
    r = S.observables
    mu, E0, Eind = S.dipoles()
    ts = S.tile_stats()
    S.close()
    # the class really is what the test is about: the one off-diagonal tile pair is far beyond the boundary, stored inside it
    # (the class is decided with a safety margin on the bounding boxes, so 1e-9 beyond the boundary may still be "stored": the safe side)
    if solver == "compact":
        assert ts["tile_pairs"] == 3 and (ts["thole_far"] == 1 if side >= 1e-6 else (ts["thole_far"] == 0 if side <= 0 else True)), ts
    scale = np.abs(ref["mu"]).max()
    assert np.abs(mu - ref["mu"]).max() <= 1e-9 * scale, (side, np.abs(mu - ref["mu"]).max() / scale)
    per_dipole = np.abs(mu - ref["mu"]).max(axis=1) / np.maximum(np.abs(ref["mu"]).max(axis=1), 1e-3 * scale)
    assert per_dipole.max() <= 1e-9, (side, per_dipole.max())
    assert np.abs(Eind - ref["ef_induced"]).max() <= 1e-9 * np.abs(ref["ef_induced"]).max()
    assert rel(r["polarization_energy"], ref["polarization_energy"]) <= 1e-9
