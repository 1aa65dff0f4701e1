"""GPU (MI355X): round 5 -- the dipole update riding the panel launch (kernels_panel.hip: last-arriving workgroup per tile), checked against
the same sums taken by a launch of its own, bit for bit.  Reference: contract_dipoles System.Energy.cpp:3564-3598, are_we_done_yet
:3215-3239."""
import numpy as np
import pytest

import util
from mpmcxx_amd import energy, gen_box, pqr

pytestmark = pytest.mark.gpu


def big_box(tmp_path, name="ion10k_polar"):
    inp, _ = gen_box.materialize(name, str(tmp_path))
    return pqr.load_case(inp)


def evaluate(atoms, basis, opts, **switches):
    S = energy.System(atoms, basis, opts)
    for k, v in switches.items():
        S.configure(k, v)
    e = S.energy()
    obs = dict(S.observables)
    mu, E0, F = S.dipoles()
    S.close()
    return e, obs, mu, F


@pytest.mark.parametrize("extra", [{}, {"polar_precision": 1e-7, "polar_max_iter": 30}, {"polar_rrms": 1}])
def test_fused_update_equals_the_separate_launch_bit_for_bit(tmp_path, extra):
    """10 000 atoms (157 tiles, 1 648 panel entries): every tile's update is run by whichever workgroup arrives last -- the result must not
    know.  A stale slot (a hand-off that lost a write-through store) would show up here as a differing bit."""
    atoms, basis, opts = big_box(tmp_path)
    opts = dict(opts, **extra)
    ref = evaluate(atoms, basis, opts, fused_update=0, panel_reverse=0)
    for sw in ({"fused_update": 1, "panel_reverse": 1}, {"fused_update": 1, "panel_reverse": 0}, {"fused_update": 0, "panel_reverse": 1}):
        got = evaluate(atoms, basis, opts, **sw)
        assert got[0] == ref[0], sw
        assert got[1]["polarization_energy"] == ref[1]["polarization_energy"] and got[1]["polar_iterations"] == ref[1]["polar_iterations"]
        assert got[1]["dipole_rrms"] == ref[1]["dipole_rrms"] or (np.isnan(got[1]["dipole_rrms"]) and np.isnan(ref[1]["dipole_rrms"]))
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), sw
    if "polar_precision" in extra:
        assert 1 < ref[1]["polar_iterations"] < 30


def test_fused_update_on_small_and_skewed_boxes_matches_the_oracle():
    for name in ("ion216_polar", "ion216_triclinic", "ion1000_triclinic", "ion1000_polar"):
        atoms, basis, opts = util.load_fixture(name)
        a = evaluate(atoms, basis, dict(opts, solver="compact"), fused_update=1)
        b = evaluate(atoms, basis, dict(opts, solver="compact"), fused_update=0)
        assert a[0] == b[0] and np.array_equal(a[2], b[2]), name
        g = util.golden(name)
        assert util.close(a[1]["polarization_energy"], g["polar"]), name


def test_fused_update_under_uneven_load_is_reproducible(tmp_path):
    """32 beads in flight (their kernels interleave on the CUs: arrival orders differ from step to step and from bead to bead), six rounds:
    every bead must reproduce its own first result bit for bit, and equal the separate-launch path."""
    atoms, basis, opts = big_box(tmp_path)
    beads = []
    for b in range(32):
        beads.append(energy.System(dict(atoms, pos=gen_box.bead_positions(atoms["pos"], b)), basis, opts))
    first = None
    for _ in range(6):
        _, per, _ = energy.pi_potential_local(beads)
        vals = [(p["energy"], p["polarization_energy"]) for p in per]
        mus = [b.dipoles()[0] for b in beads[:4]]
        if first is None:
            first = (vals, mus)
        else:
            assert vals == first[0]
            assert all(np.array_equal(x, y) for x, y in zip(mus, first[1]))
    for b in beads:
        b.close()
    S = energy.System(dict(atoms, pos=gen_box.bead_positions(atoms["pos"], 3)), basis, opts)
    S.configure("fused_update", 0)
    assert S.energy() == first[0][3][0]
    S.close()
