"""GPU (MI355X): regression tests for the round-2 advisor findings fixed in round 3, and for the round's own host-side changes."""
import ctypes as C

import numpy as np
import pytest

import util
from mpmcxx_amd import energy

pytestmark = pytest.mark.gpu


def test_gauss_seidel_after_a_carried_insertion_sweeps_in_atom_order():
    """upload_atoms skipped the sort whenever the spatial order had been carried across an insertion, even when set_options had
    asked for a new order in between -- Gauss-Seidel sweeps (polar_gs) must run in the reference's atom order (System.Energy.cpp:3569)."""
    atoms, basis, opts = util.load_fixture("ion216_polar")
    opts = dict(opts, polar_max_iter=4)
    S = energy.System(atoms, basis, opts)
    S.energy()
    # one atom inserted at the end of the list (a contiguous insertion: the order is carried) ...
    a2 = {k: np.concatenate([v, v[-1:]]) for k, v in atoms.items()}
    a2["pos"][-1] = atoms["pos"][-1] + np.array([1.7, -1.3, 0.9])
    a2["mol_id"][-1] = atoms["mol_id"].max() + 1
    S.set_atoms(a2)
    # ... then the sweeps are switched to Gauss-Seidel BEFORE the next evaluation
    gs = dict(opts, polar_gs=1)
    S.set_options(gs)
    e = S.energy()
    F = energy.System(a2, basis, gs)
    ef = F.energy()
    assert e == ef, (e, ef)  # same order, same arithmetic: bit for bit
    from oracle import OracleSystem

    ref = OracleSystem(a2, basis, gs).energy()
    assert util.close(e, ref["energy"]) and util.max_rel(S.dipoles()[0], ref["mu"]) < util.REL_TOL
    S.close()
    F.close()


def test_first_evaluation_of_many_fresh_contexts():
    """the launch-number slot the waits poll was never initialised; a recycled pinned block could hold an old context's 1.0 and
    the first poll of a new context would then return the old results.  Fresh contexts back to back, one evaluation each."""
    atoms, basis, opts = util.load_fixture("lj64")
    g = util.golden("lj64")
    rng = np.random.default_rng(0)
    for k in range(40):
        a = dict(atoms, pos=atoms["pos"] + (rng.normal(scale=0.02, size=atoms["pos"].shape) if k % 2 else 0.0))
        S = energy.System(a, basis, opts)
        e = S.energy()
        if k % 2 == 0:
            assert util.close(e, g["total"])
        else:
            assert e != g["total"]
        S.close()


def test_wait_counters_count():
    """mpmc_debug_wait_counters: short evaluations are polled for (seen), long ones synchronise their stream."""
    L = energy.lib()
    L.mpmc_debug_wait_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    atoms, basis, opts = util.load_fixture("lj64")
    S = energy.System(atoms, basis, opts)
    for _ in range(5):
        S.energy()
    w = (C.c_longlong * 4)()
    assert L.mpmc_debug_wait_counters(S.handle, w) == 0
    assert w[0] + w[1] + w[2] >= 5 and w[0] >= 4, list(w)
    S.close()


def test_configure_rejects_unknown_keys_and_bad_values():
    atoms, basis, opts = util.load_fixture("lj64")
    S = energy.System(atoms, basis, opts)
    with pytest.raises(energy.MpmcError):
        S.configure("no_such_switch", 1)
    with pytest.raises(energy.MpmcError):
        S.configure("pair_kernel", 7)
    with pytest.raises(energy.MpmcError):
        S.configure("pair_waves", 3)
    S.close()


@pytest.mark.parametrize("key,value", [("side_stream", 0), ("side_stream", 1), ("pair_waves", 1), ("pair_waves", 4), ("panels", 0), ("uniform_images", 0),
                                       ("tile_classes", 0), ("recip_table", 0), ("spatial_sort", 0), ("polar_delta", 0), ("pair_kernel", 1), ("pair_kernel", 2),
                                       ("lazy_side_stream", 0)])
def test_every_switch_leaves_the_reference_numbers(key, value):
    """the measurement switches of mpmc_debug_configure select other kernels or orders, never other physics"""
    g = util.golden("ion1000_polar")
    atoms, basis, opts = util.load_fixture("ion1000_polar")
    energy.configure(key, value)
    try:
        S = energy.System(atoms, basis, opts)
    finally:
        energy.configure(key, {"side_stream": -1, "pair_waves": 0, "pair_kernel": 0}.get(key, 1))
    S.energy()
    util.assert_counts(S.observables, g, False, label=f"{key}={value}")
    util.assert_energies(S.observables, g, False, label=f"{key}={value}")
    assert util.max_rel(S.dipoles()[0].reshape(-1), g["mu"]) < util.REL_TOL
    S.close()


def test_a_failed_wait_does_not_lock_the_context():
    """round-4 advisor: an error inside mpmc_energy_wait left `pending` set and every later enqueue was refused ("still in flight")."""
    atoms, basis, opts = util.load_fixture("ion216_polar")
    S = energy.System(atoms, basis, opts)
    e0 = S.energy()
    S.configure("fail_next_wait", 1)  # the next wait fails as if the runtime had refused it
    with pytest.raises(energy.MpmcError):
        S.energy()
    assert S.energy() == e0  # the context evaluates again, same bits
    S.energy_async()
    S.configure("fail_next_wait", 1)
    with pytest.raises(energy.MpmcError):
        S.energy_wait()
    assert S.energy() == e0
    S.close()
