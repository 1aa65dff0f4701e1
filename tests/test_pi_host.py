"""CPU: the path-integral bead loop's host logic (bead->rank sharding, the 4-scalar exchange, ordered mean),
with the ORACLE standing in for the per-bead evaluator (tests may use it).  world_size 2 over gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from mpmcxx_amd import pi
from oracle import OracleSystem, pi_aggregate

P = 4


def bead_atoms(base, b):
    rng = np.random.default_rng(100 + b)
    a = dict(base)
    a["pos"] = base["pos"] + rng.normal(scale=0.05, size=base["pos"].shape)
    return a


def per_bead_reference():
    atoms, basis, opts = util.load_fixture("ion64_es")
    vals = []
    for b in range(P):
        r = OracleSystem(bead_atoms(atoms, b), basis, opts).energy(want_atoms=False)
        vals.append([r["rd_energy"], r["coulombic_energy"], r["polarization_energy"], r["vdw_energy"]])
    return np.array(vals)


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        atoms, basis, opts = util.load_fixture("ion64_es")

        def local_eval():
            vals = []
            for b in pi.beads_of_rank(P, rank, world):
                r = OracleSystem(bead_atoms(atoms, b), basis, opts).energy(want_atoms=False)
                vals.append([r["rd_energy"], r["coulombic_energy"], r["polarization_energy"], r["vdw_energy"]])
            return np.array(vals)

        v, obs = pi.pi_calculate_potential(local_eval, P, rank, world, mode=mode)
        out[rank] = (v, obs.tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["gather", "reduce"])
def test_pi_potential_world2_gloo(mode):
    ref = per_bead_reference()
    v_ref, obs_ref = pi_aggregate(ref[:, 0], ref[:, 1], ref[:, 2], ref[:, 3])
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 500) + (0 if mode == "gather" else 1)
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    assert set(out.keys()) == {0, 1}
    for rank in (0, 1):
        v, obs = out[rank]
        if mode == "gather":  # ordered sum: bit-identical to the reference's s = 0..P-1 loop
            assert v == v_ref and obs == obs_ref.tolist()
        else:
            assert util.close(v, v_ref, 1e-14)
    assert out[0] == out[1]


def test_sharding_is_round_robin_and_complete():
    for world in (1, 2, 4, 8):
        owned = [pi.beads_of_rank(32, r, world) for r in range(world)]
        assert sorted(b for o in owned for b in o) == list(range(32))
        assert all(len(o) == 32 // world for o in owned)
        assert all(b % world == r for r, o in enumerate(owned) for b in o)


def test_combine_single_rank_matches_reference_order():
    ref = per_bead_reference()
    v, obs = pi.combine(ref, P)
    v_ref, obs_ref = pi_aggregate(ref[:, 0], ref[:, 1], ref[:, 2], ref[:, 3])
    assert v == v_ref and np.array_equal(obs, obs_ref)
