"""GPU: the cross-GPU combine of PI_calculate_potential on RCCL, below Python (include/mpmc_energy.h: mpmc_comm_*, mpmc_pi_gather_beads,
mpmc_pi_allreduce).  Reference: 4 x MPI_Allgather + ordered sum (PathIntegral.cpp:763-766, :786-801).

A one-GPU box can only form communicators of one rank (RCCL: one rank per device) -- those tests run everywhere and prove that RCCL is
loaded, initialised and carries the bytes; the tests that need two devices skip cleanly below that and run on a multi-GPU node."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util
from mpmcxx_amd import energy, ranks

# No torch import in this file (round 4 needed one, FIRST, to dodge an exit-time abort): the library now shares an RCCL the host program
# already mapped and otherwise opens, RTLD_LOCAL, the copy next to the HIP runtime it is bound to (csrc/comm.cpp rccl();
# tests/test_ranks.py replays both load orders in child processes).

pytestmark = pytest.mark.gpu


def make_beads(n, devices=(0,), name="ion216_polar"):
    atoms, basis, opts = util.load_fixture(name)
    out = []
    for b in range(n):
        rng = np.random.default_rng(100 + b)
        out.append(energy.System(dict(atoms, pos=atoms["pos"] + rng.normal(scale=0.05, size=atoms["pos"].shape)), basis, opts, device=devices[b % len(devices)]))
    return out


def test_rccl_is_loaded_below_python():
    assert energy.rccl_version() >= 20000


def test_pi_allreduce_on_one_device_equals_the_local_loop():
    beads = make_beads(4)
    s_local, per_local, f_local = energy.pi_potential_local(beads)
    assert energy.pi_allreduce_info(beads)[0] == 1  # (one device; the communicator may exist already from an earlier test of this process)
    s_rccl, per_rccl, f_rccl = energy.pi_allreduce(beads)  # values travel device -> ncclAllGather (1 rank) -> host
    assert energy.pi_allreduce_info(beads) == (1, 1)  # ABI 5: one device, the process-wide communicator of that device has one rank
    assert np.array_equal(s_local, s_rccl) and f_local == f_rccl
    assert [p["energy"] for p in per_local] == [p["energy"] for p in per_rccl]
    from oracle import pi_aggregate

    v, obs = energy.pi_finish(s_rccl, 4)
    v_ref, obs_ref = pi_aggregate([p["rd_energy"] for p in per_rccl], [p["coulombic_energy"] for p in per_rccl], [p["polarization_energy"] for p in per_rccl])
    assert v == v_ref and np.array_equal(obs, obs_ref)
    for b in beads:
        b.close()


def test_single_rank_communicator_round_trip():
    uid = energy.Comm.unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    cm = energy.Comm(1, 0, uid, 0)
    x = np.arange(12, dtype=np.float64).reshape(3, 4) * np.pi
    assert np.array_equal(cm.gather_beads(x), x)
    assert np.array_equal(cm.allgather(x.reshape(-1)), x.reshape(1, -1))
    big = np.random.default_rng(0).normal(size=(2, 3 * 1000))  # centres of mass of the kinetic estimator: buffers grow
    assert np.array_equal(cm.gather_beads(big), big)
    cm.close()


def test_thread_per_device_path_on_virtual_devices_of_one_gpu():
    """mpmc_pi_allreduce with G > 1: one host thread per device evaluates that device's beads, the per-bead values are gathered and summed
    in bead order.  RCCL admits one rank per physical device, so on a one-GPU box the beads are given VIRTUAL devices (test hook
    "virtual_device": same worker threads, same hand-off, same ordered combine; the gather is a host copy) -- repeated, with 3 "devices"
    and 7 beads (uneven shares), against the one-thread local loop."""
    beads = make_beads(7, name="ion216_polar")
    s_local, per_local, f_local = energy.pi_potential_local(beads)
    for k, b in enumerate(beads):
        b.configure("virtual_device", k % 3)
    assert energy.pi_allreduce_info(beads)[0] == 3
    for _ in range(5):
        s_thr, per_thr, f_thr = energy.pi_allreduce(beads)
        assert np.array_equal(s_local, s_thr) and f_local == f_thr
        assert [p["energy"] for p in per_local] == [p["energy"] for p in per_thr]
    assert energy.pi_allreduce_info(beads) == (3, 3)
    for b in beads:
        b.close()


def test_two_devices_one_process_bead_b_on_device_b_mod_g():
    if energy.device_count() < 2:
        pytest.skip("needs two GPUs (one process driving several devices: ncclCommInitAll)")
    one = make_beads(4, devices=(0,))
    s1, per1, _ = energy.pi_allreduce(one)
    two = make_beads(4, devices=(0, 1))
    s2, per2, _ = energy.pi_allreduce(two)
    assert np.array_equal(s1, s2) and [p["energy"] for p in per1] == [p["energy"] for p in per2]
    for b in one + two:
        b.close()


def last_json(txt):
    return json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])


def test_two_ranks_on_two_gpus_over_rccl():
    """bench.py's multi-rank path with nccl (= RCCL) on one rank per GPU, combine inside libmpmc_energy.so"""
    if energy.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL: one rank per device)")
    args = ["--beads", "4", "--natoms", "1000", "--steps", "2", "--warmup", "1", "--cpu-baseline", "none", "--no-extra-passes"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + args, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    outs = {}
    for impl in ("cabi", "torch"):
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(ranks.free_port()),
                              "bench.py", "--gpus", "2", "--combine-impl", impl] + args, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert two.returncode == 0, two.stderr[-2000:]
        outs[impl] = last_json(two.stdout)
    a = last_json(one.stdout)
    for impl, b in outs.items():
        assert b["n_gpus"] == 2 and b["config"]["world_size"] == 2 and "cabi_comm_failed" not in b
        assert sorted(r["device"] for r in b["config"]["ranks"]) == ["hip:0", "hip:1"]
        assert a["V_mean_K"] == b["V_mean_K"] and a["obs_rd_es_pol_vdw"] == b["obs_rd_es_pol_vdw"], impl
    assert "mpmc_pi_gather_beads" in outs["cabi"]["config"]["combine_impl"]
    assert all(r["comm_n_ranks"] == 2 and not r["torch_imported"] for r in outs["cabi"]["config"]["ranks"])
