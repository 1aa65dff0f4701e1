"""CPU: the torch-free rank plumbing of the multi-GPU job (mpmcxx_amd/ranks.py) -- child-process spawner, loopback socket hub, the votes
around the C-ABI communicator -- and the load order of the ROCm libraries.  Reference: one executable under mpirun, MPI_Init /
MPI_Comm_rank / MPI_Allgather (src/args_etc.h:153-186, PathIntegral.cpp:757-768)."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

import util
from mpmcxx_amd import pi, ranks


def _hub_worker(rank, world, path, case, q):
    try:
        hub = ranks.Hub.join(rank, world, path, timeout_s=30)
        out = {}
        out["exchange"] = hub.exchange({"rank": rank, "x": 0.1 * (rank + 1)})
        out["max"] = hub.max(1.5 + rank)
        out["bcast"] = hub.bcast("from0" if rank == 0 else "ignored")
        local = np.array([[np.pi * (rank + 1), -1.0 / 3.0, 1e-300 * rank, float(s)] for s in range(2)])  # slot-major rows of this rank
        g = hub.gather_beads(local)
        out["gathered"] = g.tobytes().hex()
        v, obs = pi.combine(local, 2 * world, rank, world, comm=hub)
        out["v"], out["obs"] = v, obs.tolist()
        coms = np.arange(2 * 5 * 3, dtype=np.float64).reshape(2, 5, 3) + 1000.0 * rank  # centres of mass of the kinetic estimator: (n_local, n_molecules, 3)
        allc = pi.gather_beads(coms, 2 * world, rank, world, comm=hub)
        out["coms_ok"] = bool(allc.shape == (2 * world, 5, 3) and all(np.array_equal(allc[s], np.arange(15, dtype=np.float64).reshape(5, 3) + 15.0 * (s // world) + 1000.0 * (s % world))
                                                                        for s in range(2 * world)))
        hub.barrier()
        hub.close()
        q.put((rank, out))
    except Exception as e:  # noqa: BLE001
        q.put((rank, {"error": repr(e)}))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_hub_all_gathers_exactly_and_in_rank_order(world):
    path = os.path.join(tempfile.mkdtemp(), "hub.json")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_hub_worker, args=(r, world, path, "plain", q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=60) for _ in range(world))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert not any("error" in r for r in res.values()), res
    for r in range(world):
        assert [e["rank"] for e in res[r]["exchange"]] == list(range(world))
        assert [e["x"] for e in res[r]["exchange"]] == [0.1 * (k + 1) for k in range(world)]  # floats travel exactly
        assert res[r]["max"] == 1.5 + world - 1 and res[r]["bcast"] == "from0" and res[r]["coms_ok"]
        assert res[r] == res[0] or {k: v for k, v in res[r].items()} == {k: v for k, v in res[0].items()}
    # bead order: bead s = rank s % world, slot s // world; ordered sum identical on every rank and equal to the serial loop
    g = np.frombuffer(bytes.fromhex(res[0]["gathered"]), dtype=np.float64).reshape(2 * world, 4)
    expect = np.array([[np.pi * (s % world + 1), -1.0 / 3.0, 1e-300 * (s % world), float(s // world)] for s in range(2 * world)])
    assert np.array_equal(g, expect)
    obs = np.zeros(4)
    for s in range(2 * world):
        obs += expect[s]
    obs /= 2 * world
    assert res[0]["obs"] == obs.tolist() and not os.path.exists(path)  # rank 0 removed the rendezvous file


def test_a_stale_rendezvous_file_is_ignored():
    """a file left by a finished job (dead pid, dead port) must not capture the ranks of the next one with the same name."""
    path = os.path.join(tempfile.mkdtemp(), "hub.json")
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    with open(path, "w") as f:
        json.dump({"port": ranks.free_port(), "pid": dead.pid, "token": "old", "world": 2, "created": time.time() - 100}, f)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p1 = ctx.Process(target=_hub_worker, args=(1, 2, path, "plain", q))
    p1.start()
    time.sleep(1.0)  # rank 1 is polling the stale file by now
    p0 = ctx.Process(target=_hub_worker, args=(0, 2, path, "plain", q))
    p0.start()
    res = dict(q.get(timeout=60) for _ in range(2))
    p0.join(30), p1.join(30)
    assert not any("error" in r for r in res.values()), res


def test_spawn_starts_fresh_children_and_relays_the_worst_exit_code():
    code = ("import os, sys\n"
            "from mpmcxx_amd import ranks\n"
            "r, w, l = ranks.env_rank()\n"
            "hub = ranks.Hub.join(r, w, timeout_s=30)\n"
            "pids = hub.exchange(os.getpid())\n"
            "assert len(set(pids)) == w == 3 and 'torch' not in sys.modules\n"
            "hub.barrier(); hub.close()\n"
            "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r and 7 or 0)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", ranks.RDZV_ENV)}
    env["PYTHONPATH"] = util.ROOT
    ok = subprocess.run([sys.executable, "-c", f"import sys; from mpmcxx_amd import ranks; sys.exit(ranks.spawn(3, [sys.executable, '-c', {code!r}]))"],
                        env=env, cwd=util.ROOT, timeout=120)
    assert ok.returncode == 0
    bad = subprocess.run([sys.executable, "-c", f"import sys; from mpmcxx_amd import ranks; sys.exit(ranks.spawn(3, [sys.executable, '-c', {code!r}], {{'FAIL_RANK': '1'}}))"],
                         env=env, cwd=util.ROOT, timeout=120)
    assert bad.returncode == 7


def test_the_parent_of_a_bare_multi_gpu_run_and_its_ranks_stay_torch_free():
    """static: bench.py's parent branch comes before anything loads the HIP library, nothing execs, and torch is imported only under the
    opt-in --combine-impl torch; ranks.py imports no GPU runtime at all."""
    src = open(os.path.join(util.ROOT, "bench.py")).read()
    main = src[src.index("def main"):]
    assert "os.exec" not in src and "execv" not in src
    assert main.index("sys.exit(ranks.spawn(") < main.index("from mpmcxx_amd import energy")
    assert main.count("import torch") == 1 and main.index("if use_torch:") < main.index("import torch") < main.index("from mpmcxx_amd import energy")
    rsrc = open(os.path.join(util.ROOT, "mpmcxx_amd", "ranks.py")).read()
    assert "import torch" not in rsrc and "energy" not in [ln.split()[-1] for ln in rsrc.splitlines() if ln.startswith(("import ", "from "))]


def test_bare_multi_gpu_command_starts_ranks_and_relays_their_exit_code():
    """`python3 bench.py --gpus 2` with no launcher: the parent starts two children itself; without a GPU they stop at "no HIP device",
    which is what proves that they were started, and the parent relays a non-zero code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", ranks.RDZV_ENV)}
    args = ["--gpus", "2", "--combine-impl", "hub", "--beads", "4", "--natoms", "1000", "--steps", "1", "--warmup", "0", "--cpu-baseline", "none"]
    from mpmcxx_amd import energy

    if energy.device_count() < 2:
        args += ["--force-device", "0"]
    p = subprocess.run([sys.executable, "bench.py"] + args, cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert "starting 2 ranks" in p.stderr, p.stderr[-2000:]
    if energy.device_count() < 1:
        assert p.returncode != 0 and "no HIP device visible" in p.stderr  # the ranks' own reason, not an argument check of the parent
    else:
        assert p.returncode == 0, p.stderr[-2000:]


# ---- the votes around the C-ABI communicator (ranks.join_rccl_communicator), two hub ranks, RCCL replaced by stand-ins -------------------
class _CommError(Exception):
    pass


class _StandInComm:
    def __init__(self, world):
        self.world, self.closed = world, False

    def allgather(self, local):
        return np.arange(self.world, dtype=np.float64).reshape(self.world, 1)

    def close(self):
        self.closed = True


def _vote_worker(rank, world, path, case, q):
    hub = ranks.Hub.join(rank, world, path, timeout_s=30)
    made = {}

    def make_comm(uid):
        assert uid == b"id-of-rank-0"
        if case == "rank1_hangs" and rank == 1:
            time.sleep(120)  # (a daemon thread: the worker leaves through os._exit below, as bench.py does)
        if case == "rank1_fails" and rank == 1:
            raise _CommError("no transport")
        made["comm"] = _StandInComm(world)
        return made["comm"]

    ready = not (case == "rank0_not_ready" and rank == 0)
    t0 = time.time()
    comm, stuck, why = ranks.join_rccl_communicator(hub, ready, lambda: b"id-of-rank-0", make_comm, 3.0, _CommError, log=lambda m: None)
    q.put((rank, (comm is not None, stuck, why, made.get("comm").closed if made.get("comm") else None, time.time() - t0)))
    hub.barrier()
    hub.close()
    if case == "rank1_hangs":
        q.close()
        q.join_thread()  # (the queue's feeder thread must have written the result before the process leaves without destructors)
        os._exit(0)


@pytest.mark.parametrize("case", ["all_join", "rank1_hangs", "rank1_fails", "rank0_not_ready"])
def test_every_rank_takes_the_same_route_around_the_cabi_communicator(case):
    path = os.path.join(tempfile.mkdtemp(), "hub.json")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_vote_worker, args=(r, 2, path, case, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=90) for _ in range(2))
    for p in ps:
        p.join(30)
    have = [out[r][0] for r in (0, 1)]
    assert have[0] == have[1]  # all ranks or none
    if case == "all_join":
        assert have == [True, True] and not out[0][1] and out[0][2] == ""
    else:
        assert have == [False, False]
        assert out[0][1] == out[1][1] == (case == "rank1_hangs")
        assert out[0][2] and out[1][2]
        if case == "rank1_hangs":
            assert "timed out" in out[0][2] and out[0][4] < 30 and out[0][3] is False  # rank 0's communicator is left alone, nobody waits for the hung rank
        if case == "rank1_fails":
            assert out[0][3] is True  # rank 0 had joined: its communicator is closed again


# ---- which RCCL, and a clean exit in either load order (round 4: "double free or corruption" in librocm_smi64's destructors) ---------------
@pytest.mark.parametrize("order", ["library_first", "torch_first", "library_only"])
def test_rccl_is_shared_not_duplicated_and_the_process_exits_cleanly(order):
    body = {"library_first": "from mpmcxx_amd import energy\nv = energy.rccl_version()\nimport torch\n",
            "torch_first": "import torch\nfrom mpmcxx_amd import energy\nv = energy.rccl_version()\n",
            "library_only": "from mpmcxx_amd import energy\nv = energy.rccl_version()\n"}[order]
    code = body + "import json\nprint(json.dumps({'v': v, 'path': energy.rccl_library_path(), 'libs': energy.loaded_rocm_libs()}))\n"
    p = subprocess.run([sys.executable, "-c", code], cwd=util.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=util.ROOT))
    assert p.returncode == 0, (p.returncode, p.stderr[-1500:])  # 134 = the exit-time abort
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert info["v"] >= 20000 and info["path"]
    if order == "torch_first":  # the host program's copy is shared: ONE librccl, ONE libamdhip64, ONE librocm_smi64 in the process
        assert "already mapped" in info["path"]
        assert all(len(info["libs"].get(k, [])) == 1 for k in ("libamdhip64", "librccl", "librocm_smi64")), info["libs"]
    if order == "library_only":  # a torch-free rank: the RCCL next to the HIP runtime the library is bound to, one of each
        hip = info["libs"]["libamdhip64"]
        assert len(hip) == 1 and len(info["libs"]["librccl"]) == 1 and "libtorch_hip" not in info["libs"]
        assert os.path.dirname(os.path.realpath(info["libs"]["librccl"][0])) == os.path.dirname(os.path.realpath(hip[0]))
