"""ranks.py -- the rank processes of one multi-GPU job, with ONE ROCm each: plain child processes and a loopback socket between them.

The reference is one C++ executable under mpirun: every rank calls MPI_Init, asks MPI_Comm_rank / MPI_Comm_size and meets the others in
MPI_Allgather / MPI_Barrier (reference src/main.cpp, src/args_etc.h:153-186; PathIntegral.cpp:757-768).  What MPI is to it, this module is to
a host program of the C ABI that does not want a second GPU runtime in its ranks (importing PyTorch brings its own bundled ROCm):

  spawn(n, argv)  -- `python3 bench.py --gpus N` started bare: N fresh children (subprocess.Popen, never an exec, before anything touched
                     the GPU), each with RANK / LOCAL_RANK / WORLD_SIZE and the path of the rendezvous file;
  Hub.join(...)   -- the job's out-of-band channel.  Rank 0 listens on an ephemeral loopback port and publishes it in the rendezvous file;
                     the others connect.  One primitive, `exchange(obj)` = all-gather of a JSON-able object through rank 0; barrier, max,
                     broadcast and the host fall-back of the bead combine are built on it.  Python floats travel as their shortest
                     round-trip repr, i.e. exactly.

The DATA path of the job is not here: the 4 fp64 per bead go through ncclAllGather inside libmpmc_energy.so (mpmc_pi_gather_beads); the hub
carries RCCL's 128-byte unique id to the ranks, the votes around the communicator's initialisation and the timing barriers.
Single node only (loopback), like the bench contract.
"""
from __future__ import annotations

import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import time
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

RDZV_ENV = "MPMC_RDZV_FILE"


def env_rank() -> Tuple[int, int, int]:
    """(rank, world, local_rank) as a launcher exported them (torch.distributed.run, mpirun's OMPI_* / PMI_* or spawn() below)."""
    e = os.environ
    world = int(e.get("WORLD_SIZE") or e.get("OMPI_COMM_WORLD_SIZE") or e.get("PMI_SIZE") or 1)
    rank = int(e.get("RANK") or e.get("OMPI_COMM_WORLD_RANK") or e.get("PMI_RANK") or 0)
    local = int(e.get("LOCAL_RANK") or e.get("OMPI_COMM_WORLD_LOCAL_RANK") or rank)
    return rank, world, local


def rendezvous_file() -> str:
    """where rank 0 publishes the hub's port: the file spawn() named, or -- under somebody else's launcher -- a name every rank of THAT
    launch derives alike (its MASTER_ADDR / MASTER_PORT / run id; the port itself belongs to the launcher's own store and is not touched)."""
    p = os.environ.get(RDZV_ENV)
    if p:
        return p
    e = os.environ
    key = "_".join(str(e.get(k, "x")) for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")).replace("/", "-").replace(":", "-")
    return os.path.join(tempfile.gettempdir(), f"mpmc_rdzv_{os.getuid()}_{key}.json")


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn(n: int, argv: Sequence[str], env_extra: Optional[dict] = None, grace_s: float = 15.0) -> int:
    """start n ranks of `argv` as children of this process and wait for them; returns the job's exit code (0 only if every rank returned
    0).  The children inherit stdout / stderr: rank 0's JSON line is this command's JSON line.  When a rank fails the others get `grace_s`
    to notice (their hub connection breaks) and are then terminated by PID -- never by pattern."""
    rdzv_dir = tempfile.mkdtemp(prefix="mpmc_rdzv_")
    env = dict(os.environ)
    env.update(env_extra or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks of one node (RCCL; see bench.py)
    env.update({RDZV_ENV: os.path.join(rdzv_dir, "hub.json"), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                "MASTER_PORT": str(free_port())})  # (MASTER_*: only a rank that opts into torch.distributed reads them)
    procs: List[subprocess.Popen] = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen(list(argv), env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
        codes: List[Optional[int]] = [None] * n
        failed_at = None
        while any(c is None for c in codes):
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = time.time()
            if failed_at is not None and time.time() - failed_at > grace_s:
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                        try:
                            codes[r] = p.wait(10)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[r] = p.wait()
            time.sleep(0.02)
        bad = [c for c in codes if c]
        return 0 if not bad else (bad[0] if bad[0] > 0 else 128 - bad[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            for f in os.listdir(rdzv_dir):
                os.remove(os.path.join(rdzv_dir, f))
            os.rmdir(rdzv_dir)
        except OSError:
            pass


def _send(sock: socket.socket, obj: Any):
    b = json.dumps(obj).encode()
    sock.sendall(struct.pack("!I", len(b)) + b)


def _recv(sock: socket.socket) -> Any:
    def exactly(k: int) -> bytes:
        buf = bytearray()
        while len(buf) < k:
            part = sock.recv(k - len(buf))
            if not part:
                raise ConnectionError("a rank of the job closed its hub connection")
            buf += part
        return bytes(buf)

    (k,) = struct.unpack("!I", exactly(4))
    return json.loads(exactly(k).decode())


class Hub:
    """all-gather of small JSON-able objects among the ranks of one node, through rank 0.  Collective: every rank calls the same methods
    in the same order (like the MPI calls of the reference).  A peer that dies breaks the connection and every rank raises."""

    def __init__(self, rank: int, world: int, socks: List[socket.socket], listener: Optional[socket.socket], path: Optional[str]):
        self.rank, self.world = rank, world
        self._socks, self._listener, self._path = socks, listener, path

    @staticmethod
    def join(rank: int, world: int, path: Optional[str] = None, timeout_s: float = 180.0, io_timeout_s: float = 900.0) -> "Hub":
        path = path or rendezvous_file()
        if world == 1:
            return Hub(0, 1, [], None, None)
        deadline = time.time() + timeout_s
        if rank == 0:
            lst = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            lst.bind(("127.0.0.1", 0))
            lst.listen(world)
            token = os.urandom(8).hex()
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump({"port": lst.getsockname()[1], "pid": os.getpid(), "token": token, "world": world, "created": time.time()}, f)
            os.replace(tmp, path)  # (atomic: a reader sees the old file, if any, or this one -- and checks pid and token)
            conns: List[Optional[socket.socket]] = [None] * world
            while any(c is None for c in conns[1:]):
                lst.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = lst.accept()
                except socket.timeout:
                    lst.close()
                    raise TimeoutError(f"rank 0: {sum(c is None for c in conns[1:])} of {world - 1} ranks did not reach the hub within {timeout_s:.0f} s")
                c.settimeout(10.0)
                try:
                    hello = _recv(c)
                except (OSError, ValueError):
                    c.close()
                    continue
                r = hello.get("rank", -1) if isinstance(hello, dict) else -1
                if not isinstance(hello, dict) or hello.get("token") != token or not 1 <= r < world or conns[r] is not None:
                    c.close()  # a stranger, or a rank of another job that read a stale file
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(io_timeout_s)
                conns[r] = c
            for c in conns[1:]:
                _send(c, {"ok": True})
            return Hub(0, world, [c for c in conns[1:] if c is not None], lst, path)
        last = "the rendezvous file never appeared"
        while time.time() < deadline:
            try:
                with open(path) as f:
                    info = json.load(f)
                if info.get("world") != world:
                    raise ValueError("the rendezvous file belongs to a job of another size")
                os.kill(int(info["pid"]), 0)  # rank 0 of a finished job is gone: a stale file
                s = socket.create_connection(("127.0.0.1", int(info["port"])), timeout=5.0)
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _send(s, {"rank": rank, "token": info["token"]})
                s.settimeout(max(1.0, deadline - time.time()))
                if _recv(s).get("ok"):
                    s.settimeout(io_timeout_s)
                    return Hub(rank, world, [s], None, None)
                s.close()
            except (OSError, ValueError, KeyError, ConnectionError) as e:
                last = f"{type(e).__name__}: {e}"
            time.sleep(0.05)
        raise TimeoutError(f"rank {rank}: no hub at {path} within {timeout_s:.0f} s ({last})")

    # -- the one primitive ---------------------------------------------------------------------------------------------
    def exchange(self, obj: Any) -> List[Any]:
        """every rank's `obj`, in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            everything = [obj] + [_recv(s) for s in self._socks]
            for s in self._socks:
                _send(s, everything)
            return everything
        _send(self._socks[0], obj)
        return _recv(self._socks[0])

    def barrier(self):
        self.exchange(None)

    def max(self, x: float) -> float:
        return max(self.exchange(float(x)))

    def min(self, x: float) -> float:
        return min(self.exchange(float(x)))

    def bcast(self, obj: Any, src: int = 0) -> Any:
        return self.exchange(obj if self.rank == src else None)[src]

    def allgather_f64(self, local: np.ndarray) -> np.ndarray:
        """(count,) fp64 of this rank -> (world, count), bit for bit (the host fall-back of the bead combine: what the reference's
        MPI_Allgather does, PathIntegral.cpp:763-766)."""
        a = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
        rows = self.exchange(a.tobytes().hex())
        return np.stack([np.frombuffer(bytes.fromhex(h), dtype=np.float64) for h in rows])

    def gather_beads(self, local: np.ndarray) -> np.ndarray:
        """(n_local, stride) of this rank's beads -> (P, stride) in bead order (bead s = rank s % world, slot s // world): the same
        contract as energy.Comm.gather_beads, over the hub."""
        a = np.ascontiguousarray(local, dtype=np.float64)
        n_local = a.shape[0]
        g = self.allgather_f64(a).reshape((self.world, n_local) + a.shape[1:])
        return np.ascontiguousarray(np.swapaxes(g, 0, 1).reshape((self.world * n_local,) + a.shape[1:]))

    n_ranks = property(lambda self: self.world)

    def close(self):
        for s in self._socks:
            try:
                s.close()
            except OSError:
                pass
        self._socks = []
        if self._listener is not None:
            self._listener.close()
            self._listener = None
        if self._path:
            try:
                os.remove(self._path)
            except OSError:
                pass
            self._path = None


def join_rccl_communicator(hub: Hub, ready_here: bool, make_unique_id: Callable[[], bytes], make_comm: Callable[[bytes], Any], timeout_s: float,
                           comm_error: type, log=None):
    """The RCCL communicator of the C ABI for the hub's ranks, or None on EVERY rank.  Returns (comm, a_thread_is_stuck, why_not).

    Rank 0 makes RCCL's unique id (mpmc_comm_unique_id), the hub carries its 128 bytes, every rank joins (mpmc_comm_init_rank).  Three votes
    keep the ranks together: (1) what a rank does on its own first (device index valid, RCCL symbols resolved) -- a rank that failed there
    alone would leave the others inside the collective initialisation; (2) ncclCommInitRank and a probe all-gather BLOCK until every rank is
    in them, and on a node where that never happens the job would hang: they run on a helper thread, and a rank that is not through after
    `timeout_s` votes no; (3) the outcome is a MIN over ranks: all ranks use the communicator or none does.  A rank whose helper thread is
    still inside RCCL must not be timed (any_stuck): the caller reports the run as degraded and leaves without running destructors.
    `make_unique_id`, `make_comm`, `comm_error` are parameters so that the votes can be exercised without RCCL (tests/test_ranks.py)."""
    import threading

    rank, world = hub.rank, hub.world
    log = log or (lambda msg: print(msg, file=sys.stderr, flush=True))
    uid = None
    if hub.min(1 if ready_here else 0) == 1:
        mine = None
        if rank == 0:
            try:
                mine = make_unique_id().hex()
            except comm_error:
                mine = None
        uid = hub.bcast(mine)
    comm, stuck, ok, why = None, False, 1, ""
    if uid is None:
        ok, why = 0, "RCCL could not be opened below the host program on every rank"
    else:
        box = {}

        def join_ranks():
            try:
                c = make_comm(bytes.fromhex(uid))
                probe = c.allgather(np.array([float(rank)]))
                box["probe_ok"] = bool(np.array_equal(np.asarray(probe).reshape(-1), np.arange(world, dtype=np.float64)))
                box["comm"] = c
            except comm_error as e:
                box["err"] = str(e)

        th = threading.Thread(target=join_ranks, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            stuck, ok = True, 0
            log(f"[rank {rank}] the C-ABI communicator did not come up within {timeout_s:.0f} s")
        elif "err" in box:
            ok = 0
            log(f"[rank {rank}] mpmc_comm_init_rank failed ({box['err']})")
        else:
            comm = box["comm"]
            if not box["probe_ok"]:
                ok = 0
                log(f"[rank {rank}] the probe all-gather over the C-ABI communicator returned the wrong ranks")
    all_ok = hub.min(ok) == 1
    any_stuck = hub.max(1 if stuck else 0) == 1
    if not all_ok:
        if comm is not None and not any_stuck:
            comm.close()  # (with a rank still inside the collective initialisation the communicator is left alone: destroying it can block too)
        comm = None
        if any_stuck:
            why = f"the C-ABI communicator timed out after {timeout_s:.0f} s on some rank"
        elif not why:
            why = "mpmc_comm_init_rank or its probe all-gather failed on some rank"
    return comm, any_stuck, why
