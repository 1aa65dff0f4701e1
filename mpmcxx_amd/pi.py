"""Path-integral per-bead energy loop -- host-side mirror of
``SimulationControl::PI_calculate_potential`` (reference src/SimulationControl.PathIntegral.cpp:752-805).

The reference evaluates ``systems[s]->energy()`` for the P Trotter beads (one OpenMP thread or one MPI rank per
bead), all-gathers four doubles per bead (``MPI_Allgather`` x4, :763-766), sums them in bead order s = 0..P-1 and
divides by P (:786-801).  Here the beads are independent device contexts sharded over ranks (bead s lives on rank
``s % world``, local slot ``s // world``); the exchange is ONE all-gather of 4 fp64 per bead: through ``comm`` -- an
``energy.Comm`` (ncclAllGather inside libmpmc_energy.so over RCCL / xGMI: the production path, no torch in the rank) or a
``ranks.Hub`` (loopback sockets: rehearsal and fall-back) -- or, opt-in, over ``torch.distributed`` (nccl / gloo; the CPU tests).

The kinetic half of the estimator (``PI_calculate_kinetic``, :806-824) needs the centres of mass of ADJACENT images of
every molecule (``PI_chain_mass_length2``, :908-965).  The reference keeps all P images on every MPI rank; here the ring
of images crosses ranks, so ``pi_calculate_kinetic`` all-gathers 3 fp64 per molecule and bead once per call.

``mode="gather"`` (default) all-gathers the per-bead values and sums them in bead order, which reproduces the
reference's summation order bit for bit on every rank; ``mode="reduce"`` all-reduces the rank-local partial sums
(one 32-byte message, order differs in the last bit).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def beads_of_rank(P: int, rank: int, world: int) -> List[int]:
    """global bead indices owned by `rank` (round-robin)."""
    return list(range(rank, P, world))


def combine(per_bead_local: np.ndarray, P: int, rank: int = 0, world: int = 1, group=None, mode: str = "gather",
            device: Optional[str] = None, comm=None) -> Tuple[float, np.ndarray]:
    """per_bead_local: (n_local, 4) = {rd, coulombic, polarization, vdw} of this rank's beads, in local-slot order.
    Returns (V, obs4) exactly as PI_calculate_potential: obs = ordered sum / P, V = rd + coulombic + vdw + polarization.
    comm: an `energy.Comm` (RCCL communicator of the C ABI): the exchange is then ONE ncclAllGather inside libmpmc_energy.so
    (mpmc_pi_gather_beads); or a `ranks.Hub` (same `gather_beads` contract over loopback sockets).  torch.distributed is not involved."""
    per_bead_local = np.ascontiguousarray(per_bead_local, dtype=np.float64).reshape(-1, 4)
    n_local = per_bead_local.shape[0]
    if comm is not None:
        if comm.n_ranks * n_local != P:
            raise ValueError("every rank must own P / n_ranks beads")
        all_beads = comm.gather_beads(per_bead_local)
    elif world == 1:
        all_beads = per_bead_local
    else:
        import torch
        import torch.distributed as dist

        if P % world:
            raise ValueError("P must be a multiple of the number of ranks")
        if n_local != P // world:
            raise ValueError("every rank must own P / world beads")
        dev = device or "cpu"
        mine = torch.from_numpy(per_bead_local.copy()).to(dev)
        if mode == "reduce":
            part = mine.sum(dim=0)
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group)
            s = part.cpu().numpy()
            obs = s / P
            return float(obs[0] + obs[1] + obs[3] + obs[2]), obs
        gathered = torch.empty((world, n_local, 4), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered.view(world * n_local, 4), mine, group=group)
        g = gathered.cpu().numpy()  # [rank][slot] -> bead = slot * world + rank
        all_beads = np.empty((P, 4))
        for r in range(world):
            for slot in range(n_local):
                all_beads[slot * world + r] = g[r, slot]
    obs = np.zeros(4)
    for s in range(all_beads.shape[0]):  # ordered accumulation, reference :791-796
        obs += all_beads[s]
    obs /= P  # :798-801
    return float(obs[0] + obs[1] + obs[3] + obs[2]), obs  # :803-804


def pi_calculate_potential(local_eval: Callable[[], np.ndarray], P: int, rank: int = 0, world: int = 1, group=None,
                           mode: str = "gather", device: Optional[str] = None, comm=None) -> Tuple[float, np.ndarray]:
    """local_eval() -> (n_local, 4) per-bead energies of this rank (HIP path: energy.pi_potential_local)."""
    return combine(local_eval(), P, rank, world, group, mode, device, comm)


def molecule_coms(pos: np.ndarray, mass: np.ndarray, mol_id: np.ndarray, frozen: np.ndarray):
    """Molecule::update_COM for every molecule of one image (reference src/Molecule.cpp:259-281), accumulated in atom
    order like the reference's list walk.  Returns (com (n_molecules, 3), mol_mass, movable): movable[m] = image 0's
    molecule m counts in System::countN (src/System.cpp:909-931), i.e. its LAST atom row is not frozen (the reader overwrites
    molecule->frozen on every row, :684)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    mass = np.asarray(mass, dtype=np.float64)
    mol_id = np.asarray(mol_id)
    n = len(mass)
    first = np.flatnonzero(np.r_[True, mol_id[1:] != mol_id[:-1]]) if n else np.zeros(0, dtype=int)
    seg = np.cumsum(np.r_[True, mol_id[1:] != mol_id[:-1]]) - 1 if n else np.zeros(0, dtype=int)
    nmol = len(first)
    m = np.zeros(nmol)
    c = np.zeros((nmol, 3))
    np.add.at(m, seg, mass)  # unbuffered, in atom order
    np.add.at(c, seg, mass[:, None] * pos)
    last = np.r_[first[1:] - 1, n - 1] if n else np.zeros(0, dtype=int)
    return c / m[:, None], m, (np.asarray(frozen)[last] == 0).astype(np.int32)


def gather_beads(local: np.ndarray, P: int, rank: int = 0, world: int = 1, group=None, device: Optional[str] = None, comm=None) -> np.ndarray:
    """local: (n_local, ...) values of this rank's beads in local-slot order -> (P, ...) in bead order (bead = slot * world + rank).
    comm: an `energy.Comm` or `ranks.Hub` (their `gather_beads`); else torch.distributed."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if comm is not None:
        if comm.n_ranks * local.shape[0] != P:
            raise ValueError("every rank must own P / n_ranks beads")
        return comm.gather_beads(local)
    if world == 1:
        return local
    import torch
    import torch.distributed as dist

    n_local = local.shape[0]
    if P % world or n_local != P // world:
        raise ValueError("every rank must own P / world beads")
    dev = device or "cpu"
    mine = torch.from_numpy(local.reshape(n_local, -1).copy()).to(dev)
    gathered = torch.empty((world * n_local, mine.shape[1]), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    g = gathered.cpu().numpy().reshape((world, n_local) + local.shape[1:])
    return np.ascontiguousarray(np.swapaxes(g, 0, 1).reshape((P,) + local.shape[1:]))


def pi_calculate_kinetic(local_coms: np.ndarray, mol_mass: np.ndarray, movable: np.ndarray, P: int, temperature: float,
                         rank: int = 0, world: int = 1, group=None, device: Optional[str] = None, comm=None) -> Tuple[float, float]:
    """SimulationControl::PI_calculate_kinetic (reference PathIntegral.cpp:806-824).  local_coms: (n_local, n_molecules, 3)
    from `molecule_coms` of this rank's beads.  Returns (K [Kelvin], chain_mass_len2 [kg m^2])."""
    from . import energy as _e

    coms = gather_beads(local_coms, P, rank, world, group, device, comm)
    chain = _e.pi_chain_mass_length2(coms, mol_mass, movable)
    N = float(np.count_nonzero(movable))
    return _e.pi_kinetic(chain, N, P, temperature), chain


def pi_calculate_energy(kinetic: float, potential: float) -> float:
    """SimulationControl::PI_calculate_energy (PathIntegral.cpp:734-749): estimator = kinetic + potential."""
    return kinetic + potential
