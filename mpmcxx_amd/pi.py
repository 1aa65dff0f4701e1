"""Path-integral per-bead energy loop -- host-side mirror of
``SimulationControl::PI_calculate_potential`` (reference src/SimulationControl.PathIntegral.cpp:752-805).

The reference evaluates ``systems[s]->energy()`` for the P Trotter beads (one OpenMP thread or one MPI rank per
bead), all-gathers four doubles per bead (``MPI_Allgather`` x4, :763-766), sums them in bead order s = 0..P-1 and
divides by P (:786-801).  Here the beads are independent device contexts sharded over ranks (bead s lives on rank
``s % world``, local slot ``s // world``); the exchange is ONE collective of 4 fp64 per bead over
``torch.distributed`` (backend nccl == RCCL over xGMI on the GPU node, gloo in the CPU tests).

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
            device: Optional[str] = None) -> Tuple[float, np.ndarray]:
    """per_bead_local: (n_local, 4) = {rd, coulombic, polarization, vdw} of this rank's beads, in local-slot order.
    Returns (V, obs4) exactly as PI_calculate_potential: obs = ordered sum / P, V = rd + coulombic + vdw + polarization."""
    per_bead_local = np.ascontiguousarray(per_bead_local, dtype=np.float64).reshape(-1, 4)
    n_local = per_bead_local.shape[0]
    if world == 1:
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
                           mode: str = "gather", device: Optional[str] = None) -> Tuple[float, np.ndarray]:
    """local_eval() -> (n_local, 4) per-bead energies of this rank (HIP path: energy.pi_potential_local)."""
    return combine(local_eval(), P, rank, world, group, mode, device)
