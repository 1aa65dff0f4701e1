"""Synthetic periodic boxes for the energy hot path -- TEST/BENCH INPUT GENERATOR.

Pure data generation (no physics): writes the reference's own on-disk formats so that the
reference harness (oracle/_ref/ref_harness) and our loaders (mpmcxx_amd/pqr.py) parse the SAME
text and therefore see bit-identical doubles.

PQR token grammar follows reference src/System.cpp:583-687
  ATOM atom_id atom_type molecule_type FLAG molecule_id x y z mass charge[e] alpha eps sigma omega gwp_alpha
Input-file grammar ("keyword value" per line) follows reference src/SimulationControl.cpp:204-267.

Generators (recipes from SURVEY.md Appendix A):
  lattice_box   : jittered simple-cubic lattice of single-site atoms, alternating +-0.1 e
  molecular_box : rigid 3-site molecules (exclusions, intramolecular erf term) + one neutral
                  polarizable atom (exercises the es_excluded "quirk 2")
"""
from __future__ import annotations

import math
import os
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence


@dataclass
class AtomRow:
    atom_id: int
    atomtype: str
    moltype: str
    flag: str  # M movable / F frozen
    mol_id: int
    x: float
    y: float
    z: float
    mass: float
    charge_e: float
    alpha: float
    eps: float
    sigma: float

    def line(self) -> str:
        return (
            f"ATOM {self.atom_id:6d} {self.atomtype:<4s} {self.moltype:<4s} {self.flag} {self.mol_id:6d} "
            f"{self.x:12.6f} {self.y:12.6f} {self.z:12.6f} {self.mass:9.5f} {self.charge_e:9.5f} "
            f"{self.alpha:8.5f} {self.eps:10.5f} {self.sigma:8.5f} 0.00000 0.00000"
        )


def lattice_box(
    n_atoms: int,
    L: float,
    seed: int,
    charged: bool = True,
    alpha: float = 1.6411,
    eps: float = 119.8,
    sigma: float = 3.405,
    charge: float = 0.1,
    frozen_every: int = 0,
) -> List[AtomRow]:
    """Jittered simple-cubic lattice, one atom per molecule, charges +q (odd id) / -q (even id)."""
    n = int(math.ceil(n_atoms ** (1.0 / 3.0) - 1e-9))
    a = L / n
    rng = random.Random(seed)
    rows: List[AtomRow] = []
    k = 0
    for ix in range(n):
        for iy in range(n):
            for iz in range(n):
                if k >= n_atoms:
                    break
                k += 1
                x = (ix + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
                y = (iy + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
                z = (iz + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
                q = (charge if (k % 2 == 1) else -charge) if charged else 0.0
                flag = "F" if (frozen_every and k % frozen_every == 0) else "M"
                rows.append(AtomRow(k, "Ar", "Ar", flag, k, x, y, z, 39.948, q, alpha, eps, sigma))
    return rows


def lattice_box_cell(n_atoms: int, basis, seed: int, charge: float = 0.1, alpha: float = 1.6411, eps: float = 119.8, sigma: float = 3.405) -> List[AtomRow]:
    """Jittered lattice in the FRACTIONAL coordinates of an arbitrary (triclinic) cell, one atom per molecule, charges +-q: atoms fill the
    cell evenly whatever its shape (lattice_box fills a cube, which a skewed cell of another volume wraps onto itself)."""
    n = int(math.ceil(n_atoms ** (1.0 / 3.0) - 1e-9))
    rng = random.Random(seed)
    rows: List[AtomRow] = []
    k = 0
    for ix in range(n):
        for iy in range(n):
            for iz in range(n):
                if k >= n_atoms:
                    break
                k += 1
                f = [(i + 0.5) / n - 0.5 + rng.uniform(-0.1 / n, 0.1 / n) for i in (ix, iy, iz)]
                x, y, z = (sum(f[q] * basis[q][p] for q in range(3)) for p in range(3))
                rows.append(AtomRow(k, "Ar", "Ar", "M", k, x, y, z, 39.948, charge if (k % 2 == 1) else -charge, alpha, eps, sigma))
    return rows


def _rot(rng: random.Random):
    """random rotation matrix from a uniformly drawn unit quaternion."""
    while True:
        q = [rng.gauss(0, 1) for _ in range(4)]
        nrm = math.sqrt(sum(c * c for c in q))
        if nrm > 1e-6:
            break
    w, x, y, z = (c / nrm for c in q)
    return [
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ]


def molecular_box(n_mol: int, L: float, seed: int, extra_neutral: bool = True) -> List[AtomRow]:
    """n_mol rigid bent 3-site molecules on a jittered lattice + (optionally) one neutral polarizable atom.

    Site parameters: O  q=-0.8 e, alpha 1.45, eps 78, sigma 3.15 ; H q=+0.4 e, eps=sigma=0,
    the first H carries alpha 0.30, the second none.  O-H 0.9572 A, HOH 104.52 deg.
    """
    n = int(math.ceil((n_mol + (1 if extra_neutral else 0)) ** (1.0 / 3.0) - 1e-9))
    a = L / n
    rng = random.Random(seed)
    rows: List[AtomRow] = []
    half = math.radians(104.52) / 2
    local = [
        (0.0, 0.0, 0.0),
        (0.9572 * math.sin(half), 0.9572 * math.cos(half), 0.0),
        (-0.9572 * math.sin(half), 0.9572 * math.cos(half), 0.0),
    ]
    sites = [("O", -0.8, 1.45, 78.0, 3.15, 15.9994), ("H", 0.4, 0.30, 0.0, 0.0, 1.00794), ("H", 0.4, 0.0, 0.0, 0.0, 1.00794)]
    aid = 0
    mid = 0
    cells = [(ix, iy, iz) for ix in range(n) for iy in range(n) for iz in range(n)]
    for (ix, iy, iz) in cells[:n_mol]:
        mid += 1
        cx = (ix + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
        cy = (iy + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
        cz = (iz + 0.5) * a - L / 2 + rng.uniform(-0.1 * a, 0.1 * a)
        R = _rot(rng)
        for (name, q, al, ep, sg, mass), (lx, ly, lz) in zip(sites, local):
            aid += 1
            x = cx + R[0][0] * lx + R[0][1] * ly + R[0][2] * lz
            y = cy + R[1][0] * lx + R[1][1] * ly + R[1][2] * lz
            z = cz + R[2][0] * lx + R[2][1] * ly + R[2][2] * lz
            rows.append(AtomRow(aid, name, "H2O", "M", mid, x, y, z, mass, q, al, ep, sg))
    if extra_neutral:
        ix, iy, iz = cells[n_mol]
        mid += 1
        aid += 1
        rows.append(
            AtomRow(aid, "Xe", "Xe", "M", mid, (ix + 0.5) * a - L / 2, (iy + 0.5) * a - L / 2, (iz + 0.5) * a - L / 2,
                    131.293, 0.0, 4.044, 221.0, 4.1)
        )
    return rows


def write_pqr(path: str, rows: Sequence[AtomRow]) -> None:
    with open(path, "w") as f:
        for r in rows:
            f.write(r.line() + "\n")
        f.write("END\n")


DEFAULT_OPTS: Dict[str, object] = {
    "job_name": "t",
    "ensemble": "nvt",
    "temperature": 100.0,
    "numsteps": 1,
    "corrtime": 1,
    "seed": 1,
    "move_factor": 0.01,
    "rot_factor": 0.01,
}

QUIET = {
    "pop_histogram": "off",
    "traj_output": "off",
    "pqr_restart": "off",
    "pqr_output": "off",
    "energy_output": "off",
    "dipole_output": "off",
    "field_output": "off",
}


def write_input(path: str, pqr_name: str, basis: Sequence[Sequence[float]], opts: Dict[str, object]) -> None:
    """opts: hot-path keywords (rd_only, polarization, polar_*, ewald_*) with reference spelling."""
    lines: List[str] = []
    merged: Dict[str, object] = dict(DEFAULT_OPTS)
    merged.update(opts)
    for k, v in merged.items():
        lines.append(f"{k} {_fmt(v)}")
    for i, b in enumerate(basis):
        lines.append(f"basis{i + 1} {b[0]!r} {b[1]!r} {b[2]!r}")
    lines.append(f"pqr_input {pqr_name}")
    for k, v in QUIET.items():
        lines.append(f"{k} {v}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _fmt(v: object) -> str:
    if isinstance(v, bool):
        return "on" if v else "off"
    if isinstance(v, float):
        return repr(v)
    return str(v)


def cubic(L: float):
    return [[L, 0.0, 0.0], [0.0, L, 0.0], [0.0, 0.0, L]]


POLAR_OPTS = {
    "polarization": "on",
    "polar_damp_type": "exponential",
    "polar_damp": 2.1304,
    "polar_iterative": "on",
    "polar_max_iter": 10,
    "polar_ewald": "on",
    "ewald_kmax": 7,
}

# named fixtures: (rows builder, basis, options).  Sizes/seeds follow SURVEY.md Appendix A / §8d.
def fixture(name: str):
    if name == "ar2":  # Ar2 at exactly 4 A in a 10^4 A box (pi001 geometry), LJ only
        rows = [
            AtomRow(1, "Ar", "Ar", "M", 1, 0.0, 0.0, -2.0, 39.948, 0.0, 0.0, 119.8, 3.405),
            AtomRow(2, "Ar", "Ar", "M", 2, 0.0, 0.0, 2.0, 39.948, 0.0, 0.0, 119.8, 3.405),
        ]
        return rows, cubic(10000.0), {"rd_only": "on"}
    if name == "lj64":
        return lattice_box(64, 16.0, 3, charged=False, alpha=0.0), cubic(16.0), {"rd_only": "on"}
    if name == "ion64_es":  # LJ + Ewald, no polarization
        return lattice_box(64, 16.0, 3), cubic(16.0), {"ewald_kmax": 7}
    if name == "ion216_polar":  # SURVEY §4 anchor: rd -107838.41890820718 ...
        return lattice_box(216, 24.0, 7), cubic(24.0), dict(POLAR_OPTS)
    if name == "ion216_polar_nopbc":  # static field without Ewald (thole_field_nopbc)
        o = dict(POLAR_OPTS)
        o["polar_ewald"] = "off"
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "ion216_triclinic":
        basis = [[24.0, 0.0, 0.0], [3.0, 23.0, 0.0], [-2.0, 4.0, 22.0]]
        return lattice_box(216, 24.0, 7), basis, dict(POLAR_OPTS)
    if name == "ion1000_triclinic":  # 16 tiles in a skewed cell: tile classes, uniform images and panels for a non-orthorhombic basis
        basis = [[40.0, 0.0, 0.0], [5.0, 38.0, 0.0], [-4.0, 6.0, 37.0]]
        return lattice_box_cell(1000, basis, 21), basis, dict(POLAR_OPTS)
    if name == "ion8000_triclinic":  # 125 tiles: far-field, beyond-cutoff and common-image tile pairs in a skewed cell (large fixture: regenerated, not stored)
        basis = [[79.8, 0.0, 0.0], [9.0, 77.0, 0.0], [-6.0, 11.0, 75.0]]
        return lattice_box_cell(8000, basis, 22), basis, dict(POLAR_OPTS)
    if name == "ion216_frozen":  # every 5th atom frozen (quirk 4: frozen handling differs per term)
        return lattice_box(216, 24.0, 7, frozen_every=5), cubic(24.0), dict(POLAR_OPTS)
    if name == "ion216_framework":  # the usual MPMC layout: ONE frozen molecule (150 sites spanning three 64-atom tiles) + 66 mobile atoms
        rows = lattice_box(216, 24.0, 7)
        for r in rows[:150]:
            r.flag, r.mol_id, r.moltype = "F", 1, "MOF"
        for k, r in enumerate(rows[150:]):
            r.mol_id = 2 + k
        return rows, cubic(24.0), dict(POLAR_OPTS)
    if name == "ion216_precision":  # precision-terminated solve
        o = dict(POLAR_OPTS)
        del o["polar_max_iter"]
        o["polar_precision"] = 1e-7
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "ion216_gamma":  # polar_gamma pre-scaling, 3 iterations
        o = dict(POLAR_OPTS)
        o["polar_max_iter"] = 3
        o["polar_gamma"] = 1.03
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "ion216_alpha":  # user-set ewald_alpha / polar_ewald_alpha (differ from 3.5/rc)
        o = dict(POLAR_OPTS)
        o["ewald_alpha"] = 0.31
        o["polar_ewald_alpha"] = 0.27
        o["ewald_kmax"] = 5
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "water64_polar":  # SURVEY §4 molecular fixture: exclusions, intramolecular erf, quirk 2
        return molecular_box(64, 14.0, 5), cubic(14.0), dict(POLAR_OPTS)
    if name == "ion216_wolf":  # Wolf electrostatics instead of Ewald (coulombic_wolf)
        return lattice_box(216, 24.0, 7), cubic(24.0), {"wolf": "on"}
    if name == "water64_fh2":  # Feynman-Hibbs second-order corrections to LJ and real-space Coulomb
        return molecular_box(64, 14.0, 5), cubic(14.0), {"feynman_hibbs": "on", "feynman_hibbs_order": 2, "temperature": 77.0}
    if name == "water64_fh4":
        return molecular_box(64, 14.0, 5), cubic(14.0), {"feynman_hibbs": "on", "feynman_hibbs_order": 4, "temperature": 40.0}
    if name == "ion216_fh4_polar":
        o = dict(POLAR_OPTS)
        o.update({"feynman_hibbs": "on", "feynman_hibbs_order": 4, "temperature": 30.0})
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "ion216_gs":  # Gauss-Seidel sweeps (polar_gs): in-place dipole updates in atom order
        o = dict(POLAR_OPTS)
        o["polar_gs"] = "on"
        o["polar_max_iter"] = 6
        return lattice_box(216, 24.0, 7), cubic(24.0), o
    if name == "water64_gs_precision":  # Gauss-Seidel, precision-terminated, molecular box (non-polarizable sites, exclusions)
        o = dict(POLAR_OPTS)
        del o["polar_max_iter"]
        o["polar_gs"] = "on"
        o["polar_precision"] = 1e-8
        o["polar_rrms"] = "on"
        return molecular_box(64, 14.0, 5), cubic(14.0), o
    if name == "ion1000_gs":  # 16 tiles: the blocked sweep crosses many tile boundaries
        o = dict(POLAR_OPTS)
        o["polar_gs"] = "on"
        o["polar_max_iter"] = 4
        return lattice_box(1000, 40.0, 11), cubic(40.0), o
    if name == "lj1000":  # BASELINE config 2
        return lattice_box(1000, 40.0, 11, charged=False, alpha=0.0), cubic(40.0), {"rd_only": "on"}
    if name == "ion1000_polar":
        return lattice_box(1000, 40.0, 11), cubic(40.0), dict(POLAR_OPTS)
    if name == "ion10k_es":  # BASELINE config 3
        return lattice_box(10000, 86.0, 13), cubic(86.0), {"ewald_kmax": 7}
    if name == "ion10k_polar":  # BASELINE config 4
        return lattice_box(10000, 86.0, 13), cubic(86.0), dict(POLAR_OPTS)
    if name.startswith("ion10k_polar_bead"):  # BASELINE config 5: image `b` of the 32-bead ensemble of the config-4 box
        rows = lattice_box(10000, 86.0, 13)
        pos = bead_positions([(r.x, r.y, r.z) for r in rows], int(name[len("ion10k_polar_bead"):]))
        for r, (x, y, z) in zip(rows, pos):
            r.x, r.y, r.z = float(x), float(y), float(z)
        return rows, cubic(86.0), dict(POLAR_OPTS)
    raise KeyError(name)


def bead_positions(base_pos, bead: int, sigma: float = 0.05):
    """Image `bead` of a path-integral ensemble: base positions + Gaussian displacement (sigma in A), numpy default_rng([17, bead])
    (SURVEY §8d config 5), QUANTISED to the 6 decimals of a PQR coordinate column: what bench.py evaluates on the GPU is then, double
    for double, what the reference reads from `ion10k_polar_beadB.pqr` (tests/golden/ion10k_polar_bead{0,1}.json)."""
    import numpy as np

    # the PQR text the base positions came from has 6 decimals too: go through the same text -> double conversion
    base = np.array(np.char.mod("%.6f", np.asarray(base_pos, dtype=np.float64)), dtype=np.float64)
    rng = np.random.default_rng([17, bead])
    moved = base + rng.normal(scale=sigma, size=base.shape)
    return np.array(np.char.mod("%.6f", moved), dtype=np.float64)


SMALL_FIXTURES = [
    "ar2", "lj64", "ion64_es", "ion216_polar", "ion216_polar_nopbc", "ion216_triclinic", "ion216_frozen",
    "ion216_precision", "ion216_gamma", "ion216_alpha", "water64_polar", "lj1000", "ion1000_polar",
    "ion216_wolf", "water64_fh2", "water64_fh4", "ion216_fh4_polar", "ion216_gs", "water64_gs_precision", "ion1000_gs", "ion216_framework",
    "ion1000_triclinic",
]
LARGE_FIXTURES = ["ion10k_es", "ion10k_polar", "ion10k_polar_bead0", "ion10k_polar_bead1", "ion8000_triclinic"]


def materialize(name: str, outdir: str):
    """write NAME.pqr / NAME.in into outdir; returns (in_path, pqr_path)."""
    rows, basis, opts = fixture(name)
    os.makedirs(outdir, exist_ok=True)
    pqr = os.path.join(outdir, f"{name}.pqr")
    inp = os.path.join(outdir, f"{name}.in")
    write_pqr(pqr, rows)
    write_input(inp, f"{name}.pqr", basis, opts)
    return inp, pqr


if __name__ == "__main__":
    import sys

    out = sys.argv[1] if len(sys.argv) > 1 else "."
    for nm in (sys.argv[2:] or SMALL_FIXTURES):
        print(materialize(nm, out))
