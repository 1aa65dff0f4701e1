"""Readers for the reference's two on-disk formats, restricted to what the energy hot path consumes.

* ``read_pqr``   -- PQR geometry, token grammar of reference src/System.cpp:583-700
  (``ATOM id type moltype FLAG molid x y z mass charge alpha eps sigma omega gwp_alpha [c6 c8 c10 c9]``;
  rows with molecule type ``BOX`` ignored :592; stops at ``END`` :590; charge scaled by
  E2REDUCED = 408.7816 :624; a new molecule starts when ``molid`` changes :672).
* ``read_input`` -- ``keyword value...`` input file (src/SimulationControl.cpp:204-267), hot-path keywords
  only; any other reference keyword that would change the energy raises.

They produce flat numpy arrays (struct-of-arrays): this is the flattening a drop-in adapter performs on
the reference's Molecule->Atom linked lists.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np

E2REDUCED = 408.7816  # reference src/constants.h:35

# keywords that select physics outside SURVEY §8 -- the replacement must refuse them (§8a note 7)
UNSUPPORTED_ON = [
    "rd_crystal", "spectre", "gwp", "sg", "polarvdw", "cdvdw", "polar_ewald_full",
    "polar_wolf", "polar_wolf_full", "polar_palmo", "polar_gs_ranked", "polar_sor", "polar_esor", "polar_zodid",
    "waldmanhagler", "halgren_mixing", "c6_mixing", "dreiding", "lj_buffered_14_7", "disp_expansion",
    "axilrod_teller", "rd_anharmonic", "cavity_autoreject", "cavity_autoreject_absolute", "cuda", "opencl",
]


def read_pqr(path: str) -> Dict[str, np.ndarray]:
    pos: List[Tuple[float, float, float]] = []
    q: List[float] = []
    alpha: List[float] = []
    eps: List[float] = []
    sig: List[float] = []
    mass: List[float] = []
    mol: List[int] = []
    frozen: List[int] = []
    disp: List[int] = []
    cur_mol_token = None
    mol_index = -1
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0][:3].upper() == "END":
                break
            if t[0].upper() != "ATOM" or t[3].upper() == "BOX":
                continue
            if len(t) < 16:
                raise ValueError(f"{path}: short ATOM row: {line!r}")
            molid = int(t[5])
            if molid != cur_mol_token:
                cur_mol_token = molid
                mol_index += 1
            pos.append((float(t[6]), float(t[7]), float(t[8])))
            mass.append(float(t[9]))
            q.append(float(t[10]) * E2REDUCED)
            alpha.append(float(t[11]))
            eps.append(float(t[12]))
            sig.append(float(t[13]))
            mol.append(mol_index)
            flag = t[4].upper()
            if flag in ("A", "S", "T"):
                raise NotImplementedError(f"{path}: atom flag {flag!r} (adiabatic/spectre/target) is outside the energy hot path")
            frozen.append(1 if flag == "F" else 0)
            c = [float(x) for x in t[16:19]]
            disp.append(1 if any(v != 0.0 for v in c) else 0)
    return {
        "pos": np.ascontiguousarray(np.array(pos, dtype=np.float64).reshape(-1, 3)),
        "charge": np.array(q, dtype=np.float64),
        "polarizability": np.array(alpha, dtype=np.float64),
        "epsilon": np.array(eps, dtype=np.float64),
        "sigma": np.array(sig, dtype=np.float64),
        "mass": np.array(mass, dtype=np.float64),
        "mol_id": np.array(mol, dtype=np.int32),
        "frozen": np.array(frozen, dtype=np.int32),
        "has_disp": np.array(disp, dtype=np.int32),
    }


def _onoff(v: str) -> int:
    v = v.lower()
    if v == "on":
        return 1
    if v == "off":
        return 0
    raise ValueError(f"expected on/off, got {v!r}")


def read_input(path: str) -> Dict[str, object]:
    """returns {'basis': (3,3) array, 'pqr_input': str, 'options': {...}} with reference defaults
    (src/System.h:21-24,510-831: ewald_kmax 7, rd_lrc on, polar_gamma 1.0, polar_max_iter 10)."""
    opts: Dict[str, object] = {
        "rd_only": 0, "rd_lrc": 1, "polarization": 0, "polar_iterative": 0, "polar_ewald": 0, "polar_max_iter": 10,
        "polar_gs": 0, "polar_rrms": 0, "ewald_kmax": 7, "polar_precision": 0.0, "polar_gamma": 1.0, "polar_damp": 0.0,
        "damp_type": None, "ewald_alpha": None, "polar_ewald_alpha": None,
        "wolf": 0, "feynman_hibbs": 0, "feynman_hibbs_order": 0, "temperature": 0.0,
    }
    basis = np.zeros((3, 3), dtype=np.float64)
    pqr = None
    ensemble = None
    with open(path) as f:
        for raw in f:
            line = raw.split("!")[0].split("#")[0].strip()
            if not line:
                continue
            t = line.split()
            k = t[0].lower()
            v = t[1:]
            if k in ("basis1", "basis2", "basis3"):
                basis[int(k[-1]) - 1] = [float(x) for x in v[:3]]
            elif k == "pqr_input":
                pqr = v[0]
            elif k == "ensemble":
                ensemble = v[0].lower()
            elif k in ("rd_only", "rd_lrc", "polarization", "polar_iterative", "polar_ewald", "polar_gs", "polar_rrms", "wolf", "feynman_hibbs"):
                opts[k] = _onoff(v[0])
            elif k in ("polar_max_iter", "ewald_kmax", "feynman_hibbs_order"):
                opts[k] = int(v[0])
            elif k in ("polar_precision", "polar_gamma", "polar_damp", "ewald_alpha", "polar_ewald_alpha", "temperature"):
                opts[k] = float(v[0])
            elif k == "polar_damp_type":
                opts["damp_type"] = v[0].lower()
            elif k in UNSUPPORTED_ON and (not v or v[0].lower() != "off"):
                raise NotImplementedError(f"{path}: keyword {k!r} selects physics outside the energy hot path (SURVEY §8a note 7)")
            # everything else (job_name, temperature, numsteps, output switches, ...) does not enter energy()
    if pqr is None:
        raise ValueError(f"{path}: no pqr_input")
    return {"basis": basis, "pqr_input": os.path.join(os.path.dirname(os.path.abspath(path)), pqr), "options": opts,
            "ensemble": ensemble}


def load_case(in_path: str):
    """convenience: (atoms dict, basis, options) for an input file and the PQR it names."""
    cfg = read_input(in_path)
    atoms = read_pqr(cfg["pqr_input"])
    return atoms, cfg["basis"], cfg["options"]
