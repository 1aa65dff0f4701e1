"""ctypes binding of the CPU oracle (oracle/libmpmc_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (mpmcxx_amd) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpmc_oracle.so")


class OrcSystem(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("pos", C.POINTER(C.c_double)),
        ("charge", C.POINTER(C.c_double)),
        ("polarizability", C.POINTER(C.c_double)),
        ("epsilon", C.POINTER(C.c_double)),
        ("sigma", C.POINTER(C.c_double)),
        ("mol_id", C.POINTER(C.c_int)),
        ("frozen", C.POINTER(C.c_int)),
        ("has_disp", C.POINTER(C.c_int)),
        ("basis", C.c_double * 9),
        ("recip", C.c_double * 9),
        ("volume", C.c_double),
        ("cutoff", C.c_double),
        ("rd_only", C.c_int),
        ("rd_lrc", C.c_int),
        ("polarization", C.c_int),
        ("polar_iterative", C.c_int),
        ("polar_ewald", C.c_int),
        ("polar_max_iter", C.c_int),
        ("polar_gs", C.c_int),
        ("polar_rrms", C.c_int),
        ("ewald_kmax", C.c_int),
        ("polar_precision", C.c_double),
        ("polar_gamma", C.c_double),
        ("polar_damp", C.c_double),
        ("ewald_alpha", C.c_double),
        ("polar_ewald_alpha", C.c_double),
        ("wolf", C.c_int),
        ("feynman_hibbs", C.c_int),
        ("feynman_hibbs_order", C.c_int),
        ("temperature", C.c_double),
        ("mass", C.POINTER(C.c_double)),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("energy", C.c_double), ("rd_energy", C.c_double), ("coulombic_energy", C.c_double),
        ("polarization_energy", C.c_double), ("vdw_energy", C.c_double),
        ("es_real", C.c_double), ("es_recip", C.c_double), ("es_self", C.c_double),
        ("lj_pairs", C.c_double), ("lrc_pair", C.c_double), ("lrc_self", C.c_double),
        ("dipole_rrms", C.c_double),
        ("n_pairs", C.c_longlong), ("n_intra", C.c_longlong), ("n_rd_excluded", C.c_longlong),
        ("n_es_excluded", C.c_longlong), ("n_frozen", C.c_longlong),
        ("n_lj_in_cutoff", C.c_longlong), ("n_es_in_cutoff", C.c_longlong),
        ("polar_iterations", C.c_int), ("iterator_failed", C.c_int),
    ]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "mpmc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_pbc_update.argtypes = [dp, dp, dp, dp]
        L.orc_pbc_update.restype = None
        L.orc_energy.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcResult), dp, dp, dp]
        L.orc_energy.restype = C.c_int
        L.orc_lj.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcResult)]
        L.orc_lj.restype = C.c_double
        L.orc_lj_exact.argtypes = [C.POINTER(OrcSystem), dp]
        L.orc_lj_exact.restype = None
        L.orc_coulombic_real.argtypes = [C.POINTER(OrcSystem), C.POINTER(OrcResult)]
        L.orc_coulombic_real.restype = C.c_double
        L.orc_coulombic_reciprocal.argtypes = [C.POINTER(OrcSystem)]
        L.orc_coulombic_reciprocal.restype = C.c_double
        L.orc_coulombic_self.argtypes = [C.POINTER(OrcSystem)]
        L.orc_coulombic_self.restype = C.c_double
        L.orc_thole_field.argtypes = [C.POINTER(OrcSystem), dp]
        L.orc_thole_field.restype = None
        L.orc_thole_amatrix_block.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_int, dp]
        L.orc_thole_amatrix_block.restype = None
        L.orc_minimum_image.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_int, dp, dp]
        L.orc_minimum_image.restype = C.c_double
        L.orc_time_sample.argtypes = [C.POINTER(OrcSystem), C.c_int, dp]
        L.orc_time_sample.restype = C.c_double
        L.orc_pi_aggregate.argtypes = [C.c_int, dp, dp, dp, dp, dp]
        L.orc_pi_aggregate.restype = C.c_double
        L.orc_pi_kinetic.argtypes = [C.c_int, C.c_int, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double, dp]
        L.orc_pi_kinetic.restype = C.c_double
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def pbc_update(basis: np.ndarray):
    b = np.ascontiguousarray(basis, dtype=np.float64).reshape(9)
    R = np.zeros(9)
    vol = C.c_double()
    cut = C.c_double()
    lib().orc_pbc_update(_dp(b), _dp(R), C.byref(vol), C.byref(cut))
    return R.reshape(3, 3), vol.value, cut.value


class OracleSystem:
    """owns numpy copies of the atom arrays and an orc_system pointing at them."""

    def __init__(self, atoms: Dict[str, np.ndarray], basis: np.ndarray, options: Dict[str, object]):
        self.n = int(atoms["pos"].shape[0])
        self._keep = {
            "pos": np.ascontiguousarray(atoms["pos"], dtype=np.float64).reshape(-1),
            "charge": np.ascontiguousarray(atoms["charge"], dtype=np.float64),
            "polarizability": np.ascontiguousarray(atoms["polarizability"], dtype=np.float64),
            "epsilon": np.ascontiguousarray(atoms["epsilon"], dtype=np.float64),
            "sigma": np.ascontiguousarray(atoms["sigma"], dtype=np.float64),
            "mol_id": np.ascontiguousarray(atoms["mol_id"], dtype=np.int32),
            "frozen": np.ascontiguousarray(atoms["frozen"], dtype=np.int32),
            "has_disp": np.ascontiguousarray(atoms.get("has_disp", np.zeros(self.n, dtype=np.int32)), dtype=np.int32),
            "mass": np.ascontiguousarray(atoms.get("mass", np.ones(self.n)), dtype=np.float64),
        }
        s = OrcSystem()
        s.n = self.n
        k = self._keep
        s.pos, s.charge, s.polarizability = _dp(k["pos"]), _dp(k["charge"]), _dp(k["polarizability"])
        s.epsilon, s.sigma = _dp(k["epsilon"]), _dp(k["sigma"])
        s.mol_id, s.frozen, s.has_disp = _ip(k["mol_id"]), _ip(k["frozen"]), _ip(k["has_disp"])
        R, vol, cut = pbc_update(basis)
        b = np.ascontiguousarray(basis, dtype=np.float64).reshape(9)
        for i in range(9):
            s.basis[i] = b[i]
            s.recip[i] = R.reshape(9)[i]
        s.volume, s.cutoff = vol, cut
        o = options
        s.rd_only, s.rd_lrc = int(o.get("rd_only", 0)), int(o.get("rd_lrc", 1))
        s.polarization, s.polar_iterative = int(o.get("polarization", 0)), int(o.get("polar_iterative", 0))
        s.polar_ewald, s.polar_max_iter = int(o.get("polar_ewald", 0)), int(o.get("polar_max_iter", 10))
        s.polar_gs, s.polar_rrms = int(o.get("polar_gs", 0)), int(o.get("polar_rrms", 0))
        s.ewald_kmax = int(o.get("ewald_kmax", 7))
        s.polar_precision = float(o.get("polar_precision", 0.0))
        s.polar_gamma = float(o.get("polar_gamma", 1.0))
        s.polar_damp = float(o.get("polar_damp", 0.0))
        ea, pea = o.get("ewald_alpha"), o.get("polar_ewald_alpha")
        # update_pbc defaults (reference System.cpp:871-874)
        s.ewald_alpha = float(ea) if ea is not None else 3.5 / cut
        s.polar_ewald_alpha = float(pea) if pea is not None else 3.5 / cut
        s.wolf = int(o.get("wolf", 0))
        s.feynman_hibbs = int(o.get("feynman_hibbs", 0))
        fo = int(o.get("feynman_hibbs_order", 0) or 0)
        s.feynman_hibbs_order = fo if fo in (2, 4) else 2  # check_feynman_hibbs_options defaults to 2 (SimulationControl.cpp:2497-2500)
        s.temperature = float(o.get("temperature", 0.0) or 0.0)
        s.mass = _dp(k["mass"])
        self.s = s
        self.recip, self.volume, self.cutoff = R, vol, cut

    def energy(self, want_atoms: bool = True):
        res = OrcResult()
        E = np.zeros((self.n, 3))
        mu = np.zeros((self.n, 3))
        F = np.zeros((self.n, 3))
        rc = lib().orc_energy(C.byref(self.s), C.byref(res), _dp(E), _dp(mu), _dp(F))
        if rc != 0:
            raise RuntimeError(f"orc_energy rc={rc}")
        out = {f: getattr(res, f) for f, _ in OrcResult._fields_}
        if want_atoms:
            out.update(ef_static=E, mu=mu, ef_induced=F)
        return out

    def lj_exact(self):
        """lj() with exactly rounded sums next to the reference's list-order sums (orc_lj_exact): the measuring stick that separates the
        HIP path's error from the reference's own accumulation drift (System.Energy.cpp:1011)."""
        o = np.zeros(7)
        lib().orc_lj_exact(C.byref(self.s), _dp(o))
        return {"lj_pairs_exact": o[0], "lrc_pair_exact": o[1], "lrc_self_exact": o[2], "rd_exact": o[3],
                "rd_list_order": o[4], "lj_pairs_list_order": o[5], "lrc_pair_list_order": o[6]}

    def time_sample(self, stride: int):
        """(estimated seconds per stage of one full evaluation [7], wall seconds spent)"""
        out = np.zeros(7)
        wall = lib().orc_time_sample(C.byref(self.s), int(stride), _dp(out))
        return out, wall

    def thole_field(self):
        E = np.zeros((self.n, 3))
        lib().orc_thole_field(C.byref(self.s), _dp(E))
        return E

    def amatrix_block(self, i: int, j: int):
        blk = np.zeros(9)
        lib().orc_thole_amatrix_block(C.byref(self.s), i, j, _dp(blk))
        return blk

    def minimum_image(self, i: int, j: int):
        d = np.zeros(3)
        r = C.c_double()
        rimg = lib().orc_minimum_image(C.byref(self.s), i, j, _dp(d), C.byref(r))
        return rimg, d, r.value


def pi_aggregate(rd, es, pol, vdw=None):
    rd = np.ascontiguousarray(rd, dtype=np.float64)
    es = np.ascontiguousarray(es, dtype=np.float64)
    pol = np.ascontiguousarray(pol, dtype=np.float64)
    vdw = np.zeros_like(rd) if vdw is None else np.ascontiguousarray(vdw, dtype=np.float64)
    out = np.zeros(4)
    v = lib().orc_pi_aggregate(len(rd), _dp(rd), _dp(es), _dp(pol), _dp(vdw), _dp(out))
    return v, out


def pi_kinetic(pos_beads, mass, mol_id, frozen, temperature):
    """PI_calculate_kinetic restated (oracle/mpmc_oracle.c: orc_pi_kinetic).  pos_beads: (P, n, 3).  Returns (K [Kelvin], chain_mass_len2)."""
    pos = np.ascontiguousarray(pos_beads, dtype=np.float64)
    P, n, _ = pos.shape
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    mol = np.ascontiguousarray(mol_id, dtype=np.int32)
    fr = np.ascontiguousarray(frozen, dtype=np.int32)
    chain = C.c_double()
    ip = C.POINTER(C.c_int)
    k = lib().orc_pi_kinetic(P, n, _dp(pos), _dp(mass), mol.ctypes.data_as(ip), fr.ctypes.data_as(ip), float(temperature), C.byref(chain))
    return k, chain.value
