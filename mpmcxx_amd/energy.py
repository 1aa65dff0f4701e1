"""ctypes binding of the C ABI (include/mpmc_energy.h) -- host-side mirror of the reference's energy surface.

`System` mirrors the part of the reference's `System` class that the energy hot path exposes
(reference src/System.h:314-402): ``energy()``, ``lj()``, ``coulombic()``, ``coulombic_real()``,
``coulombic_reciprocal()``, ``coulombic_self()``, ``polar()``, ``thole_field()``, ``thole_amatrix()`` and the
``observables`` it fills; errors surface as ``MpmcError(code)`` with the reference's integer error codes
(src/constants.h:108-147), the Python spelling of the reference's ``throw <int>``.

There is no CPU fallback: if the HIP library is missing or no device is present this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))

# status codes (include/mpmc_energy.h)
MPMC_OK = 0
ERR_UNSUPPORTED = 4004
ERR_INVALID_SETTING = 4000
ERR_NO_DEVICE = -1
DAMPING = {"off": 0, "linear": 1, "exponential": 2, None: 2}
SOLVER = {"auto": 0, "matrix_free": 1, "compact": 2, "dense": 3}
K_NAMES = ["pair", "recip", "field", "tensor", "dipole_iter", "reduce", "dipole_far", "classes"]


class MpmcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mpmc error {code}: {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [
        ("rd_only", C.c_int32), ("rd_lrc", C.c_int32), ("polarization", C.c_int32), ("polar_iterative", C.c_int32),
        ("polar_ewald", C.c_int32), ("polar_max_iter", C.c_int32), ("polar_gs", C.c_int32), ("polar_rrms", C.c_int32),
        ("damp_type", C.c_int32), ("ewald_kmax", C.c_int32), ("solver", C.c_int32), ("wolf", C.c_int32),
        ("polar_precision", C.c_double), ("polar_gamma", C.c_double), ("polar_damp", C.c_double),
        ("ewald_alpha", C.c_double), ("polar_ewald_alpha", C.c_double), ("unsupported_flags", C.c_uint64),
        ("feynman_hibbs", C.c_int32), ("feynman_hibbs_order", C.c_int32), ("temperature", C.c_double),
    ]


class Result(C.Structure):
    _fields_ = [
        ("energy", C.c_double), ("rd_energy", C.c_double), ("coulombic_energy", C.c_double), ("polarization_energy", C.c_double),
        ("vdw_energy", C.c_double), ("three_body_energy", C.c_double), ("kinetic_energy", C.c_double),
        ("es_real", C.c_double), ("es_recip", C.c_double), ("es_self", C.c_double),
        ("lj_pairs", C.c_double), ("lrc_pair", C.c_double), ("lrc_self", C.c_double),
        ("dipole_rrms", C.c_double), ("N", C.c_double), ("NU", C.c_double),
        ("n_pairs", C.c_int64), ("n_lj_in_cutoff", C.c_int64), ("n_es_in_cutoff", C.c_int64), ("n_intra", C.c_int64),
        ("n_rd_excluded", C.c_int64), ("n_es_excluded", C.c_int64), ("n_frozen", C.c_int64),
        ("polar_iterations", C.c_int32), ("iterator_failed", C.c_int32),
    ]

    def as_dict(self) -> Dict[str, float]:
        return {f: getattr(self, f) for f, _ in self._fields_}


class GibbsMove(C.Structure):
    _fields_ = [("movetype", C.c_int32 * 2), ("temperature", C.c_double), ("init_energy", C.c_double * 2), ("final_energy", C.c_double * 2),
                ("N", C.c_double * 2), ("volume", C.c_double * 2), ("checkpoint_volume_0", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_int64 * 8)]


_lib = None


def lib():
    """load (building if needed) mpmcxx_amd/libmpmc_energy.so; raises if it cannot be produced."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MPMC_ENERGY_LIB") or _build.LIB  # override: same-box A/B of two builds (tools/ab_bench.sh)
    if not os.path.exists(path):
        if path != _build.LIB:
            raise FileNotFoundError(path)
        path = _build.build_library()
    L = C.CDLL(path)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    vp = C.c_void_p
    L.mpmc_abi_version.restype = C.c_int
    L.mpmc_device_count.argtypes = [C.POINTER(C.c_int)]
    if hasattr(L, "mpmc_device_synchronize") or not os.environ.get("MPMC_ENERGY_LIB"):  # (ABI 6)
        L.mpmc_device_synchronize.argtypes = [C.c_int]
        L.mpmc_device_name.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.mpmc_last_error.argtypes = [vp]
    L.mpmc_last_error.restype = C.c_char_p
    L.mpmc_pbc_compute.argtypes = [dp, dp, dp, dp]
    L.mpmc_default_options.argtypes = [C.POINTER(Options)]
    L.mpmc_default_options.restype = None
    L.mpmc_ctx_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.mpmc_ctx_destroy.argtypes = [vp]
    L.mpmc_set_box.argtypes = [vp, dp, dp, C.c_double, C.c_double]
    L.mpmc_set_options.argtypes = [vp, C.POINTER(Options)]
    L.mpmc_set_atoms.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, ip, ip, ip, dp]
    L.mpmc_update_positions.argtypes = [vp, C.c_int, C.c_int, dp]
    L.mpmc_set_positions_device.argtypes = [vp, vp]
    L.mpmc_energy.argtypes = [vp, C.POINTER(Result)]
    L.mpmc_energy_async.argtypes = [vp]
    L.mpmc_energy_wait.argtypes = [vp, C.POINTER(Result)]
    for name in ("mpmc_lj", "mpmc_coulombic", "mpmc_coulombic_real", "mpmc_coulombic_reciprocal", "mpmc_coulombic_self", "mpmc_polar"):
        getattr(L, name).argtypes = [vp, dp]
    L.mpmc_thole_field.argtypes = [vp, dp]
    L.mpmc_thole_amatrix.argtypes = [vp, C.c_int, C.c_int, dp]
    L.mpmc_get_dipoles.argtypes = [vp, dp, dp, dp]
    L.mpmc_update_com.argtypes = [vp, dp, dp, dp, C.POINTER(C.c_int)]
    L.mpmc_pi_potential_local.argtypes = [C.POINTER(vp), C.c_int, dp, C.POINTER(Result), C.POINTER(C.c_int)]
    L.mpmc_pi_potential_local_host.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(dp), dp, C.POINTER(Result), C.POINTER(C.c_int)]
    L.mpmc_pi_finish.argtypes = [dp, C.c_int, dp]
    L.mpmc_last_batch_size.argtypes = [vp]
    L.mpmc_pi_chain_mass_length2.argtypes = [C.c_int, C.c_int, dp, dp, ip]
    L.mpmc_pi_chain_mass_length2.restype = C.c_double
    L.mpmc_pi_kinetic.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]
    L.mpmc_pi_kinetic.restype = C.c_double
    L.mpmc_pi_finish.restype = C.c_double
    L.mpmc_set_profiling.argtypes = [vp, C.c_int]
    L.mpmc_get_timings.argtypes = [vp, C.POINTER(Timings), C.c_int]
    L.mpmc_synchronize.argtypes = [vp]
    L.mpmc_memory_usage.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.mpmc_get_tile_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.mpmc_trial_begin.argtypes = [vp, C.c_int, C.c_int, dp]
    L.mpmc_trial_energy.argtypes = [vp, C.POINTER(Result)]
    L.mpmc_trial_accept.argtypes = [vp]
    L.mpmc_trial_reject.argtypes = [vp]
    L.mpmc_gibbs_energy.argtypes = [vp, vp, C.POINTER(Result), C.POINTER(Result)]
    L.mpmc_gibbs_boltzmann_factor.argtypes = [C.POINTER(GibbsMove), dp, dp]
    L.mpmc_rccl_version.argtypes = [C.POINTER(C.c_int)]
    if hasattr(L, "mpmc_rccl_library_path") or not os.environ.get("MPMC_ENERGY_LIB"):  # (ABI 6)
        L.mpmc_rccl_library_path.argtypes = []
        L.mpmc_rccl_library_path.restype = C.c_char_p
    L.mpmc_comm_unique_id.argtypes = [C.c_char_p]
    L.mpmc_comm_init_rank.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.mpmc_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    L.mpmc_comm_destroy.argtypes = [vp]
    L.mpmc_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mpmc_comm_last_error.argtypes = [vp]
    L.mpmc_comm_last_error.restype = C.c_char_p
    L.mpmc_comm_allgather_f64.argtypes = [vp, dp, C.c_int64, dp]
    L.mpmc_pi_gather_beads.argtypes = [vp, dp, C.c_int, C.c_int, dp]
    if hasattr(L, "mpmc_hint_in_flight") or not os.environ.get("MPMC_ENERGY_LIB"):
        L.mpmc_hint_in_flight.argtypes = [vp, C.c_int]
    L.mpmc_pi_allreduce.argtypes = [C.POINTER(vp), C.c_int, dp, C.POINTER(Result), C.POINTER(C.c_int)]
    if hasattr(L, "mpmc_pi_allreduce_info") or not os.environ.get("MPMC_ENERGY_LIB"):  # (an older build under the A/B override lacks the ABI-5 entry)
        L.mpmc_pi_allreduce_info.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mpmc_debug_configure.argtypes = [vp, C.c_char_p, C.c_double]
    L.mpmc_debug_last_pair_kernel.argtypes = [vp]
    L.mpmc_debug_last_trial_was_full.argtypes = [vp]
    L.mpmc_debug_erfc_table.argtypes = [C.c_double, dp, dp]
    if hasattr(L, "mpmc_debug_erfc_table_field") or not os.environ.get("MPMC_ENERGY_LIB"):
        L.mpmc_debug_erfc_table_field.argtypes = [C.c_double, dp]
    L.mpmc_debug_pair_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.mpmc_debug_time_panel.argtypes = [vp, C.c_int, dp]
    L.mpmc_debug_time_pair.argtypes = [vp, C.c_int, dp]
    _lib = L
    return L


def configure(key: str, value: float):
    """measurement / A-B switch for contexts created AFTER this call (mpmc_debug_configure with a null context)."""
    rc = lib().mpmc_debug_configure(None, key.encode(), float(value))
    if rc != MPMC_OK:
        raise MpmcError(rc, f"mpmc_debug_configure: unknown key or bad value: {key}={value}")


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def device_count() -> int:
    n = C.c_int(0)
    lib().mpmc_device_count(C.byref(n))
    return n.value


def device_synchronize(device: int = 0):
    """everything this process enqueued on `device` has finished (mpmc_device_synchronize)."""
    rc = lib().mpmc_device_synchronize(int(device))
    if rc != MPMC_OK:
        raise MpmcError(rc, f"mpmc_device_synchronize({device})")


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    rc = lib().mpmc_device_name(int(device), buf, 256)
    if rc != MPMC_OK:
        raise MpmcError(rc, f"mpmc_device_name({device})")
    return buf.value.decode()


def pbc_compute(basis: np.ndarray):
    """PeriodicBoundary::update: returns (reciprocal(3,3), volume, cutoff)."""
    b = np.ascontiguousarray(basis, dtype=np.float64).reshape(9)
    R = np.zeros(9)
    vol, cut = C.c_double(), C.c_double()
    rc = lib().mpmc_pbc_compute(_dp(b), _dp(R), C.byref(vol), C.byref(cut))
    if rc != MPMC_OK:
        raise MpmcError(rc, "invalid box")
    return R.reshape(3, 3), vol.value, cut.value


def make_options(opts: Dict[str, object]) -> Options:
    o = Options()
    lib().mpmc_default_options(C.byref(o))
    for k in ("rd_only", "rd_lrc", "polarization", "polar_iterative", "polar_ewald", "polar_max_iter", "polar_gs", "polar_rrms", "ewald_kmax",
              "wolf", "feynman_hibbs", "feynman_hibbs_order"):
        if k in opts and opts[k] is not None:
            setattr(o, k, int(opts[k]))
    for k in ("polar_precision", "polar_gamma", "polar_damp", "temperature"):
        if k in opts and opts[k] is not None:
            setattr(o, k, float(opts[k]))
    for k in ("ewald_alpha", "polar_ewald_alpha"):
        v = opts.get(k)
        setattr(o, k, float(v) if v is not None else 0.0)
    dt = opts.get("damp_type")
    o.damp_type = DAMPING[dt] if not isinstance(dt, int) else dt
    sv = opts.get("solver", "auto")
    o.solver = SOLVER[sv] if not isinstance(sv, int) else sv
    o.unsupported_flags = int(opts.get("unsupported_flags", 0))
    return o


class System:
    """Device-resident state of one box (one reference `System` / one PI bead)."""

    def __init__(self, atoms: Dict[str, np.ndarray], basis: np.ndarray, options: Dict[str, object], device: int = 0,
                 max_atoms: Optional[int] = None):
        L = lib()
        self._L = L
        self._h = C.c_void_p()
        n = int(np.asarray(atoms["pos"]).shape[0])
        rc = L.mpmc_ctx_create(int(device), int(max_atoms or n), C.byref(self._h))
        if rc != MPMC_OK:
            raise MpmcError(rc, (L.mpmc_last_error(None) or b"").decode())
        self.n = n
        self._obs: Dict[str, float] = {}
        self._obs_lazy = None  # (ctypes Result array, index): turned into a dict when somebody looks (the PI loops fill 32 of these per step)
        self.set_box(basis)
        self.set_options(options)
        self.set_atoms(atoms)

    # -- plumbing -------------------------------------------------------------------------------------------
    @property
    def observables(self) -> Dict[str, float]:
        if self._obs_lazy is not None:
            arr, i = self._obs_lazy
            self._obs, self._obs_lazy = arr[i].as_dict(), None
        return self._obs

    @observables.setter
    def observables(self, d):
        self._obs, self._obs_lazy = d, None

    def _check(self, rc: int):
        if rc != MPMC_OK:
            raise MpmcError(rc, (self._L.mpmc_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.mpmc_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def configure(self, key: str, value: float):
        """measurement / A-B switch of THIS context (see mpmc_debug_configure in csrc/context.cpp)."""
        self._check(self._L.mpmc_debug_configure(self._h, key.encode(), float(value)))

    def last_trial_was_full(self) -> bool:
        """whether the last trial move ran a full evaluation of the trial configuration (True) or per-move delta energies (False)."""
        return self._L.mpmc_debug_last_trial_was_full(self._h) == 1

    def last_pair_kernel(self) -> str:
        return "sweep" if self._L.mpmc_debug_last_pair_kernel(self._h) == 1 else "fused"

    # -- state ------------------------------------------------------------------------------------------------
    def set_box(self, basis: np.ndarray):
        b = np.ascontiguousarray(basis, dtype=np.float64).reshape(9)
        self._check(self._L.mpmc_set_box(self._h, _dp(b), None, 0.0, 0.0))
        self.basis = b.reshape(3, 3).copy()

    def set_options(self, options: Dict[str, object]):
        self._opts = make_options(options)
        self._check(self._L.mpmc_set_options(self._h, C.byref(self._opts)))
        self.options = dict(options)

    def set_atoms(self, atoms: Dict[str, np.ndarray]):
        f = lambda k: np.ascontiguousarray(atoms[k], dtype=np.float64)
        g = lambda k: np.ascontiguousarray(atoms[k], dtype=np.int32)
        pos = f("pos").reshape(-1)
        n = pos.size // 3
        disp = g("has_disp") if "has_disp" in atoms else None
        mass = f("mass") if "mass" in atoms else None
        self._check(self._L.mpmc_set_atoms(self._h, n, _dp(pos), _dp(f("charge")), _dp(f("polarizability")), _dp(f("epsilon")),
                                           _dp(f("sigma")), _ip(g("mol_id")), _ip(g("frozen")), _ip(disp), _dp(mass)))
        self.n = n

    def update_positions(self, first: int, pos: np.ndarray):
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
        self._check(self._L.mpmc_update_positions(self._h, int(first), p.size // 3, _dp(p)))

    def set_positions_device(self, data_ptr: int):
        """positions from device memory ([n][3] fp64).  The caller synchronizes the stream that produced them first (torch: `torch.cuda.current_stream().synchronize()`);
        the library reads them on its own stream and is done with the buffer when this returns."""
        self._check(self._L.mpmc_set_positions_device(self._h, C.c_void_p(data_ptr)))

    # -- double System::energy() (reference src/System.Energy.cpp:19) ---------------------------------------------
    def energy(self) -> float:
        r = Result()
        self._check(self._L.mpmc_energy(self._h, C.byref(r)))
        self.observables = r.as_dict()
        return r.energy

    def hint_in_flight(self, n: int):
        """scheduling hint for energy_async: how many evaluations the caller keeps in flight together with this one (mpmc_hint_in_flight)."""
        self._check(self._L.mpmc_hint_in_flight(self._h, int(n)))

    def energy_async(self):
        self._check(self._L.mpmc_energy_async(self._h))

    def energy_wait(self) -> float:
        r = Result()
        self._check(self._L.mpmc_energy_wait(self._h, C.byref(r)))
        self.observables = r.as_dict()
        return r.energy

    # -- trial moves (reference: per-pair recalculate_energy cache, src/System.cpp:1211-1224) ------------------------------
    def trial_energy(self, first: int, pos: np.ndarray) -> float:
        """energy of the configuration in which atoms first.. take the positions `pos`; follow with accept() or reject()."""
        p = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1)
        self._check(self._L.mpmc_trial_begin(self._h, int(first), p.size // 3, _dp(p)))
        r = Result()
        rc = self._L.mpmc_trial_energy(self._h, C.byref(r))
        if rc != MPMC_OK:
            self._L.mpmc_trial_reject(self._h)
            self._check(rc)
        self.trial_observables = r.as_dict()
        return r.energy

    def accept(self):
        self._check(self._L.mpmc_trial_accept(self._h))
        self.observables = dict(self.trial_observables)

    def reject(self):
        self._check(self._L.mpmc_trial_reject(self._h))

    def _scalar(self, fn) -> float:
        v = C.c_double()
        self._check(fn(self._h, C.byref(v)))
        return v.value

    def lj(self) -> float:
        return self._scalar(self._L.mpmc_lj)

    def coulombic(self) -> float:
        return self._scalar(self._L.mpmc_coulombic)

    def coulombic_real(self) -> float:
        return self._scalar(self._L.mpmc_coulombic_real)

    def coulombic_reciprocal(self) -> float:
        return self._scalar(self._L.mpmc_coulombic_reciprocal)

    def coulombic_self(self) -> float:
        return self._scalar(self._L.mpmc_coulombic_self)

    def polar(self) -> float:
        return self._scalar(self._L.mpmc_polar)

    def thole_field(self) -> np.ndarray:
        E = np.zeros((self.n, 3))
        self._check(self._L.mpmc_thole_field(self._h, _dp(E)))
        return E

    def thole_amatrix(self, row0: int = 0, nrows: Optional[int] = None) -> np.ndarray:
        nrows = 3 * self.n - row0 if nrows is None else nrows
        A = np.zeros((nrows, 3 * self.n))
        self._check(self._L.mpmc_thole_amatrix(self._h, int(row0), int(nrows), _dp(A)))
        return A

    def dipoles(self):
        mu, E, F = np.zeros((self.n, 3)), np.zeros((self.n, 3)), np.zeros((self.n, 3))
        self._check(self._L.mpmc_get_dipoles(self._h, _dp(mu), _dp(E), _dp(F)))
        return mu, E, F

    def update_com(self):
        nm = C.c_int(0)
        com = np.zeros((self.n, 3))
        wcom = np.zeros((self.n, 3))
        wpos = np.zeros((self.n, 3))
        self._check(self._L.mpmc_update_com(self._h, _dp(com), _dp(wcom), _dp(wpos), C.byref(nm)))
        return com[: nm.value], wcom[: nm.value], wpos

    # -- measurement --------------------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._L.mpmc_set_profiling(self._h, 1 if on else 0))

    def timings(self, reset: bool = False) -> Dict[str, Dict[str, float]]:
        t = Timings()
        self._check(self._L.mpmc_get_timings(self._h, C.byref(t), 1 if reset else 0))
        return {K_NAMES[i]: {"ms": t.ms[i], "launches": int(t.launches[i])} for i in range(len(K_NAMES))}

    def synchronize(self):
        self._check(self._L.mpmc_synchronize(self._h))

    def tile_stats(self) -> Dict[str, int]:
        """tile-pair classes of the last evaluation (see mpmc_get_tile_stats)."""
        a = (C.c_int64 * 4)()
        self._check(self._L.mpmc_get_tile_stats(self._h, a))
        return {"tile_pairs": a[0], "thole_stored": a[1], "thole_far": a[2], "beyond_cutoff": a[3]}

    def pair_stats(self) -> Dict[str, int]:
        """exact atom-pair counts behind the tile-pair classes of the last evaluation (mpmc_debug_pair_stats)."""
        a = (C.c_int64 * 12)()
        self._check(self._L.mpmc_debug_pair_stats(self._h, a))
        keys = ["pairs", "pairs_stored", "pairs_far", "pairs_beyond_cutoff", "nonuniform_dims_x_pairs_stored", "nonuniform_dims_x_pairs_far",
                "pairs_swept", "nonuniform_dims_x_pairs_swept", "tile_pairs", "tile_pairs_stored", "tile_pairs_far", "tile_pairs_beyond_cutoff"]
        return {k: int(a[i]) for i, k in enumerate(keys)}

    def time_kernel(self, which: str, reps: int = 100) -> float:
        """ms per launch of the dominant kernels of the LAST evaluation, `reps` launches back to back between one pair of HIP events on the
        context's stream: which = "panel" (Jacobi contraction) | "pair" (pair sweep)."""
        v = C.c_double(0.0)
        fn = self._L.mpmc_debug_time_panel if which == "panel" else self._L.mpmc_debug_time_pair
        self._check(fn(self._h, int(reps), C.byref(v)))
        return v.value

    def last_batch_size(self) -> int:
        """systems that shared each launch of the dipole iterations in the last evaluation (pi_potential_local batches compatible beads)."""
        return int(self._L.mpmc_last_batch_size(self._h))

    def memory_usage(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self._L.mpmc_memory_usage(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value


class ResultList:
    """per-bead results of a PI loop: the ctypes array the library filled, turned into dicts on access (a 30-field dict per bead and step
    is 1 % of a 32-bead step when nobody reads it)."""

    def __init__(self, arr, n: int):
        self._arr, self._n = arr, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._arr[k].as_dict() for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._arr[i].as_dict()

    def __iter__(self):
        return (self._arr[k].as_dict() for k in range(self._n))

    def table4(self) -> np.ndarray:
        """(n, 4) = {rd, coulombic, polarization, vdw} per bead -- what PI_calculate_potential combines (PathIntegral.cpp:763-766)."""
        raw = np.frombuffer(self._arr, dtype=np.float64, count=self._n * (C.sizeof(Result) // 8)).reshape(self._n, C.sizeof(Result) // 8)
        return np.ascontiguousarray(raw[:, 1:5])  # fields 1..4 of mpmc_result: rd_energy, coulombic_energy, polarization_energy, vdw_energy


def _adopt(beads, res, n):
    per = ResultList(res, n)
    for i, b in enumerate(beads):
        b._obs_lazy = (res, i)
    return per


def pi_potential_local(beads: Sequence[System], host_positions: Optional[Sequence[np.ndarray]] = None):
    """local leg of SimulationControl::PI_calculate_potential (reference PathIntegral.cpp:752-805):
    returns (sums4 = ordered sums of {rd, coulombic, polarization, vdw} over this rank's beads, per-bead results, failed).
    host_positions: one C-contiguous (n, 3) float64 array per bead -- the coordinates travel inside the call
    (mpmc_pi_potential_local_host: bead b's upload overlaps the evaluation of the beads in front of it)."""
    L = lib()
    n = len(beads)
    arr = (C.c_void_p * max(n, 1))(*[b.handle for b in beads])
    sums = np.zeros(4)
    res = (Result * max(n, 1))()
    failed = C.c_int(0)
    if host_positions is not None:
        if len(host_positions) != n:
            raise ValueError("host_positions: one array per bead")
        keep = [np.ascontiguousarray(p, dtype=np.float64) for p in host_positions]
        ptrs = (C.POINTER(C.c_double) * max(n, 1))(*[_dp(p) for p in keep])
        rc = L.mpmc_pi_potential_local_host(arr, n, ptrs, _dp(sums), res, C.byref(failed))
    else:
        rc = L.mpmc_pi_potential_local(arr, n, _dp(sums), res, C.byref(failed))
    if rc != MPMC_OK:
        msg = b""
        for b in beads:
            msg = L.mpmc_last_error(b.handle) or msg
        raise MpmcError(rc, msg.decode())
    return sums, _adopt(beads, res, n), bool(failed.value)


def gibbs_energy(box_a: System, box_b: System):
    """both boxes of a Gibbs ensemble (reference SimulationControl.Gibbs.cpp:179-180): enqueued together, waited for together; the boxes
    usually live on two devices (System(..., device=0) / System(..., device=1)).  Returns (E_a, E_b)."""
    ra, rb = Result(), Result()
    rc = lib().mpmc_gibbs_energy(box_a.handle, box_b.handle, C.byref(ra), C.byref(rb))
    if rc != MPMC_OK:
        raise MpmcError(rc, ((lib().mpmc_last_error(box_a.handle) or b"") + b" / " + (lib().mpmc_last_error(box_b.handle) or b"")).decode())
    box_a.observables, box_b.observables = ra.as_dict(), rb.as_dict()
    return ra.energy, rb.energy


def gibbs_boltzmann_factor(movetype, temperature, init_energy, final_energy, N, volume, checkpoint_volume_0, current=(float("nan"), float("nan"))):
    """SimulationControl::boltzmann_factor_NVT_Gibbs (reference Gibbs.cpp:358-522).  Returns (status, [bf_a, bf_b], [energy_a, energy_b]);
    entries the reference leaves untouched come back as `current` / the final energies."""
    m = GibbsMove()
    m.movetype[0], m.movetype[1] = int(movetype[0]), int(movetype[1])
    m.temperature = float(temperature)
    for k in range(2):
        m.init_energy[k], m.final_energy[k], m.N[k], m.volume[k] = float(init_energy[k]), float(final_energy[k]), float(N[k]), float(volume[k])
    m.checkpoint_volume_0 = float(checkpoint_volume_0)
    bf = np.array(current, dtype=np.float64)
    en = np.array(final_energy, dtype=np.float64)
    rc = lib().mpmc_gibbs_boltzmann_factor(C.byref(m), _dp(bf), _dp(en))
    return rc, bf, en


def pi_allreduce(beads: Sequence[System]):
    """SimulationControl::PI_calculate_potential for ONE process that drives several GPUs (bead b usually on device b mod G): evaluation with
    one host thread per device, per-bead values gathered over RCCL (ncclCommInitAll communicator), ordered sum s = 0..P-1.
    Returns (sums4, per-bead results, failed) like pi_potential_local."""
    L = lib()
    n = len(beads)
    arr = (C.c_void_p * max(n, 1))(*[b.handle for b in beads])
    sums = np.zeros(4)
    res = (Result * max(n, 1))()
    failed = C.c_int(0)
    rc = L.mpmc_pi_allreduce(arr, n, _dp(sums), res, C.byref(failed))
    if rc != MPMC_OK:
        raise MpmcError(rc, ((L.mpmc_last_error(beads[0].handle) if beads else b"") or L.mpmc_comm_last_error(None) or b"").decode())
    return sums, _adopt(beads, res, n), bool(failed.value)


def pi_allreduce_info(beads: Sequence[System]):
    """(number of distinct devices the beads live on, size of the process-wide RCCL communicator pi_allreduce made for them; 0 = not yet)."""
    n = len(beads)
    arr = (C.c_void_p * max(n, 1))(*[b.handle for b in beads])
    nd, nr = C.c_int(0), C.c_int(0)
    rc = lib().mpmc_pi_allreduce_info(arr, n, C.byref(nd), C.byref(nr))
    if rc != MPMC_OK:
        raise MpmcError(rc, "mpmc_pi_allreduce_info")
    return nd.value, nr.value


def rccl_version() -> int:
    v = C.c_int(0)
    rc = lib().mpmc_rccl_version(C.byref(v))
    if rc != MPMC_OK:
        raise MpmcError(rc, (lib().mpmc_comm_last_error(None) or b"").decode())
    return v.value


def rccl_library_path() -> str:
    """the file RCCL's entry points were resolved from and why that copy (mpmc_rccl_library_path); "" if RCCL could not be opened."""
    return (lib().mpmc_rccl_library_path() or b"").decode()


def loaded_rocm_libs() -> Dict[str, list]:
    """the ROCm runtime libraries mapped into THIS process (/proc/self/maps): {"libamdhip64": [paths], "librccl": [...], ...}.
    One path per name = one ROCm in the process (what a rank of the multi-GPU job must show)."""
    names = ("libamdhip64", "librccl", "libhsa-runtime64", "librocm_smi64", "libmpmc_energy", "libtorch_hip")
    found: Dict[str, list] = {k: [] for k in names}
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                path = ln.split(None, 5)[-1].strip() if ln.count("/") else ""
                base = os.path.basename(path)
                for k in names:
                    if base.startswith(k + ".so") and path not in found[k]:
                        found[k].append(path)
    except OSError:
        pass
    return {k: v for k, v in found.items() if v}


class Comm:
    """RCCL communicator of the C ABI, one process per GPU: rank 0 makes the id (`Comm.unique_id()`), the launcher's own channel carries
    the 128 bytes to the other ranks (bench.py: torch.distributed's store), every rank constructs `Comm(n_ranks, rank, id, device)`."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        rc = lib().mpmc_comm_unique_id(buf)
        if rc != MPMC_OK:
            raise MpmcError(rc, (lib().mpmc_comm_last_error(None) or b"").decode())
        return buf.raw

    def __init__(self, n_ranks: int, rank: int, uid: bytes, device: int):
        self._L = lib()
        self._h = C.c_void_p()
        if len(uid) != Comm.ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        rc = self._L.mpmc_comm_init_rank(C.byref(self._h), int(n_ranks), int(rank), uid, int(device))
        if rc != MPMC_OK:
            raise MpmcError(rc, (self._L.mpmc_comm_last_error(None) or b"").decode())
        self.n_ranks, self.rank, self.device = int(n_ranks), int(rank), int(device)

    def _check(self, rc: int):
        if rc != MPMC_OK:
            raise MpmcError(rc, (self._L.mpmc_comm_last_error(self._h) or b"").decode())

    def allgather(self, local: np.ndarray) -> np.ndarray:
        """(count,) fp64 of this rank -> (n_ranks, count) in rank order."""
        a = np.ascontiguousarray(local, dtype=np.float64).reshape(-1)
        out = np.zeros((self.n_ranks, a.size))
        self._check(self._L.mpmc_comm_allgather_f64(self._h, _dp(a), a.size, _dp(out)))
        return out

    def gather_beads(self, local: np.ndarray) -> np.ndarray:
        """(n_local, stride) of this rank's beads (local-slot order) -> (P, stride) in bead order; bead s = rank s % n_ranks, slot s // n_ranks."""
        a = np.ascontiguousarray(local, dtype=np.float64)
        n_local = a.shape[0]
        a2 = a.reshape(n_local, -1)
        out = np.zeros((n_local * self.n_ranks, a2.shape[1]))
        self._check(self._L.mpmc_pi_gather_beads(self._h, _dp(a2), n_local, a2.shape[1], _dp(out)))
        return out.reshape((n_local * self.n_ranks,) + a.shape[1:])

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.mpmc_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pi_finish(sums4_global: np.ndarray, P: int):
    s = np.ascontiguousarray(sums4_global, dtype=np.float64)
    obs = np.zeros(4)
    v = lib().mpmc_pi_finish(_dp(s), int(P), _dp(obs))
    return v, obs


def pi_chain_mass_length2(coms: np.ndarray, mol_mass: np.ndarray, movable: Optional[np.ndarray] = None) -> float:
    """SimulationControl::PI_chain_mass_length2_ENTIRE_SYSTEM (reference PathIntegral.cpp:851-965).
    coms: (P, n_molecules, 3) centres of mass of the P images; host arithmetic inside libmpmc_energy.so."""
    c = np.ascontiguousarray(coms, dtype=np.float64)
    P, nmol, _ = c.shape
    m = np.ascontiguousarray(mol_mass, dtype=np.float64)
    mv = None if movable is None else np.ascontiguousarray(movable, dtype=np.int32)
    return float(lib().mpmc_pi_chain_mass_length2(P, nmol, _dp(c), _dp(m), _ip(mv)))


def pi_kinetic(chain_mass_len2: float, N: float, P: int, temperature: float, orient_mu_len2: float = 0.0) -> float:
    """SimulationControl::PI_calculate_kinetic (reference PathIntegral.cpp:806-824), Kelvin."""
    return float(lib().mpmc_pi_kinetic(float(chain_mass_len2), float(orient_mu_len2), float(N), int(P), float(temperature)))
