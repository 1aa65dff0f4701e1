#!/usr/bin/env python3
"""bench.py -- energy evaluations per second on the BASELINE.json headline workload.

Workload (BASELINE.json configs[3] sharded as configs[4]): a P = 32 bead path-integral ensemble of the 10 000-atom
polarizable box (LJ + LRC, Ewald real/reciprocal/self with kmax 7, Thole static field + 10 Jacobi dipole
iterations, polar_ewald on).  One "step" = one SimulationControl::PI_calculate_potential
(reference PathIntegral.cpp:752-805): a full stateless energy() of every bead + the 4-scalar combine.
Beads are sharded round-robin over the ranks (one process per GPU); the combine is ONE ncclAllGather of 4 fp64 per
bead on RCCL over xGMI inside libmpmc_energy.so (mpmc_pi_gather_beads), followed by the reference's ordered sum.
Total work is fixed => "scaling": "strong".  At --gpus 1 all 32 beads run on the one GPU.

value = (P * steps) / wall  [energy evaluations / s, whole job], inputs resident in HBM before the timed region.

A rank process carries ONE ROCm: it imports neither torch nor any other GPU runtime -- libmpmc_energy.so (linked against /opt/rocm's
libamdhip64) and the librccl next to that runtime are all it maps (config.ranks[*].rocm_libs lists them from /proc/self/maps).
  * `python3 bench.py --gpus N` started bare: this process -- before it touches HIP -- starts N children of itself
    (mpmcxx_amd/ranks.py: subprocess.Popen, never an exec) and relays rank 0's JSON line and the job's exit code;
  * under a launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) it is one of the ranks and takes RANK /
    LOCAL_RANK / WORLD_SIZE from the environment; the ranks meet over a loopback socket hub (ranks.Hub), which carries RCCL's 128-byte
    unique id, the votes around ncclCommInitRank, the timing barriers and the MAX over ranks;
  * --combine-impl torch is the opt-in alternative (imports torch, torch.distributed for everything); --combine-impl hub sends the 4
    doubles per bead over the socket hub (what the reference's MPI_Allgather does: rehearsal of the N-rank path on a one-GPU box, and
    the fall-back when ncclCommInitRank returns an error);
  * --launch inprocess keeps ONE process that drives the N devices through mpmc_pi_allreduce (one host thread per device,
    ncclCommInitAll) -- the shape of the reference's OpenMP build (PathIntegral.cpp:772-779).

The JSON line: the contract's keys, `roofline` (dominant kernel alone on the GPU by HIP events on its own stream; inside it, because
the driver's record keeps that object whole: other_kernels, whole_step, other_configs = BASELINE configs[1..3] + the dense-matrix form of
configs[3], pcie_inclusive_value, four_beads_in_flight) and `cpu_baseline` (the reference's own object code on one host core).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec; the fp64 matrix peak is the same number)
FP64_FMA_SUSTAINED_TFLOPS = 59.0  # back-to-back v_fma_f64 on all SIMDs (tools/microbench_f64.hip, profiles/r01_microbench_f64.txt)
# Algorithmic fp64 flops per unordered pair, counted the way the peak is: FMA = 2, add / sub / mul = 1, v_rsq / v_rcp = 1;
# rounding, conversions, compares and lane moves are not flops.  "nu" = dimensions of the pair's tile pair WITHOUT a tile-pair-wide
# periodic image.  The counts follow the arithmetic the kernels do (checked against the PMC's FMA / MUL / ADD counts, profiles/*_pmc_stalls.txt).
#   Jacobi contraction, (a, b) read from the store: 39 + 3 nu; recomputed (far field): 52 + 3 nu
#   pair sweep, every walked pair: 16 + 3 nu; inside the cutoff: LJ 9 + erfc table 14 (+ field 27) = 23 / 50; Thole (a, b) of a stored pair: 47
#   reciprocal space: K N (6 + ~40) for the structure factors, the same again for the field
FLOP_JAC_STORED, FLOP_JAC_FAR, FLOP_JAC_PER_NU = 39.0, 52.0, 3.0
FLOP_SWEEP_BASE, FLOP_SWEEP_PER_NU, FLOP_SWEEP_CUTOFF, FLOP_SWEEP_CUTOFF_NO_FIELD, FLOP_SWEEP_STORE = 16.0, 3.0, 50.0, 23.0, 47.0
FLOP_RECIP_PER_K_ATOM = 2 * 46.0
CONFIGS = (("configs[1]: 1 000-atom LJ box (rd_only), 1 GPU", "lj1000"), ("configs[2]: 10 000-atom LJ + Ewald box (kmax 7), 1 GPU", "ion10k_es"),
           ("configs[3]: 10 000-atom LJ + Ewald + Thole box (10 Jacobi iterations), 1 GPU", "ion10k_polar"))


def build_case(natoms: int, workdir: str):
    """the config-4 box through the reference's own file formats (so every loader sees the same doubles)."""
    from mpmcxx_amd import gen_box, pqr

    if natoms == 10000:
        inp, _ = gen_box.materialize("ion10k_polar", workdir)
    else:  # reduced sizes are for quick functional runs only (NOT a valid benchmark number)
        L = 86.0 * (natoms / 10000.0) ** (1.0 / 3.0)
        gen_box.write_pqr(os.path.join(workdir, "box.pqr"), gen_box.lattice_box(natoms, L, 13))
        gen_box.write_input(os.path.join(workdir, "box.in"), "box.pqr", gen_box.cubic(L), dict(gen_box.POLAR_OPTS))
        inp = os.path.join(workdir, "box.in")
    return pqr.load_case(inp)


def bead_positions(pos: np.ndarray, bead: int) -> np.ndarray:
    """bead b = base positions + Gaussian bead displacement (sigma 0.05 A), numpy default_rng([17, b]), on the
    6-decimal grid of a PQR file: exactly the boxes tests/golden/ion10k_polar_bead{0,1}.json hold reference energies for."""
    from mpmcxx_amd import gen_box

    return gen_box.bead_positions(pos, bead)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(kind: str, atoms, basis, opts, workdir: str, gpu_bead0):
    """bead 0 of the ensemble on one host core.  gpu_bead0: what the HIP path returned for the same configuration."""
    if kind == "none":
        return None
    if kind == "auto":
        kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")) else "port"
    n = atoms["pos"].shape[0]
    common = {"unit": "energy-evals/s", "cores": 1, "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
    if kind == "reference":
        import subprocess

        from mpmcxx_amd import gen_box

        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        if not os.path.exists(harness):
            raise RuntimeError("oracle/_ref/ref_harness is not present (built only where /root/reference exists)")
        rows = gen_box.lattice_box(n, float(np.asarray(basis)[0][0]), 13)  # bead 0 through the reference's own file formats
        for r, (x, y, z) in zip(rows, atoms["pos"]):
            r.x, r.y, r.z = float(x), float(y), float(z)
        gen_box.write_pqr(os.path.join(workdir, "bead0.pqr"), rows)
        gen_box.write_input(os.path.join(workdir, "bead0.in"), "bead0.pqr", np.asarray(basis).tolist(), dict(gen_box.POLAR_OPTS))
        t0 = time.time()
        p = subprocess.run([harness, "bead0.in", "--time", "1"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        res = json.loads(p.stdout[p.stdout.rfind("\n{") + 1:])
        sec = res["time_mean_s"]
        out = dict(common, value=1.0 / sec, kind="reference",
                   sample=f"1 steady-state full-recompute System::energy() of bead 0 of the ensemble ({n} atoms) by the reference's object code "
                          f"(oracle/_ref/ref_harness; {sec:.2f} s; harness wall incl. pair-list setup {time.time() - t0:.0f} s)",
                   energy=res["total"])
        ref = {"energy": res["total"], "rd_energy": res["rd"], "coulombic_energy": res["es"], "polarization_energy": res["polar"]}
    else:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the checker: this leg is the only place bench.py touches oracle/
        from oracle import OracleSystem

        S = OracleSystem(atoms, basis, opts)
        t0 = time.time()
        r = S.energy(want_atoms=False)  # ONE full evaluation (~10 s of one host core at 10 000 atoms): the bounded CPU sample
        sec = time.time() - t0
        out = dict(common, value=1.0 / sec, kind="port",
                   sample=f"1 full evaluation of bead 0 of the ensemble ({n} atoms) by the scalar C oracle (oracle/mpmc_oracle.c, the dense-A algorithm "
                          f"of the reference; {sec:.1f} s)", energy=r["energy"])
        ref = r
    if gpu_bead0 is not None:
        errs = {k: abs(gpu_bead0[k] - ref[k]) / abs(ref[k]) for k in ("energy", "rd_energy", "coulombic_energy", "polarization_energy") if ref.get(k)}
        out["parity_rel_err"] = max(errs.values())
        out["parity_rel_err_by_term"] = errs
        out["gpu_energy_same_configuration"] = gpu_bead0["energy"]
    return out


def sweep_flops(ps, cut: float, polar: bool) -> float:
    return (FLOP_SWEEP_BASE * ps["pairs_swept"] + FLOP_SWEEP_PER_NU * ps["nonuniform_dims_x_pairs_swept"]
            + (FLOP_SWEEP_CUTOFF if polar else FLOP_SWEEP_CUTOFF_NO_FIELD) * cut + (FLOP_SWEEP_STORE * ps["pairs_stored"] if polar else 0.0))


def fp64_entry(kernel: str, flops: float, ms: float, clock: str) -> dict:
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": kernel, "bound": "fp64_valu", "avg_launch_ms": ms, "algorithmic_flops": flops, "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS, "clock": clock}


def dense_symv_entry(n: int, ms: float) -> dict:
    """the reference's 3N x 3N A matrix in HBM, contraction on v_mfma_f64_16x16x4_f64: HBM-bound.  A is symmetric (thole_amatrix
    System.Energy.cpp:2748-2757) and the contraction reads its upper BLOCK triangle only -- tile pairs I <= J of 192 x 192 doubles --
    forming both products per block: those are the algorithmic bytes of a symmetric matrix-vector product."""
    n3 = 3 * ((n + 63) // 64 * 64)
    ntl = n3 // 192
    alg = 8.0 * 192 * 192 * (ntl * (ntl + 1) // 2)
    ach = alg / (ms * 1e-3) / 1e9
    return {"kernel": "k_dense_symv", "bound": "hbm", "avg_launch_ms": ms, "algorithmic_bytes_per_launch": alg, "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "whole_matrix_bytes": 8.0 * n3 * n3,
            "clock": "HIP events around every launch on the kernel's stream (kernel alone on the GPU)",
            "mfma_side": {"issued_tflops": 2.0 * 16 * 192 * 192 * (ntl * (ntl - 1) + ntl) / (ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "useful_fraction": 1.0 / 16.0,
                          "what": "v_mfma_f64_16x16x4_f64 with the vector replicated over one operand: two products per off-diagonal block, one per diagonal block"}}


def other_configs(headline_beads: int, headline_value: float, device: int, workdir: str):
    """BASELINE configs[1..3] on this GPU + configs[3] taken literally (dense 3N x 3N matrix), measured right
    behind the timed region: one system evaluated back to back ("alone": what System::mc sees, MonteCarlo.cpp:47), 32 copies with jittered
    positions in flight (energy.pi_potential_local), and the roofline entry of the configuration's dominant kernel.  tools/config_rates.py
    is the same loop with more repetitions."""
    from mpmcxx_amd import energy, gen_box, pqr

    out = []
    for label, name in CONFIGS:
        inp, _ = gen_box.materialize(name, workdir)
        atoms, basis, opts = pqr.load_case(inp)
        S = energy.System(atoms, basis, opts, device=device)
        e = S.energy()
        S.energy()
        reps = 400 if name == "lj1000" else (100 if name == "ion10k_es" else 50)  # (the same loop lengths as tools/config_rates.py, give or take: short loops read the clock ramp)
        t0 = time.perf_counter()
        for _ in range(reps):
            S.energy()
        alone = (time.perf_counter() - t0) / reps
        r = S.observables
        entry = {"workload": label, "energy_K": e, "evals_per_s_alone": 1.0 / alone, "ms_per_eval_alone": alone * 1e3}
        if name == "lj1000":
            # the whole evaluation is ONE launch (k_pair_fused<TAIL>); its duration is the evaluation's wall time -- dispatch, 64-step latency
            # chain and the polled result included: a latency regime, not a roofline one
            npairs = atoms["pos"].shape[0] * (atoms["pos"].shape[0] - 1) // 2
            fl = (FLOP_SWEEP_BASE + 3 * FLOP_SWEEP_PER_NU) * npairs + 10.0 * float(r["n_lj_in_cutoff"])
            entry.update(fp64_entry("k_pair_fused<TAIL> (single launch)", fl, alone * 1e3, "wall time of the evaluation (one launch + dispatch + the host's poll)"))
        else:
            ps = S.pair_stats()
            ms = S.time_kernel("pair", 30)
            fl = sweep_flops(ps, float(r["n_es_in_cutoff"]), name == "ion10k_polar")
            entry.update(fp64_entry("k_pair_sweep" if S.last_pair_kernel() == "sweep" else "k_pair_fused", fl, ms,
                                    "30 launches back to back between ONE pair of HIP events on the kernel's stream (kernel alone on the GPU)"))
            entry["pairs_in_cutoff"] = int(r["n_es_in_cutoff"])
        S.close()
        if name == "ion10k_polar" and headline_beads >= 32 and headline_value:
            entry["evals_per_s_in_flight"] = headline_value  # the headline IS this configuration with 32 beads in flight
        else:
            copies = []
            for b in range(32):
                a = dict(atoms)
                a["pos"] = atoms["pos"] + np.random.default_rng([17, b]).normal(scale=0.05, size=atoms["pos"].shape)
                copies.append(energy.System(a, basis, opts, device=device))
            energy.pi_potential_local(copies)
            steps = 20 if name == "lj1000" else (6 if name == "ion10k_es" else 2)
            t0 = time.perf_counter()
            for _ in range(steps):
                energy.pi_potential_local(copies)
            entry["evals_per_s_in_flight"] = (steps * 32) / (time.perf_counter() - t0)
            for c in copies:
                c.close()
        out.append(entry)
    # configs[3] literal: "Thole iterative dipole solve as dense 3N MFMA" -- the A matrix of thole_amatrix (System.Energy.cpp:2661-2770) in
    # device memory (7.2 GB), contracted ten times per evaluation by k_dense_symv.  One system.
    inp, _ = gen_box.materialize("ion10k_polar", workdir)
    atoms, basis, opts = pqr.load_case(inp)
    S = energy.System(atoms, basis, dict(opts, solver="dense"), device=device)
    e = S.energy()
    S.energy()
    t0 = time.perf_counter()
    for _ in range(3):
        S.energy()
    alone = (time.perf_counter() - t0) / 3
    S.set_profiling(True)
    S.energy()
    tm = S.timings(reset=True)["dipole_iter"]
    S.set_profiling(False)
    mem_total, _ = S.memory_usage()
    n = atoms["pos"].shape[0]
    S.close()
    entry = {"workload": "configs[3] literal: the same box, Thole solve over the dense 3N x 3N A matrix on the fp64 matrix cores (--solver dense), 1 GPU",
             "energy_K": e, "evals_per_s_alone": 1.0 / alone, "ms_per_eval_alone": alone * 1e3, "evals_per_s_in_flight": None, "device_bytes": mem_total}
    entry.update(dense_symv_entry(n, tm["ms"] / max(tm["launches"], 1)))
    out.append(entry)
    return out


class TorchGroup:
    """--combine-impl torch (opt-in): torch.distributed in the hub's place.  Importing torch maps PyTorch's bundled ROCm into the rank."""

    def __init__(self, rank, world, backend, device):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.dev = f"cuda:{device}" if backend == "nccl" else "cpu"
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(self.dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def exchange(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.dist.barrier()

    def max(self, x):
        return max(self.exchange(float(x)))

    def close(self):
        self.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launch", choices=["ranks", "inprocess"], default="ranks",
                    help="ranks: one process per GPU (started by this command itself when no launcher did); inprocess: ONE process drives the "
                         "--gpus devices through mpmc_pi_allreduce (one host thread per device, ncclCommInitAll)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--beads", type=int, default=32)
    ap.add_argument("--natoms", type=int, default=10000)
    ap.add_argument("--solver", default="auto")
    ap.add_argument("--concurrency", choices=["async", "serial"], default="async",
                    help="async: all local beads enqueued on their own streams before the first wait; serial: one bead at a time")
    ap.add_argument("--cpu-baseline", choices=["auto", "port", "reference", "none"], default="auto",
                    help="auto: the reference's own object code (oracle/_ref/ref_harness, ~25 s) where it was built, else the C port of the oracle")
    ap.add_argument("--combine", choices=["gather", "reduce"], default="gather", help="reduce: only with --combine-impl torch")
    ap.add_argument("--combine-impl", choices=["cabi", "hub", "torch"], default="cabi",
                    help="cabi: ncclAllGather inside libmpmc_energy.so over a communicator from mpmc_comm_init_rank, torch-free ranks (falls back to "
                         "hub when RCCL returns an error, recorded in config.combine_impl); hub: the 4 doubles per bead over the loopback socket hub; "
                         "torch: torch.distributed (imports torch)")
    ap.add_argument("--comm-init-timeout", type=float, default=120.0,
                    help="seconds a rank waits for ncclCommInitRank + one probe all-gather (blocking collectives) before the job gives up on the run")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="diagnostic: per-launch HIP events on one bead's stream INSIDE the timed region (by default it carries no instrumentation)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs[1..3] + dense pass (roofline.other_configs)")
    ap.add_argument("--no-extra-passes", action="store_true", help="diagnostic: skip the isolated-kernel, PCIe-inclusive and 4-bead passes after the timed region")
    ap.add_argument("--host-positions", action="store_true",
                    help="re-upload every bead's positions from host buffers inside each timed step (the PCIe-inclusive rate; the default run "
                         "measures it in a short extra pass and reports it as pcie_inclusive_value, never as value)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl", help="--combine-impl torch only: nccl (= RCCL) or gloo")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this device")
    ap.add_argument("--beads-per-gpu-rehearsal", type=int, default=0,
                    help="one GPU, N beads in flight on it -- the per-GPU load of an (--beads / N)-GPU run, everything else as in that run; NOT the headline")
    ap.add_argument("--configure", action="append", default=[], metavar="KEY=VALUE",
                    help="measurement switch for every context of this run (mpmc_debug_configure, e.g. side_stream=0 pair_kernel=1); repeatable")
    args = ap.parse_args()

    inprocess = args.launch == "inprocess" and args.gpus > 1
    from mpmcxx_amd import ranks  # (pure Python: no GPU runtime is loaded by this import)

    if args.gpus > 1 and not inprocess and "WORLD_SIZE" not in os.environ:
        # bare multi-GPU command: this parent only starts the ranks (fresh children, before anything touched HIP) and relays their exit code
        print(f"[bench.py] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks of {os.path.basename(__file__)} (mpmcxx_amd.ranks.spawn)",
              file=sys.stderr, flush=True)
        sys.exit(ranks.spawn(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], {"MPMC_BENCH_SELF_LAUNCHED": "1"}))

    # RCCL between the processes of one node goes through dmabuf IPC on this pool: the host driver does not support the legacy IPC mode, and
    # without this variable the first multi-process collective fails with "hipIpcGetMemHandle: invalid argument".  It must be in the
    # environment before the HIP runtime starts, i.e. before libmpmc_energy.so is loaded (a no-op for one rank).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local_rank = ranks.env_rank()
    use_torch = args.combine_impl == "torch" and world > 1
    if use_torch:
        import torch  # noqa: F401  (opt-in; FIRST, so that the library binds to the runtime PyTorch carries: still one HIP runtime in the process)

    from mpmcxx_amd import energy, pi

    for kv in args.configure:
        k_, _, v_ = kv.partition("=")
        energy.configure(k_, float(v_))
    if inprocess:
        if world != 1:
            raise SystemExit("--launch inprocess is ONE process: start it without a launcher")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    n_visible = energy.device_count()
    if n_visible < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the energy path has no CPU fallback)")
    if args.force_device is not None:
        local_rank = args.force_device
    # devices this process drives: its own (one rank per GPU) or all of them (--launch inprocess: bead b on devices[b mod N])
    n_dev = args.gpus if inprocess else 1
    devices = [local_rank] * n_dev if (not inprocess or args.force_device is not None) else list(range(n_dev))
    if max(devices) >= n_visible:
        raise SystemExit(f"--gpus {args.gpus}: device {max(devices)} asked for, only {n_visible} HIP device(s) visible")

    # ---- the job's channel and the collective of the combine ---------------------------------------------------------------
    group = None
    if world > 1:
        group = TorchGroup(rank, world, args.dist_backend, local_rank) if use_torch else ranks.Hub.join(rank, world)
    comm, comm_stuck, comm_failed_why = None, False, ""
    combine_impl = "none (one rank)"
    rccl_ver, rccl_path = None, ""
    try:
        rccl_ver, rccl_path = energy.rccl_version(), energy.rccl_library_path()
    except energy.MpmcError:
        pass
    if world > 1:
        if use_torch:
            combine_impl = f"torch.distributed ({args.dist_backend})"
        elif args.combine_impl == "hub":
            combine_impl = "loopback socket hub (mpmcxx_amd.ranks.Hub.gather_beads: a host all-gather, like the reference's MPI_Allgather)"
        else:
            ready_here = bool(rccl_ver) and 0 <= local_rank < n_visible
            comm, comm_stuck, comm_failed_why = ranks.join_rccl_communicator(group, ready_here, energy.Comm.unique_id,
                                                                             lambda uid: energy.Comm(world, rank, uid, local_rank),
                                                                             args.comm_init_timeout, energy.MpmcError)
            if comm is not None:
                combine_impl = "libmpmc_energy.so: mpmc_pi_gather_beads (ncclAllGather, communicator from mpmc_comm_init_rank)"
            else:
                combine_impl = f"loopback socket hub (FALL-BACK: {comm_failed_why})"
    if comm_stuck:
        # a helper thread of some rank is still inside a blocking RCCL call: a process in that state is not timed.  One degraded line,
        # no teardown of anything RCCL might hold, non-zero exit.
        if rank == 0:
            print(json.dumps({"metric": "energy-evals/sec (10k-atom LJ+Ewald+polar box); 1/2/4/8-GPU scaling", "value": None, "unit": "energy-evals/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                              "vs_baseline": None, "dtype": "f64", "data": "synthetic", "cabi_comm_timed_out": True, "degraded": comm_failed_why,
                              "config": {"workload": "not run", "rccl_library": rccl_path, "rccl_version": rccl_ver}}), flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)

    P = args.beads
    n_gpus = world * n_dev
    if P % n_gpus:
        raise SystemExit("--beads must be a multiple of --gpus")
    rehearsal = int(args.beads_per_gpu_rehearsal)
    if rehearsal:
        if n_gpus != 1:
            raise SystemExit("--beads-per-gpu-rehearsal is a one-GPU run")
        P = rehearsal
    workdir = tempfile.mkdtemp(prefix="mpmc_bench_")
    atoms, basis, opts = build_case(args.natoms, workdir)
    opts = dict(opts)
    opts["solver"] = args.solver
    n = atoms["pos"].shape[0]

    mine = pi.beads_of_rank(P, rank, world)
    beads = []
    for k, b in enumerate(mine):
        a = dict(atoms)
        a["pos"] = bead_positions(atoms["pos"], b)
        beads.append(energy.System(a, basis, opts, device=devices[k % n_dev]))
    if inprocess and args.force_device is not None:
        # rehearsal on one GPU: bead k counts as living on "device" k mod N (the library's test hook "virtual_device"), so that the step below
        # runs the real thread-per-device path -- worker threads, hand-off, ordered combine -- with a host copy in RCCL's place
        for k, s in enumerate(beads):
            s.configure("virtual_device", k % n_dev)
    host_pos = [np.ascontiguousarray(bead_positions(atoms["pos"], b)) for b in mine]
    state = {"host_positions": bool(args.host_positions), "per": None}

    def local_eval(systems=None, hp_all=None):
        # host_positions: the boundary as a host program with its own coordinates uses it -- every bead's positions arrive in host memory
        # inside the call (mpmc_pi_potential_local_host: bead b's upload is followed at once by its enqueue)
        systems = beads if systems is None else systems
        hp = (host_pos if hp_all is None else hp_all) if state["host_positions"] else None
        if args.concurrency == "async":
            _, per, _failed = energy.pi_potential_local(systems, host_positions=hp)
            state["per"] = per
            return per.table4()  # (the four combined terms straight from the library's result block: no per-bead dicts inside the step)
        per = []
        for k, s in enumerate(systems):
            if hp is not None:
                s.update_positions(0, hp[k])
            s.energy()
            per.append(s.observables)
        state["per"] = per
        return np.array([[p["rd_energy"], p["coulombic_energy"], p["polarization_energy"], p["vdw_energy"]] for p in per])

    if inprocess:
        if args.host_positions or args.concurrency != "async":
            raise SystemExit("--launch inprocess runs the default step only (resident positions, all beads enqueued before the first wait)")
        combine_impl = "libmpmc_energy.so: mpmc_pi_allreduce (one host thread per device, ncclAllGather on an ncclCommInitAll communicator)"

        def step():
            # SimulationControl::PI_calculate_potential in ONE call of the C ABI: evaluation on every device, per-bead values gathered over
            # RCCL, the reference's ordered sum s = 0..P-1 (PathIntegral.cpp:786-801), then / P (mpmc_pi_finish)
            sums, per, _failed = energy.pi_allreduce(beads)
            state["per"] = per
            return energy.pi_finish(sums, P)
    elif use_torch:
        def step():
            return pi.pi_calculate_potential(local_eval, P, rank, world, mode=args.combine, device=group.dev)
    else:
        gatherer = comm if comm is not None else (group if world > 1 else None)

        def step():
            return pi.pi_calculate_potential(local_eval, P, rank, world, comm=gatherer)

    def fence():
        for d in sorted(set(devices)):
            energy.device_synchronize(d)
        if group is not None:
            group.barrier()

    def timed(k, fn=None):
        fn = fn or step
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            vv, oo = fn()
        fence()
        d = time.perf_counter() - t0
        if group is not None:
            d = group.max(d)
        return d, vv, oo

    def collect(systems):
        agg = {}
        for s in systems:
            for k, tv in s.timings(reset=True).items():
                a = agg.setdefault(k, {"ms": 0.0, "launches": 0})
                a["ms"] += tv["ms"]
                a["launches"] += tv["launches"]
        return agg

    for _ in range(args.warmup):
        v, obs = step()
    # The timed region carries NO instrumentation: per-launch HIP events cost the stream that carries them about 5 us of back-to-back
    # dispatch per launch, and with few beads in flight the instrumented bead is the tail of every step.
    events_in_region = bool(args.events_in_timed_region)
    for k, s in enumerate(beads):
        s.set_profiling(events_in_region and k == 0)
        s.timings(reset=True)
    dt, v, obs = timed(args.steps)
    gpu_bead0 = dict(state["per"][0]) if (rank == 0 and state["per"]) else None
    extra = not args.no_extra_passes
    if not events_in_region and extra:
        # what the kernels look like with the other beads in flight: a SEPARATE short pass behind the timed region, events on bead 0's stream
        beads[0].set_profiling(True)
        for _ in range(2):
            step()
        fence()
    agg = collect(beads)
    for s in beads:
        s.set_profiling(False)
    iters = int(beads[0].observables.get("polar_iterations", 0)) if beads else 0
    mem_total, mem_tensor = beads[0].memory_usage() if beads else (0, 0)
    tiles = beads[0].tile_stats() if beads else {}
    pairs = beads[0].pair_stats() if beads else {}
    single = n_gpus == 1 and extra  # the passes below keep nobody waiting in a barrier: one-GPU runs only

    # ---- PCIe-inclusive rate: the same step with every bead's positions handed over in host memory --------------------------
    pcie = None
    if single and not args.host_positions:
        state["host_positions"] = True
        step()
        k_pcie = max(2, min(args.steps, 5))
        d2, _, _ = timed(k_pcie)
        state["host_positions"] = False
        pcie = {"value": P * k_pcie / d2, "steps": k_pcie,
                "what": "the same step with every bead's positions handed over in HOST memory inside every timed step (mpmc_pi_potential_local_host: "
                        "one 320 KB upload per bead from the context's pinned mirror, bead b's upload followed at once by its enqueue)"}

    # ---- four beads in flight: the per-GPU load of the 8-GPU run of this ensemble (without its 4-double collective) ----------
    four = None
    if single and P >= 32 and not rehearsal and args.concurrency == "async":
        sub = beads[:4]

        def step4():
            per4 = local_eval(sub, host_pos[:4])
            return float(per4.sum()), per4

        step4()
        d4, _, _ = timed(20, step4)
        four = {"evals_per_s": 4 * 20 / d4, "x8_over_headline": 8 * (4 * 20 / d4) / (P * args.steps / dt),
                "what": "4 of the beads in flight on this GPU, 20 steps: what one GPU of the 8-GPU run of the 32-bead ensemble does between collectives; "
                        "x 8 is that run's ceiling"}

    # ---- the kernels with NOTHING else on the GPU: one bead on one stream (HIP events on that stream) ---------------------------
    iso, back_to_back, back_to_back_runs = None, {}, {}
    pair_kernel_name = "k_pair_sweep"
    if single and rank == 0:
        energy.configure("side_stream", 0)  # one stream: every kernel alone on the GPU
        try:
            a = dict(atoms)
            a["pos"] = bead_positions(atoms["pos"], mine[0])
            S1 = energy.System(a, basis, opts, device=local_rank)
        finally:
            energy.configure("side_stream", -1)
        S1.energy()
        S1.set_profiling(True)
        for _ in range(3):
            S1.energy()
        iso = collect([S1])
        S1.set_profiling(False)
        S1.energy()
        for which, key in (("panel", "dipole_iter"), ("pair", "pair")):
            try:  # three batches, the median counts (the first batch behind an idle stretch runs on ramping clocks)
                runs = sorted(S1.time_kernel(which, 100 if which == "panel" else 40) for _ in range(3))
                back_to_back[key], back_to_back_runs[key] = runs[1], runs
            except energy.MpmcError:
                pass  # (dense / matrix-free solver: no panel kernel)
        pair_kernel_name = "k_pair_sweep" if S1.last_pair_kernel() == "sweep" else "k_pair_fused"
        S1.close()

    others = None
    if single and rank == 0 and not args.no_other_configs and not rehearsal and args.natoms == 10000:
        others = other_configs(P, P * args.steps / dt, local_rank, workdir)

    pmc = {}  # per-launch PMC figures of the committed profiling passes (rocprofv3 cannot run inside this process): HBM bytes, executed flops
    for rnd in ("r04", "r05"):
        pth = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
        try:
            with open(pth) as f:
                for kname, rec in json.load(f).items():
                    pmc[kname] = dict(rec, source=f"profiles/{rnd}_traffic.json (committed rocprofv3 --pmc passes of the builder, NOT counters of this run)")
        except (OSError, ValueError):
            pass
    K = 709 if int(opts.get("ewald_kmax", 7)) == 7 else None
    cut = float(gpu_bead0["n_es_in_cutoff"]) if gpu_bead0 else 0.0
    flops_jacobi = (FLOP_JAC_STORED * pairs.get("pairs_stored", 0) + FLOP_JAC_FAR * pairs.get("pairs_far", 0)
                    + FLOP_JAC_PER_NU * (pairs.get("nonuniform_dims_x_pairs_stored", 0) + pairs.get("nonuniform_dims_x_pairs_far", 0)))
    flops_pair = sweep_flops(pairs, cut, True) if pairs else 0.0
    flops_eval = flops_pair + iters * flops_jacobi + (FLOP_RECIP_PER_K_ATOM * K * n if K else 0.0)
    bytes_jacobi = 16.0 * 4096 * pairs.get("tile_pairs_stored", 0) + n * 80.0

    my_info = {"rank": rank, "local_rank": local_rank, "device": f"hip:{local_rank}", "device_name": energy.device_name(local_rank), "pid": os.getpid(),
               "beads": mine, "comm_n_ranks": (comm.n_ranks if comm is not None else None), "torch_imported": "torch" in sys.modules,
               "rocm_libs": energy.loaded_rocm_libs()}
    infos = [my_info]
    if inprocess:  # one process: one entry per device it drove; the communicator is the library's own (ncclCommInitAll inside mpmc_pi_allreduce)
        n_devs_seen, comm_size = energy.pi_allreduce_info(beads)
        infos = [dict(my_info, rank=g, local_rank=d, device=f"hip:{d}", device_name=energy.device_name(d), beads=[b for k, b in enumerate(mine) if k % n_dev == g],
                      comm_n_ranks=comm_size, distinct_devices=n_devs_seen) for g, d in enumerate(devices)]
    if group is not None:
        infos = group.exchange(my_info)

    if rank == 0:
        value = P * args.steps / dt
        ms_per_step = dt / args.steps * 1e3
        n_local = len(beads)
        solver_used = "dense" if args.solver == "dense" else ("compact" if mem_tensor > 0 else "matrix_free")
        in_region = {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in agg.items() if tv["launches"]}
        alone = {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in (iso or {}).items() if tv["launches"]}
        src = alone if alone else in_region

        def compute_entry(kernel, cls_key, flops, launches_per_step):
            """fp64 vector-issue roofline of one kernel.  avg_launch_ms: HIP events on the kernel's stream (alone on the GPU where the
            isolated pass ran); achieved = algorithmic flops / that duration."""
            ms = back_to_back.get(cls_key) or src.get(cls_key) or 1e30
            e = fp64_entry(kernel, flops, ms,
                           "median of three batches of 100 (pair sweep: 40) launches back to back between ONE pair of HIP events on the kernel's stream, "
                           "per launch (kernel alone on the GPU)" if back_to_back.get(cls_key) else "HIP events around every launch on the kernel's stream")
            e["algorithmic_flops_per_launch"] = e.pop("algorithmic_flops")
            e.update({"traffic": None, "launches_per_step": launches_per_step, "consistent": bool(ms * launches_per_step <= ms_per_step),
                      "frac_of_sustained_fma_rate": e["achieved"] / FP64_FMA_SUSTAINED_TFLOPS})
            if back_to_back.get(cls_key):
                e["avg_launch_ms_batches"] = back_to_back_runs.get(cls_key)
                if src.get(cls_key):
                    e["avg_launch_ms_event_pair_per_launch"] = src[cls_key]
            t = pmc.get(kernel)
            if t and t.get("natoms") == n:
                e["traffic"] = t.get("hbm_bytes_per_launch")
                e["traffic_source"] = t.get("source")
                e["pmc"] = t
                if t.get("trace_avg_launch_ms"):
                    e["frac_by_trace_clock"] = flops / (t["trace_avg_launch_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
                if t.get("valu_wave_insts_per_launch"):
                    # every VALU instruction an FMA on 64 lanes would be 128 flops: the share of that which is algorithmic work, and the
                    # ceiling of THIS instruction stream at the sustained issue rate
                    e["algorithmic_flops_per_valu_slot"] = flops / (128.0 * t["valu_wave_insts_per_launch"])
                    e["ceiling_frac_of_this_instruction_stream"] = e["algorithmic_flops_per_valu_slot"] * FP64_FMA_SUSTAINED_TFLOPS / FP64_VALU_PEAK_TFLOPS
            return e

        # dominant kernel: the one with the largest share of the device time of an evaluation (alone-on-the-GPU durations)
        share = {"dipole_iter": (src.get("dipole_iter") or 0.0) * iters, "pair": src.get("pair") or 0.0}
        dom = max(share, key=share.get) if any(share.values()) else "dipole_iter"
        jac_kernel = "k_dipole_iter_panel" if solver_used == "compact" else "k_dipole_iter_hybrid"
        if solver_used == "dense":
            roof = dense_symv_entry(n, src.get("dipole_iter") or 1e30)
            roof.update({"traffic": None, "launches_per_step": iters * n_local, "consistent": bool(roof["avg_launch_ms"] * iters * n_local <= ms_per_step)})
        elif dom == "pair":
            roof = compute_entry(pair_kernel_name, "pair", flops_pair, n_local)
        else:
            roof = compute_entry(jac_kernel, "dipole_iter", flops_jacobi, iters * n_local)
            ms = roof["avg_launch_ms"]
            roof["hbm_side"] = {"achieved": bytes_jacobi / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": bytes_jacobi / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_jacobi}
        roof["measured"] = ("extra pass right after the timed region: ONE bead on ONE stream, every kernel alone on the GPU.  rocprofv3 --kernel-trace reports "
                            "a little less for the same launch (profiles/*_serial_kernel_stats.csv; frac_by_trace_clock): its stamps leave out the dispatch "
                            "and completion time between consecutive kernels of a stream" if alone else
                            "HIP events on one bead's stream with the other beads in flight (stretched durations)")
        roof["tile_pairs"] = tiles
        roof["pairs"] = dict(pairs, in_cutoff=cut)
        roof["flop_model"] = {"jacobi_stored": FLOP_JAC_STORED, "jacobi_far": FLOP_JAC_FAR, "jacobi_per_nonuniform_dim": FLOP_JAC_PER_NU,
                              "sweep_base": FLOP_SWEEP_BASE, "sweep_per_nonuniform_dim": FLOP_SWEEP_PER_NU, "sweep_in_cutoff": FLOP_SWEEP_CUTOFF,
                              "sweep_in_cutoff_no_field": FLOP_SWEEP_CUTOFF_NO_FIELD, "sweep_stored": FLOP_SWEEP_STORE,
                              "what": "fp64 flops per unordered atom pair, FMA = 2 (top of bench.py, DESIGN.md section 3); pair counts are exact atom pairs "
                                      "per tile-pair class (mpmc_debug_pair_stats)"}
        roof["share_of_device_time_alone"] = share[dom] / max(sum((src.get(k) or 0.0) * (iters if k in ("dipole_iter", "reduce") else 1) for k in src), 1e-30)
        # whole-step view, which overlap cannot distort: algorithmic flops of one evaluation x evaluations per second per GPU
        roof["whole_step"] = {"algorithmic_flops_per_eval": flops_eval, "achieved": flops_eval * value / n_gpus / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops_eval * value / n_gpus / 1e12 / FP64_VALU_PEAK_TFLOPS,
                              "what": "pair sweep + iterations x Jacobi contraction + reciprocal space, x evaluations/s per GPU"}
        roof["in_timed_region"] = {"kernel_ms": in_region,
                                   "measured": ("INSIDE the timed region (--events-in-timed-region)" if events_in_region else
                                                "in a separate 2-step pass BEHIND the timed region (the timed region carries no events)"),
                                   "note": f"{args.concurrency}: {n_local} beads in flight on this GPU, HIP events on one bead's stream; a launch here "
                                           "shares the CUs with other beads' kernels, so its duration is stretched -- not a kernel time"}
        other = {}
        if alone.get("pair") and dom != "pair":
            other["pair"] = compute_entry(pair_kernel_name, "pair", flops_pair, n_local)
        if alone.get("dipole_iter") and dom == "pair" and solver_used != "dense":
            other["dipole_iter"] = compute_entry(jac_kernel, "dipole_iter", flops_jacobi, iters * n_local)
        roof["other_kernels"] = other
        roof["alone_kernel_ms"] = alone
        if others is not None:
            roof["other_configs"] = others
        if pcie is not None:
            roof["pcie_inclusive_value"] = pcie["value"]
            roof["pcie_inclusive"] = pcie
        if four is not None:
            roof["four_beads_in_flight"] = four

        out = {
            "metric": "energy-evals/sec (10k-atom LJ+Ewald+polar box); 1/2/4/8-GPU scaling",
            "value": value, "unit": "energy-evals/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{P}-bead path-integral ensemble of the {n}-atom polarizable box (BASELINE configs[3] x configs[4]): "
                                   f"LJ+LRC, Ewald kmax {opts['ewald_kmax']}, Thole exponential damping, {iters} Jacobi iterations, polar_ewald",
                       "natoms": n, "beads": P, "beads_per_gpu": P // n_gpus, "polar_solver": solver_used, "combine": args.combine,
                       "parallelism": f"beads sharded round-robin over {n_gpus} GPU(s); one all_gather of 4 fp64 per bead per step",
                       "world_size": world, "combine_impl": combine_impl,
                       "launch": ("inprocess: one process, one host thread per device" if inprocess else
                                  ("ranks started by bench.py itself (mpmcxx_amd.ranks.spawn: plain child processes)" if os.environ.get("MPMC_BENCH_SELF_LAUNCHED") else
                                   ("ranks started by an external launcher" if world > 1 else "one process"))),
                       "job_channel": ("none (one rank)" if group is None else ("torch.distributed" if use_torch else "loopback socket hub (mpmcxx_amd.ranks.Hub)")),
                       "rccl_version": rccl_ver, "rccl_library": rccl_path, "configure": args.configure, "ranks": infos},
            "V_mean_K": v, "obs_rd_es_pol_vdw": [float(x) for x in obs],
            "kernel_ms": in_region, "device_bytes_per_bead": mem_total, "instrumented_in_timed_region": events_in_region,
            "roofline": roof,
        }
        if comm_failed_why:
            out["cabi_comm_failed"] = comm_failed_why  # (an error RCCL returned, no thread left inside it: the run is timed with the hub's host all-gather)
        if rehearsal:
            out["config"]["rehearsal"] = (f"{rehearsal} beads in flight on ONE GPU: the per-GPU load of a {args.beads // rehearsal}-GPU run of the "
                                          f"{args.beads}-bead ensemble, without the 4-double collective.  NOT the headline workload")
        if pcie is not None:
            out["pcie_inclusive_value"] = pcie["value"]
        if args.host_positions:
            out["note_host_positions"] = "positions of every bead handed over in host memory inside every timed step (PCIe-inclusive rate, not the headline)"
        if n_gpus == 1 and args.cpu_baseline != "none":
            cpu = cpu_baseline(args.cpu_baseline, {**atoms, "pos": bead_positions(atoms["pos"], 0)}, basis, opts, workdir, gpu_bead0)
            if cpu is not None:
                out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if group is not None:
        group.barrier()  # (rank 0's CPU leg is over: nobody tears the channel down under it)
    for s in beads:
        s.close()
    if comm is not None:
        comm.close()
    if group is not None:
        group.close()


if __name__ == "__main__":
    main()
