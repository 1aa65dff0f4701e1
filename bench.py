#!/usr/bin/env python3
"""bench.py -- energy evaluations per second on the BASELINE.json headline workload.

Workload (BASELINE.json configs[3] sharded as configs[4]): a P = 32 bead path-integral ensemble of the 10 000-atom
polarizable box (LJ + LRC, Ewald real/reciprocal/self with kmax 7, Thole static field + 10 Jacobi dipole
iterations, polar_ewald on).  One "step" = one SimulationControl::PI_calculate_potential
(reference PathIntegral.cpp:752-805): a full stateless energy() of every bead + the 4-scalar combine.
Beads are sharded round-robin over the ranks (one process per GPU); the combine is ONE all-gather of 4 fp64 per
bead on RCCL over xGMI -- inside libmpmc_energy.so (mpmc_pi_gather_beads; --combine-impl cabi, the default) or through
torch.distributed (--combine-impl torch) -- followed by the reference's ordered sum.  Total work is fixed => "scaling": "strong".
At --gpus 1 all 32 beads run on the one GPU, i.e. 32 evaluations of the config-4 box per step.

value = (P * steps) / wall  [energy evaluations / s, whole job], inputs resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline     -- dominant kernel (the Thole dipole-iteration kernel, one launch per Jacobi iteration).  PRIMARY figures: the kernel
                  ALONE on the GPU (HIP events on the stream it is launched on, extra pass right after the timed region), the
                  duration that rocprofv3 --kernel-trace reports for a serial run (profiles/).  The duration the same kernel shows
                  INSIDE the timed region (beads overlap on 32 streams, so it is stretched by the neighbours) is kept under
                  "in_timed_region".  `consistent` = avg_launch_ms x launches per step <= ms_per_step.
  other_configs -- BASELINE configs[1..3] on this GPU, measured in a short pass after the timed region (one system at a time and 32 jittered
                  copies in flight), each with the fp64 roofline entry of its pair kernel.
  cpu_baseline -- ONE evaluation of bead 0 of the same ensemble on one core of this host: by the reference's own object code (kind
                  "reference", oracle/_ref/ref_harness, the default wherever that binary was built) or by the C port of the oracle
                  (kind "port"); `parity_rel_err` = |E_gpu(bead 0) - E_cpu(bead 0)| / |E_cpu|.

Launching.  `python3 bench.py --gpus N` runs by itself: when no launcher's environment is present (WORLD_SIZE unset) and N > 1, this
process -- before it imports torch or touches HIP -- starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
(never an exec), i.e. N fresh ranks, one per GPU, and relays rank 0's JSON line and the exit code.  Under a launcher (the driver's
`python -m torch.distributed.run ... bench.py --gpus N`) it is one of the ranks.  `--launch inprocess` keeps ONE process that drives the N
devices through mpmc_pi_allreduce (bead b on device b mod N, one host thread per device, ncclCommInitAll communicator) -- the shape of the
reference's OpenMP build (PathIntegral.cpp:772-779).
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
FP64_FMA_SUSTAINED_TFLOPS = 59.0  # what tools/microbench_f64.hip measures for back-to-back v_fma_f64 on all SIMDs (profiles/r01_microbench_f64.txt): the chip holds ~1.8 GHz under fp64 load
# Algorithmic fp64 flops per unordered pair, counted the way the peak is: FMA = 2, add / sub / mul = 1, v_rsq / v_rcp = 1;
# rounding (v_rndne), conversions, compares and lane moves are NOT flops.  "nu" = dimensions of the pair's tile pair WITHOUT a
# tile-pair-wide periodic image (k_classify): there the minimum image costs mul + (rint) + fma = 3 (Jacobi kernels, which may fuse) or
# mul + (rint) + mul + sub = 3 (pair sweep, unfused: the squared distance decides pair inclusion and must round like the reference);
# a uniform dimension costs nothing in the Jacobi kernels (the i-atom is shifted once) and 1 subtraction in the pair sweep.
#   Jacobi contraction, (a, b) read from the store:  d 3,  mu_j.d and mu_i.d 2 x (mul + 2 fma) 10,  b x dot 2,
#       F_i += -a mu_j + (b mu_j.d) d and the same for F_j: 12 fma 24                                                   = 39 + 3 nu
#   ... recomputed (far field): + r^2 (mul + 2 fma) 5, 2 / r = rsq 1 + one Newton step in product form (mul, fma, mul) 4, 4 / r^2, 8 / r^3,
#       3 / r^2 3 (the power of two goes onto the sums once per wave)                                                    = 52 + 3 nu
#       (rounds 1-3 and most of round 4: 55 -- Newton step as a correction, 2 mul + 2 fma, and one product more)
#   pair sweep (k_pair_sweep), every walked pair: per dimension 1 (subtraction from the pre-shifted i-atom) or, without a common image,
#       sub + mul + (rint) + fma 4; r^2 (mul + 2 fma) 5, 1/r = rsq + Newton (2 mul + 2 fma) 7, r 1                          = 16 + 3 nu
#       (round 3, unfused geometry: 19 + 2 nu)
#       inside the cutoff: LJ (add, 4 mul, 2 fma) 9;  erfc from the LDS table: alpha r / H 1, degree-5 interpolant 5 fma 10, q_j erfc / r
#       (mul + fma) 3 = 14;  with the field its derivative 4 fma 8, the field factor p - (x / H) p' (fma) 2, / r^3 3, q_j, q_i 2,
#       6 fma 12 = 27                                                                                                    = 23 / 50
#       (rounds 3-4 with the erfcx x G_k exp(t) table: 43 / 63)
#       Thole damping and (a, b) for the pairs of the stored tile pairs: 1/r^3, 1/r^5 4, lambda r 1, exp 28, polynomials 11, a, b 3   = 47
#   reciprocal space: K N (6 + ~40) for the structure factors, the same again for the field
# The counts follow the ARITHMETIC THE KERNELS DO (checked against the PMC's FMA / MUL / ADD counts, profiles/*_pmc_stalls.txt): a kernel
# that reaches the same result with fewer operations gets FASTER and its `frac` goes DOWN.  The line therefore carries both: `frac` (this
# build's arithmetic) and `frac_with_round3_flop_count` (the yardstick the round-3 review quoted its targets in).
FLOP_JAC_STORED, FLOP_JAC_FAR, FLOP_JAC_PER_NU = 39.0, 52.0, 3.0
FLOP_SWEEP_BASE, FLOP_SWEEP_PER_NU, FLOP_SWEEP_CUTOFF, FLOP_SWEEP_CUTOFF_NO_FIELD, FLOP_SWEEP_STORE = 16.0, 3.0, 50.0, 23.0, 47.0
FLOP_R3 = {"jac_far": 55.0, "sweep_base": 19.0, "sweep_per_nu": 2.0, "sweep_cutoff": 63.0, "sweep_cutoff_no_field": 43.0}
FLOP_RECIP_PER_K_ATOM = 2 * 46.0


def build_case(natoms: int, workdir: str):
    """the config-4 box through the reference's own file formats (so every loader sees the same doubles)."""
    from mpmcxx_amd import gen_box
    from mpmcxx_amd import pqr

    if natoms == 10000:
        name = "ion10k_polar"
        inp, _ = gen_box.materialize(name, workdir)
    else:  # reduced sizes are for quick functional runs only (NOT a valid benchmark number)
        L = 86.0 * (natoms / 10000.0) ** (1.0 / 3.0)
        rows = gen_box.lattice_box(natoms, L, 13)
        gen_box.write_pqr(os.path.join(workdir, "box.pqr"), rows)
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
        # bead 0 through the reference's own file formats
        rows = gen_box.lattice_box(n, float(np.asarray(basis)[0][0]), 13)
        for r, (x, y, z) in zip(rows, atoms["pos"]):
            r.x, r.y, r.z = float(x), float(y), float(z)
        gen_box.write_pqr(os.path.join(workdir, "bead0.pqr"), rows)
        gen_box.write_input(os.path.join(workdir, "bead0.in"), "bead0.pqr", np.asarray(basis).tolist(), dict(gen_box.POLAR_OPTS))
        t0 = time.time()
        p = subprocess.run([harness, "bead0.in", "--time", "1"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        txt = p.stdout
        res = json.loads(txt[txt.rfind("\n{") + 1:])
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
        r = S.energy(want_atoms=False)  # ONE full evaluation (~10 s of one host core at 10 000 atoms): the bounded CPU sample of the default run
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


def other_configs(headline_beads, headline_value, local_rank: int, workdir: str):
    """BASELINE configs[1..3] on this GPU, measured right behind the timed region of the headline: one system
    evaluated back to back ("alone": what System::mc sees, MonteCarlo.cpp:47) and 32 copies with jittered positions in flight
    (energy.pi_potential_local), plus the fp64 roofline entry of the configuration's pair kernel (back-to-back launches between one pair
    of HIP events on the kernel's stream).  A couple of seconds in all; tools/config_rates.py is the same loop with more repetitions."""
    from mpmcxx_amd import energy, gen_box, pqr

    out = []
    for label, name in (("configs[1]: 1 000-atom LJ box (rd_only), 1 GPU", "lj1000"), ("configs[2]: 10 000-atom LJ + Ewald box (kmax 7), 1 GPU", "ion10k_es"),
                        ("configs[3]: 10 000-atom LJ + Ewald + Thole box (10 Jacobi iterations), 1 GPU", "ion10k_polar")):
        inp, _ = gen_box.materialize(name, workdir)
        atoms, basis, opts = pqr.load_case(inp)
        S = energy.System(atoms, basis, opts, device=local_rank)
        e = S.energy()
        S.energy()
        reps = 400 if name == "lj1000" else (60 if name == "ion10k_es" else 15)
        t0 = time.perf_counter()
        for _ in range(reps):
            S.energy()
        alone = (time.perf_counter() - t0) / reps
        r = S.observables
        entry = {"workload": label, "energy_K": e, "evals_per_s_alone": 1.0 / alone, "ms_per_eval_alone": alone * 1e3, "reps_alone": reps}
        roof = None
        if name == "lj1000":
            # the whole evaluation is ONE launch (k_pair_fused<TAIL>: no classes, every pair takes the full minimum image); its duration is
            # the evaluation's wall time -- dispatch, 64-step latency chain and the polled result included: a latency regime, not a roofline one
            npairs = atoms["pos"].shape[0] * (atoms["pos"].shape[0] - 1) // 2
            fl = (FLOP_SWEEP_BASE + 3 * FLOP_SWEEP_PER_NU) * npairs + 10.0 * float(r["n_lj_in_cutoff"])
            roof = {"kernel": "k_pair_fused<TAIL> (single launch)", "algorithmic_flops": fl, "avg_launch_ms": alone * 1e3,
                    "clock": "wall time of the evaluation (one launch + dispatch + the host's poll of the posted result)", "pairs": npairs,
                    "pairs_in_cutoff": int(r["n_lj_in_cutoff"])}
        else:
            ps = S.pair_stats()
            try:
                ms = S.time_kernel("pair", 30)
                cut = float(r["n_es_in_cutoff"])
                polar = name == "ion10k_polar"
                fl = (FLOP_SWEEP_BASE * ps["pairs_swept"] + FLOP_SWEEP_PER_NU * ps["nonuniform_dims_x_pairs_swept"]
                      + (FLOP_SWEEP_CUTOFF if polar else FLOP_SWEEP_CUTOFF_NO_FIELD) * cut + (FLOP_SWEEP_STORE * ps["pairs_stored"] if polar else 0.0))
                fl_r3 = (FLOP_R3["sweep_base"] * ps["pairs_swept"] + FLOP_R3["sweep_per_nu"] * ps["nonuniform_dims_x_pairs_swept"]
                         + (FLOP_R3["sweep_cutoff"] if polar else FLOP_R3["sweep_cutoff_no_field"]) * cut + (FLOP_SWEEP_STORE * ps["pairs_stored"] if polar else 0.0))
                roof = {"kernel": "k_pair_sweep" if S.last_pair_kernel() == "sweep" else "k_pair_fused", "algorithmic_flops": fl, "flops_by_round3_count": fl_r3, "avg_launch_ms": ms,
                        "clock": "30 launches back to back between ONE pair of HIP events on the kernel's stream (kernel alone on the GPU)",
                        "pairs": ps["pairs"], "pairs_in_cutoff": int(cut)}
            except energy.MpmcError:
                roof = None
        if roof:
            ach = roof["algorithmic_flops"] / (roof["avg_launch_ms"] * 1e-3) / 1e12
            roof.update({"bound": "fp64_valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS})
            if roof.get("flops_by_round3_count"):
                roof["frac_with_round3_flop_count"] = roof["flops_by_round3_count"] / (roof["avg_launch_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
        entry["roofline"] = roof
        S.close()
        if name == "ion10k_polar" and headline_beads >= 32 and headline_value:
            entry["evals_per_s_in_flight"] = headline_value  # the headline IS this configuration with 32 beads in flight
            entry["in_flight"] = "the headline value of this line (32 bead-displaced copies of this box)"
        else:
            copies = []
            for b in range(32):
                a = dict(atoms)
                a["pos"] = atoms["pos"] + np.random.default_rng([17, b]).normal(scale=0.05, size=atoms["pos"].shape)
                copies.append(energy.System(a, basis, opts, device=local_rank))
            energy.pi_potential_local(copies)
            steps = 20 if name == "lj1000" else (6 if name == "ion10k_es" else 2)
            t0 = time.perf_counter()
            for _ in range(steps):
                energy.pi_potential_local(copies)
            many = (time.perf_counter() - t0) / (steps * 32)
            for c in copies:
                c.close()
            entry["evals_per_s_in_flight"] = 1.0 / many
            entry["in_flight"] = f"32 copies with jittered positions, all enqueued before the first wait ({steps} steps)"
        out.append(entry)
    return out


def self_launch(args) -> int:
    """`python3 bench.py --gpus N` started bare: N fresh ranks as CHILD processes of this one (which has not imported torch and has not
    touched HIP, and never replaces itself).  The ranks inherit stdout, so rank 0's JSON line is this command's JSON line; the exit
    code is the launcher's (non-zero if any rank failed).  The reference is started as one command too (`mpmcxx -P n input.in`,
    src/args_etc.h:216-293)."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC between the ranks of one node (see main())
    env["MPMC_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench.py] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def join_cabi_communicator(dist, torch, world, rank, dev, ready_here, make_unique_id, make_comm, timeout_s, comm_error):
    """The RCCL communicator of the C ABI for a job of `world` ranks, or None on EVERY rank.  Returns (comm, a_thread_is_stuck, why_not).

    Rank 0 makes the RCCL unique id, the launcher's channel (torch.distributed) carries its 128 bytes, every rank joins.  Three votes keep
    the ranks together: (1) everything a rank does on its own first (device index valid, RCCL symbols resolved) -- a rank that failed there
    alone would leave the others inside the collective initialisation; (2) ncclCommInitRank and a probe all-gather BLOCK until every rank
    is in them, and on a node where that never happens the job would hang: they run on a helper thread, and a rank that is not through after
    `timeout_s` votes no; (3) the outcome is a MIN over ranks, so all ranks use the communicator or none does (the combine then goes
    through torch.distributed).  `make_unique_id()`, `make_comm(uid)` and `comm_error` are parameters so that the votes can be exercised
    without RCCL (tests/test_bench_launch.py: two gloo ranks, one of which hangs)."""
    import threading

    import numpy as np

    def vote(value, op):
        t = torch.tensor([int(value)], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=op)
        return int(t.item())

    uid = [None]
    if vote(1 if ready_here else 0, dist.ReduceOp.MIN) == 1:
        if rank == 0:
            try:
                uid = [make_unique_id()]
            except comm_error:
                uid = [None]
        dist.broadcast_object_list(uid, src=0)
    comm, stuck, ok, why = None, False, 1, ""
    if uid[0] is None:
        ok, why = 0, "RCCL could not be opened below Python on every rank"
    else:
        box = {}

        def join_ranks():
            try:
                c = make_comm(uid[0])
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
            print(f"[rank {rank}] the C-ABI communicator did not come up within {timeout_s:.0f} s; voting for torch.distributed", file=sys.stderr)
        elif "err" in box:
            ok = 0
            print(f"[rank {rank}] mpmc_comm_init_rank failed ({box['err']}); falling back to torch.distributed", file=sys.stderr)
        else:
            comm = box["comm"]
            if not box["probe_ok"]:
                ok = 0
                print(f"[rank {rank}] the probe all-gather over the C-ABI communicator returned the wrong ranks; falling back to torch.distributed", file=sys.stderr)
    all_ok = vote(ok, dist.ReduceOp.MIN) == 1
    any_stuck = vote(1 if stuck else 0, dist.ReduceOp.MAX) == 1
    if not all_ok:
        if comm is not None and not any_stuck:
            comm.close()  # (with a rank still inside the collective initialisation the communicator is left alone: destroying it can block too)
        comm = None
        if any_stuck:
            why = f"the C-ABI communicator timed out after {timeout_s:.0f} s on some rank"
        elif not why:
            why = "mpmc_comm_init_rank or its probe all-gather failed on some rank"
    return comm, any_stuck, why


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
    ap.add_argument("--combine", choices=["gather", "reduce"], default="gather")
    ap.add_argument("--combine-impl", choices=["cabi", "torch"], default="cabi",
                    help="cabi: ncclAllGather inside libmpmc_energy.so (mpmc_pi_gather_beads; falls back to torch if RCCL cannot be initialised "
                         "below Python, recorded in config.combine_impl); torch: torch.distributed")
    ap.add_argument("--comm-init-timeout", type=float, default=120.0,
                    help="seconds a rank waits for the C-ABI communicator (ncclCommInitRank + one probe all-gather, both blocking collectives) before "
                         "the job falls back to torch.distributed for the combine instead of hanging")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="diagnostic (rounds 1-3 behaviour): per-launch HIP events on one bead's stream INSIDE the timed region; by default the timed "
                         "region carries no instrumentation and the in-flight kernel durations come from a separate short pass behind it")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs[1..3] pass (other_configs)")
    ap.add_argument("--no-extra-passes", action="store_true", help="diagnostic: skip the isolated-kernel and PCIe-inclusive passes after the timed region")
    ap.add_argument("--host-positions", action="store_true",
                    help="re-upload every bead's positions from host buffers inside each timed step (the PCIe-inclusive rate; the default run "
                         "measures it in a short extra pass and reports it as pcie_inclusive_value, never as value)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the multi-rank path on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this device")
    ap.add_argument("--beads-per-gpu-rehearsal", type=int, default=0,
                    help="one GPU, N beads in flight on it -- the per-GPU load of an (--beads / N)-GPU run, everything else as in that run: the "
                         "rate a GPU of the multi-GPU job can reach before the 4-double collective (N = 4: the 8-GPU case); NOT the headline")
    ap.add_argument("--configure", action="append", default=[], metavar="KEY=VALUE",
                    help="measurement switch for every context of this run (mpmc_debug_configure, e.g. side_stream=0 pair_kernel=1); repeatable")
    args = ap.parse_args()

    inprocess = args.launch == "inprocess" and args.gpus > 1
    if args.gpus > 1 and not inprocess and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # (before torch / HIP: the parent only starts the ranks and relays their exit code)

    # RCCL and CUDA-tensor sharing between the processes of one node go through dmabuf IPC on this pool: the host driver does not support
    # the legacy IPC mode, and without this variable the first multi-process collective fails with "hipIpcGetMemHandle: invalid argument".
    # It must be in the environment before the HIP runtime starts, i.e. before torch is imported (a no-op for one rank).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from mpmcxx_amd import energy, pi

    for kv in args.configure:
        k_, _, v_ = kv.partition("=")
        energy.configure(k_, float(v_))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if inprocess:
        if world != 1:
            raise SystemExit("--launch inprocess is ONE process: start it without a launcher")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the energy path has no CPU fallback)")
    if args.force_device is not None:
        local_rank = args.force_device
    # devices this process drives: its own (one rank per GPU) or all of them (--launch inprocess: bead b on devices[b mod N])
    n_dev = args.gpus if inprocess else 1
    devices = [local_rank] * n_dev if (not inprocess or args.force_device is not None) else list(range(n_dev))
    if inprocess and max(devices) >= energy.device_count():
        raise SystemExit(f"--launch inprocess --gpus {args.gpus}: only {energy.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- the collective of the combine ------------------------------------------------------------------------------------
    comm = None
    combine_impl = "none (one rank)" if world == 1 else "torch.distributed"
    rccl_ver = None
    try:
        rccl_ver = energy.rccl_version()
    except energy.MpmcError:
        pass
    state_comm_stuck = False
    if world > 1 and args.combine_impl == "cabi" and args.dist_backend == "nccl" and args.combine == "gather":
        ready_here = bool(rccl_ver) and 0 <= local_rank < energy.device_count()
        comm, state_comm_stuck, why = join_cabi_communicator(dist, torch, world, rank, dev, ready_here, energy.Comm.unique_id,
                                                             lambda uid: energy.Comm(world, rank, uid, local_rank), args.comm_init_timeout,
                                                             energy.MpmcError)
        if comm is not None:
            combine_impl = "libmpmc_energy.so: mpmc_pi_gather_beads (ncclAllGather, communicator from mpmc_comm_init_rank)"
        elif why:
            combine_impl = f"torch.distributed ({why})"

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

    host_pos = [np.ascontiguousarray(bead_positions(atoms["pos"], b)) for b in mine]
    state = {"host_positions": bool(args.host_positions), "per": None}

    def local_eval():
        # host_positions: the boundary as a host program with its own coordinates uses it -- every bead's positions arrive in host memory
        # inside the call (mpmc_pi_potential_local_host: bead b's upload is followed at once by its enqueue)
        hp = host_pos if state["host_positions"] else None
        if args.concurrency == "async":
            _, per, failed = energy.pi_potential_local(beads, host_positions=hp)
            state["per"] = per
            return per.table4()  # (the four combined terms straight from the library's result block: no per-bead dicts inside the step)
        else:
            per = []
            for k, s in enumerate(beads):
                if hp is not None:
                    s.update_positions(0, hp[k])
                s.energy()
                per.append(s.observables)
        state["per"] = per
        return np.array([[p["rd_energy"], p["coulombic_energy"], p["polarization_energy"], p["vdw_energy"]] for p in per])

    coll_dev = dev if args.dist_backend == "nccl" else "cpu"

    if inprocess:
        if args.host_positions or args.concurrency != "async":
            raise SystemExit("--launch inprocess runs the default step only (resident positions, all beads enqueued before the first wait)")
        combine_impl = "libmpmc_energy.so: mpmc_pi_allreduce (one host thread per device, ncclAllGather on an ncclCommInitAll communicator)"

        def step():
            # SimulationControl::PI_calculate_potential in ONE call of the C ABI: evaluation on every device, per-bead values gathered over
            # RCCL, the reference's ordered sum s = 0..P-1 (PathIntegral.cpp:786-801), then / P (mpmc_pi_finish)
            sums, per, failed = energy.pi_allreduce(beads)
            state["per"] = per
            return energy.pi_finish(sums, P)
    else:
        def step():
            return pi.pi_calculate_potential(local_eval, P, rank, world, mode=args.combine, device=coll_dev, comm=comm)

    def fence():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            vv, oo = step()
        fence()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
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
    # The timed region carries NO instrumentation (round 4): per-launch HIP events cost the stream that carries them about 5 us of
    # back-to-back dispatch per launch, and with few beads in flight the instrumented bead is the tail of every step.
    events_in_region = bool(args.events_in_timed_region)
    for k, s in enumerate(beads):
        s.set_profiling(events_in_region and k == 0)
        s.timings(reset=True)
    dt, v, obs = timed(args.steps)
    gpu_bead0 = dict(state["per"][0]) if (rank == 0 and state["per"]) else None
    if not events_in_region and not args.no_extra_passes:
        # what the kernels look like with the other beads in flight: a SEPARATE short pass behind the timed region, events on bead 0's stream
        beads[0].set_profiling(True)
        for _ in range(2):
            step()
        fence()
    agg = collect(beads)  # per-kernel device time from HIP events with all local beads in flight (instrumented bead)
    for s in beads:
        s.set_profiling(False)
    iters = int(beads[0].observables.get("polar_iterations", 0)) if beads else 0
    mem_total, mem_tensor = beads[0].memory_usage() if beads else (0, 0)
    tiles = beads[0].tile_stats() if beads else {"tile_pairs": 0, "thole_stored": 0, "thole_far": 0, "beyond_cutoff": 0}

    # ---- PCIe-inclusive rate: the same step with every bead's positions handed over in host memory (short extra pass, all ranks) ----
    pcie = None
    if not args.no_extra_passes and not args.host_positions and n_gpus == 1:  # (multi-rank runs report the headline only: no extra collectives)
        state["host_positions"] = True
        step()
        k_pcie = max(2, min(args.steps, 5))
        d2, _, _ = timed(k_pcie)
        state["host_positions"] = False
        pcie = {"value": P * k_pcie / d2, "steps": k_pcie,
                "what": "the same step with every bead's 10 000 positions handed over in host memory inside every timed step "
                        "(mpmc_pi_potential_local_host: one 320 KB upload per bead from the context's pinned mirror, bead b's upload followed at "
                        "once by its enqueue) -- the boundary as a host program that owns the coordinates uses it"}

    # ---- the kernels with NOTHING else on the GPU: untimed pass, one bead on one stream (HIP events on that stream) -------------
    iso = None
    back_to_back, back_to_back_runs = {}, {}  # kernel class -> ms per launch, launches back to back between ONE pair of HIP events
    pairs = beads[0].pair_stats() if beads else {}
    if rank == 0 and not args.no_extra_passes and n_gpus == 1:  # (with several ranks nobody is kept waiting in a barrier: in-region durations only)
        energy.configure("side_stream", 0)  # one stream: every kernel alone on the GPU
        try:
            a = dict(atoms)
            a["pos"] = bead_positions(atoms["pos"], mine[0])
            S1 = energy.System(a, basis, opts, device=local_rank)
        finally:
            energy.configure("side_stream", -1)
        S1.energy()  # warm-up (uploads, buffers)
        S1.set_profiling(True)
        for _ in range(3):
            S1.energy()
        iso = collect([S1])
        S1.set_profiling(False)
        S1.energy()
        back_to_back_runs = {}
        for which, key in (("panel", "dipole_iter"), ("pair", "pair")):
            try:  # three batches, the median counts (the first batch behind an idle stretch runs on ramping clocks)
                runs = sorted(S1.time_kernel(which, 100 if which == "panel" else 40) for _ in range(3))
                back_to_back[key] = runs[1]
                back_to_back_runs[key] = runs
            except energy.MpmcError:
                pass  # (dense / matrix-free solver: no panel kernel)
        pair_kernel_name = "k_pair_sweep" if S1.last_pair_kernel() == "sweep" else "k_pair_fused"
        S1.close()
    else:
        pair_kernel_name = "k_pair_sweep"
    if world > 1:
        dist.barrier()

    others = None
    if rank == 0 and n_gpus == 1 and not args.no_extra_passes and not args.no_other_configs and not rehearsal and args.natoms == 10000:
        others = other_configs(P, P * args.steps / dt, local_rank, workdir)

    try:  # per-launch PMC figures of the committed profiling passes (rocprofv3 cannot run inside this process): HBM bytes, executed flops
        pmc = {}
        for rnd in ("r01", "r02", "r03", "r04"):
            pth = os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")
            if os.path.exists(pth):
                with open(pth) as f:
                    for kname, rec in json.load(f).items():
                        pmc[kname] = dict(rec, source=f"profiles/{rnd}_traffic.json (committed rocprofv3 --pmc passes of the builder, NOT counters of this run)")
    except (OSError, ValueError):
        pmc = {}
    K = 709 if int(opts.get("ewald_kmax", 7)) == 7 else None
    cut_frac = 0.0
    if gpu_bead0 and pairs.get("pairs"):
        cut_frac = float(gpu_bead0["n_es_in_cutoff"]) / pairs["pairs"]
    flops_jacobi = (FLOP_JAC_STORED * pairs.get("pairs_stored", 0) + FLOP_JAC_FAR * pairs.get("pairs_far", 0)
                    + FLOP_JAC_PER_NU * (pairs.get("nonuniform_dims_x_pairs_stored", 0) + pairs.get("nonuniform_dims_x_pairs_far", 0)))
    flops_pair = (FLOP_SWEEP_BASE * pairs.get("pairs_swept", 0) + FLOP_SWEEP_PER_NU * pairs.get("nonuniform_dims_x_pairs_swept", 0)
                  + FLOP_SWEEP_CUTOFF * cut_frac * pairs.get("pairs", 0) + FLOP_SWEEP_STORE * pairs.get("pairs_stored", 0))
    flops_eval = flops_pair + iters * flops_jacobi + (FLOP_RECIP_PER_K_ATOM * K * n if K else 0.0)
    flops_r3 = {"dipole_iter": flops_jacobi + (FLOP_R3["jac_far"] - FLOP_JAC_FAR) * pairs.get("pairs_far", 0),
                "pair": ((FLOP_R3["sweep_base"] * pairs.get("pairs_swept", 0) + FLOP_R3["sweep_per_nu"] * pairs.get("nonuniform_dims_x_pairs_swept", 0)
                          + FLOP_R3["sweep_cutoff"] * cut_frac * pairs.get("pairs", 0) + FLOP_SWEEP_STORE * pairs.get("pairs_stored", 0)))}
    bytes_jacobi = 16.0 * 4096 * pairs.get("tile_pairs_stored", 0) + n * 80.0

    if rank == 0:
        evals = P * args.steps
        value = evals / dt
        ms_per_step = dt / args.steps * 1e3
        n_local = len(beads)
        solver_used = "dense" if args.solver == "dense" else ("compact" if mem_tensor > 0 else "matrix_free")
        in_region = {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in agg.items() if tv["launches"]}
        alone = {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in (iso or {}).items() if tv["launches"]}

        def compute_entry(kernel, cls_key, src, flops, launches_per_step):
            """fp64 vector-issue roofline of one kernel.  avg_launch_ms: HIP events on the kernel's stream (alone on the GPU where the
            isolated pass ran); achieved = algorithmic flops / that duration."""
            ms = back_to_back.get(cls_key) or src.get(cls_key) or 1e30
            ach = flops / (ms * 1e-3) / 1e12
            e = {"bound": "fp64_valu", "kernel": kernel, "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": ach / FP64_VALU_PEAK_TFLOPS, "frac_algorithmic": ach / FP64_VALU_PEAK_TFLOPS, "frac_executed": None, "traffic": None,
                 "avg_launch_ms": ms, "launches_per_step": launches_per_step, "algorithmic_flops_per_launch": flops,
                 "consistent": bool(ms * launches_per_step <= ms_per_step),
                 "clock": ("median of three batches of 100 (pair sweep: 40) launches back to back between ONE pair of HIP events on the kernel's stream, "
                           "per launch (kernel alone on the GPU)" if back_to_back.get(cls_key) else "HIP events around every launch on the kernel's stream")}
            if back_to_back.get(cls_key):
                e["avg_launch_ms_batches"] = back_to_back_runs.get(cls_key)
            if flops_r3.get(cls_key):  # the same duration priced with the operation counts of the round-3 kernels (what the review's targets were quoted in)
                e["frac_with_round3_flop_count"] = flops_r3[cls_key] / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
            # context, not the judged fraction: against the FMA rate the chip sustains (clock under fp64 load), and the ceiling of THIS instruction
            # stream at that rate -- algorithmic flops per issued VALU instruction (PMC) x the sustained issue rate
            e["frac_of_sustained_fma_rate"] = ach / FP64_FMA_SUSTAINED_TFLOPS
            if src.get(cls_key) and back_to_back.get(cls_key):
                e["avg_launch_ms_event_pair_per_launch"] = src[cls_key]
            t = pmc.get(kernel)
            if t and t.get("natoms") == n:
                e["traffic"] = t.get("hbm_bytes_per_launch")
                e["traffic_source"] = t.get("source")
                e["replayed_from_committed_pmc_pass"] = ["traffic", "pmc", "frac_by_trace_clock", "frac_executed", "algorithmic_flops_per_valu_slot",
                                                         "ceiling_frac_of_this_instruction_stream"]
                e["pmc"] = t
                if t.get("trace_avg_launch_ms"):
                    e["frac_by_trace_clock"] = flops / (t["trace_avg_launch_ms"] * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
                if t.get("executed_flops_per_launch"):
                    e["frac_executed"] = t["executed_flops_per_launch"] / (ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
                if t.get("valu_wave_insts_per_launch"):
                    # every VALU instruction an FMA on 64 lanes would be 128 flops: the share of that which is algorithmic work
                    e["algorithmic_flops_per_valu_slot"] = flops / (128.0 * t["valu_wave_insts_per_launch"])
                    e["ceiling_frac_of_this_instruction_stream"] = e["algorithmic_flops_per_valu_slot"] * FP64_FMA_SUSTAINED_TFLOPS / FP64_VALU_PEAK_TFLOPS
            return e

        # dominant kernel: the one with the largest share of the device time of an evaluation (alone-on-the-GPU durations)
        src = alone if alone else in_region
        share = {"dipole_iter": (src.get("dipole_iter") or 0.0) * iters, "pair": src.get("pair") or 0.0}
        dom = max(share, key=share.get) if any(share.values()) else "dipole_iter"
        jac_kernel = "k_dense_matvec" if solver_used == "dense" else ("k_dipole_iter_panel" if solver_used == "compact" else "k_dipole_iter_hybrid")
        if solver_used == "dense":  # the reference's 3N x 3N layout, contraction on v_mfma_f64_16x16x4_f64: HBM-bound
            n3 = 3 * ((n + 63) // 64 * 64)
            ntl = n3 // 192
            ms = src.get("dipole_iter") or 1e30
            whole = 8.0 * n3 * n3
            # round 4: A is symmetric (thole_amatrix :2748-2757) and the contraction reads its upper BLOCK triangle only -- tile pairs I <= J of
            # 192 x 192 doubles -- forming both products per block: those are the algorithmic bytes of a symmetric matrix-vector product
            alg = 8.0 * 192 * 192 * (ntl * (ntl + 1) // 2)
            ach = alg / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_dense_symv", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": ms, "launches_per_step": iters * n_local, "algorithmic_bytes_per_launch": alg,
                    "whole_matrix_bytes": whole, "whole_matrix_equivalent_GBs": whole / (ms * 1e-3) / 1e9,
                    "consistent": bool(ms * iters * n_local <= ms_per_step),
                    "mfma_side": {"issued_tflops": 2.0 * 16 * 192 * 192 * (ntl * (ntl - 1) + ntl) / (ms * 1e-3) / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                                  "useful_fraction": 1.0 / 16.0,
                                  "what": "v_mfma_f64_16x16x4_f64 with the vector replicated over one operand: two products per off-diagonal block, one per diagonal block"}}
        elif dom == "pair":
            roof = compute_entry(pair_kernel_name, "pair", src, flops_pair, n_local)
        else:
            roof = compute_entry(jac_kernel, "dipole_iter", src, flops_jacobi, iters * n_local)
            ms = roof["avg_launch_ms"]
            roof["hbm_side"] = {"achieved": bytes_jacobi / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": bytes_jacobi / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_jacobi}
        roof["measured"] = ("extra pass right after the timed region: ONE bead on ONE stream, every kernel alone on the GPU.  rocprofv3 --kernel-trace "
                            "reports less for the same launch (profiles/*_serial_kernel_stats.csv; frac_by_trace_clock): its begin / end stamps leave "
                            "out the dispatch and completion time between consecutive kernels of a stream, which every wall-clock measure includes"
                            if alone else "HIP events on one bead's stream over the timed region (beads overlap: stretched durations)")
        if not roof["consistent"]:  # cannot happen while other kernels share the step; if it does, the whole-step figure is the honest one
            roof["note_inconsistent"] = "avg_launch_ms x launches_per_step exceeds ms_per_step: use whole_step"
        roof["tile_pairs"] = tiles
        roof["pairs"] = dict(pairs, in_cutoff_fraction=cut_frac)
        roof["flop_model"] = {"jacobi_stored": FLOP_JAC_STORED, "jacobi_far": FLOP_JAC_FAR, "jacobi_per_nonuniform_dim": FLOP_JAC_PER_NU,
                              "sweep_base": FLOP_SWEEP_BASE, "sweep_per_nonuniform_dim": FLOP_SWEEP_PER_NU, "sweep_in_cutoff": FLOP_SWEEP_CUTOFF,
                              "sweep_in_cutoff_no_field": FLOP_SWEEP_CUTOFF_NO_FIELD, "round3_counts": FLOP_R3,
                              "sweep_stored": FLOP_SWEEP_STORE, "what": "fp64 flops per unordered atom pair, FMA = 2 (top of bench.py, DESIGN.md section 3); "
                                                                        "pair counts are exact atom pairs per tile-pair class (mpmc_debug_pair_stats)"}
        roof["share_of_device_time_alone"] = share[dom] / max(sum((src.get(k) or 0.0) * (iters if k in ("dipole_iter", "reduce") else 1) for k in src), 1e-30)
        # whole-step view, which overlap cannot distort: algorithmic flops of one evaluation x evaluations per second per GPU
        roof["whole_step"] = {"algorithmic_flops_per_eval": flops_eval, "achieved": flops_eval * value / n_gpus / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": flops_eval * value / n_gpus / 1e12 / FP64_VALU_PEAK_TFLOPS,
                              "what": "pair sweep + iterations x Jacobi contraction + reciprocal space, x evaluations/s per GPU"}
        # secondary: the same kernels inside the timed region (other beads' kernels share the GPU) and the other big kernel
        roof["in_timed_region"] = {"kernel_ms": in_region,
                                   "measured": ("INSIDE the timed region (--events-in-timed-region)" if events_in_region else
                                                "in a separate 2-step pass BEHIND the timed region (the timed region carries no events)"),
                                   "note": f"{args.concurrency}: {n_local} beads in flight on this GPU, HIP events on one bead's stream; a launch here "
                                           "shares the CUs with other beads' kernels, so its duration is stretched -- not a kernel time"}
        other = {}
        if alone.get("pair") and dom != "pair":
            other["pair"] = compute_entry(pair_kernel_name, "pair", src, flops_pair, n_local)
        if alone.get("dipole_iter") and dom == "pair" and solver_used != "dense":
            other["dipole_iter"] = compute_entry(jac_kernel, "dipole_iter", src, flops_jacobi, iters * n_local)
        roof["other_kernels"] = other
        roof["alone_kernel_ms"] = alone
        roof["note"] = ("fp64 vector-issue bound: the kernels issue v_fma_f64 / v_mul_f64 / v_add_f64 (MI355X fp64 vector and matrix peaks are both 78.6 "
                        "TFLOP/s).  One launch per Jacobi iteration over ALL tile pairs (panels of two tile pairs per workgroup of four waves); tensors "
                        "are stored (16 B per pair) for the tile pairs within lambda r = 30 only, the rest is recomputed from the positions.")

        out = {
            "metric": "energy-evals/sec (10k-atom LJ+Ewald+polar box); 1/2/4/8-GPU scaling",
            "value": value, "unit": "energy-evals/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{P}-bead path-integral ensemble of the {n}-atom polarizable box (BASELINE configs[3] x configs[4]): "
                                   f"LJ+LRC, Ewald kmax {opts['ewald_kmax']}, Thole exponential damping, {iters} Jacobi iterations, polar_ewald",
                       "natoms": n, "beads": P, "beads_per_gpu": P // n_gpus, "polar_solver": solver_used, "combine": args.combine,
                       "parallelism": f"beads sharded round-robin over {n_gpus} GPU(s); one {'all_gather' if args.combine == 'gather' else 'all_reduce'} of 4 fp64 per bead per step",
                       "dist_backend": (args.dist_backend if world > 1 else "none (one rank)"), "world_size": world, "combine_impl": combine_impl,
                       "launch": ("inprocess: one process, one host thread per device" if inprocess else
                                  ("ranks started by bench.py itself (child torch.distributed.run)" if os.environ.get("MPMC_BENCH_SELF_LAUNCHED") else
                                   ("ranks started by an external launcher" if world > 1 else "one process"))),
                       "rccl_version": rccl_ver, "configure": args.configure},
            "V_mean_K": v, "obs_rd_es_pol_vdw": [float(x) for x in obs],
            "kernel_ms": in_region,
            "device_bytes_per_bead": mem_total,
            "roofline": roof,
        }
        if rehearsal:
            out["config"]["rehearsal"] = (f"{rehearsal} beads in flight on ONE GPU: the per-GPU load of a {args.beads // rehearsal}-GPU run of the "
                                          f"{args.beads}-bead ensemble, without the 4-double collective; value x {args.beads // rehearsal} is what that "
                                          "run can reach at most.  NOT the headline workload")
        out["instrumented_in_timed_region"] = events_in_region
        if others is not None:
            out["other_configs"] = others
        if pcie is not None:
            out["pcie_inclusive_value"] = pcie["value"]
            out["pcie_inclusive"] = pcie
        if args.host_positions:
            out["note_host_positions"] = "positions of every bead handed over in host memory inside every timed step (PCIe-inclusive rate, not the headline)"
    # which device every rank drove (proof that RCCL saw N ranks on N devices): gathered from all ranks
    my_info = {"rank": rank, "local_rank": local_rank, "device": dev, "device_name": torch.cuda.get_device_name(local_rank), "pid": os.getpid(),
               "beads": mine, "comm_n_ranks": (comm.n_ranks if comm is not None else None)}
    infos = [my_info]
    if inprocess:  # one process: one entry per device it drove; the communicator is the library's own (ncclCommInitAll inside mpmc_pi_allreduce)
        n_devs_seen, comm_size = energy.pi_allreduce_info(beads)
        infos = [{"rank": g, "local_rank": d, "device": f"cuda:{d}", "device_name": torch.cuda.get_device_name(d), "pid": os.getpid(),
                  "beads": [b for k, b in enumerate(mine) if k % n_dev == g], "comm_n_ranks": comm_size, "distinct_devices": n_devs_seen}
                 for g, d in enumerate(devices)]
    if world > 1:
        infos = [None] * world
        dist.all_gather_object(infos, my_info)
    if rank == 0:
        out["config"]["ranks"] = infos
        cpu = None
        if n_gpus == 1 and args.cpu_baseline != "none":
            cpu = cpu_baseline(args.cpu_baseline, {**atoms, "pos": bead_positions(atoms["pos"], 0)}, basis, opts, workdir, gpu_bead0)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    for s in beads:
        s.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if state_comm_stuck:  # a helper thread is still inside a blocking RCCL call: leave without running its destructors
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
