#!/usr/bin/env python3
"""bench.py -- energy evaluations per second on the BASELINE.json headline workload.

Workload (BASELINE.json configs[3] sharded as configs[4]): a P = 32 bead path-integral ensemble of the 10 000-atom
polarizable box (LJ + LRC, Ewald real/reciprocal/self with kmax 7, Thole static field + 10 Jacobi dipole
iterations, polar_ewald on).  One "step" = one SimulationControl::PI_calculate_potential
(reference PathIntegral.cpp:752-805): a full stateless energy() of every bead + the 4-scalar combine.
Beads are sharded round-robin over the ranks (one process per GPU); the combine is ONE collective of 4 fp64 per
bead over torch.distributed (backend nccl = RCCL over xGMI).  Total work is fixed => "scaling": "strong".
At --gpus 1 all 32 beads run on the one GPU, i.e. 32 evaluations of the config-4 box per step.

value = (P * steps) / wall  [energy evaluations / s, whole job], inputs resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline     -- dominant kernel (the Thole dipole-iteration kernel, one launch per Jacobi iteration), timed with
                  HIP events on the stream it is launched on (mpmc_set_profiling / mpmc_get_timings).
  cpu_baseline -- ONE evaluation of the same 10k box on one core of this host: by the reference's own object code (kind "reference",
                  oracle/_ref/ref_harness, the default wherever that binary was built) or by the C port of the oracle (kind "port").
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
# Algorithmic fp64 flops per unordered pair, counted the way the 78.6 TFLOP/s peak is: one FMA = 2 flops.
#   Jacobi contraction, tensor (a, b) read from the store: d = r_i - r_j 3, minimum image 3 x (mul, rint, fma) 9 [rint not counted],
#     mu_j.d and mu_i.d 2 x (mul + 2 fma) 10, b x dot 2, E_i += a mu_j - (b mu_j.d) d and the same for E_j 2 x 6 fma 24         = 48
#   ... tensor recomputed (far field): + r^2 (mul + 2 fma) 5, 1/r = rsq + one Newton step 7, 1/r^3 and 3/r^5 4                 = 64
#   pair sweep: d + minimum image 12, r^2 5, 1/r 8 = 25 for every pair; inside the cutoff (52.3 % of the pairs of the benchmark box)
#     LJ 11, erfc(x) exp(-x^2) 85 (two Horner polynomials of 20 + 11 FMA, range reduction, one reciprocal), Coulomb 4, field factor and
#     both atoms 20 = 120; Thole damping + (a, b) 49 for the pairs of the stored tile pairs (36 %)          25 + 0.523 x 120 + 0.36 x 49 = 105
FLOP_PAIR_STORED, FLOP_PAIR_FAR, FLOP_PAIR_SWEEP = 48.0, 64.0, 105.0
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec)


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
    """bead b = base positions + Gaussian bead displacement (sigma 0.05 A), numpy default_rng(17) stream per bead."""
    rng = np.random.default_rng([17, bead])
    return pos + rng.normal(scale=0.05, size=pos.shape)


def cpu_baseline(kind: str, atoms, basis, opts, workdir: str):
    if kind == "none":
        return None
    if kind == "auto":
        kind = "reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")) else "port"
    n = atoms["pos"].shape[0]
    if kind == "reference":
        import subprocess

        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
        if not os.path.exists(harness):
            raise RuntimeError("oracle/_ref/ref_harness is not present (built only where /root/reference exists)")
        name = "ion10k_polar.in" if n == 10000 else "box.in"
        t0 = time.time()
        p = subprocess.run([harness, name, "--time", "1"], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        txt = p.stdout
        res = json.loads(txt[txt.rfind("\n{") + 1:])
        sec = res["time_mean_s"]
        return {"value": 1.0 / sec, "unit": "energy-evals/s", "cores": 1, "kind": "reference",
                "sample": f"1 steady-state full-recompute System::energy() of the same {n}-atom box by the reference's object code "
                          f"(oracle/_ref/ref_harness; {sec:.2f} s; harness wall incl. pair-list setup {time.time() - t0:.0f} s)",
                "energy": res["total"]}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the checker: this leg is the only place bench.py touches oracle/
    from oracle import OracleSystem

    S = OracleSystem(atoms, basis, opts)
    stride = 1  # every row: ONE full evaluation (~13 s of one host core at 10 000 atoms), the bounded CPU sample of the default run
    est, wall = S.time_sample(stride)
    sec = float(est[6])
    names = ["lj+lrc", "coulombic_real", "coulombic_reciprocal+self", "thole_amatrix", "thole_field", "thole_iterative"]
    return {"value": 1.0 / sec, "unit": "energy-evals/s", "cores": 1, "kind": "port",
            "sample": f"scalar C oracle (oracle/mpmc_oracle.c, dense-A algorithm of the reference) on the same {n}-atom box: every O(N^2) stage "
                      f"of ONE evaluation, rows i = 0, {stride}, {2 * stride}, ... (1/{stride} of the pair work; reciprocal-space and O(N) "
                      f"stages in full), each stage scaled by its exact work ratio; {wall:.1f} s of CPU work -> {sec:.1f} s per full evaluation",
            "seconds_per_eval_by_stage": {k: round(float(v), 3) for k, v in zip(names, est[:6])}}


def main():
    ap = argparse.ArgumentParser()
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
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="diagnostic: leave the per-kernel HIP events off in the timed region (the roofline entry is then empty)")
    ap.add_argument("--host-positions", action="store_true",
                    help="also re-upload every bead's positions from host buffers inside each timed step (the PCIe-inclusive rate "
                         "noted in DESIGN.md §6; never the headline value)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI) on a multi-GPU node; gloo only to rehearse the multi-rank path on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=None, help="rehearsal only: put every rank on this device")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from mpmcxx_amd import energy, pi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the energy path has no CPU fallback)")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    P = args.beads
    if P % world:
        raise SystemExit("--beads must be a multiple of --gpus")
    workdir = tempfile.mkdtemp(prefix="mpmc_bench_")
    atoms, basis, opts = build_case(args.natoms, workdir)
    opts = dict(opts)
    opts["solver"] = args.solver
    n = atoms["pos"].shape[0]

    mine = pi.beads_of_rank(P, rank, world)
    beads = []
    for b in mine:
        a = dict(atoms)
        a["pos"] = bead_positions(atoms["pos"], b)
        beads.append(energy.System(a, basis, opts, device=local_rank))

    host_pos = [np.ascontiguousarray(bead_positions(atoms["pos"], b)) for b in mine] if args.host_positions else None

    def local_eval():
        if host_pos is not None:  # the boundary as the reference's adapter uses it: positions arrive in host memory every call
            for s, hp in zip(beads, host_pos):
                s.update_positions(0, hp)
        if args.concurrency == "async":
            _, per, failed = energy.pi_potential_local(beads)
        else:
            per = []
            for s in beads:
                s.energy()
                per.append(s.observables)
        return np.array([[p["rd_energy"], p["coulombic_energy"], p["polarization_energy"], p["vdw_energy"]] for p in per])

    coll_dev = dev if args.dist_backend == "nccl" else "cpu"

    def step():
        return pi.pi_calculate_potential(local_eval, P, rank, world, mode=args.combine, device=coll_dev)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        v, obs = step()
    # per-kernel HIP events on ONE bead's stream only: an event pair around every launch costs that stream about 5 us of back-to-back
    # dispatch (measured: 8 % of the whole-job rate when all 32 beads carry them, 14 % with one bead at a time), so the other
    # beads run uninstrumented and the instrumented one supplies the launch durations of the timed region
    for k, s in enumerate(beads):
        s.set_profiling((k == 0) and not args.no_kernel_timing)
        s.timings(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v, obs = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def collect():
        agg = {}
        for s in beads:
            for k, tv in s.timings(reset=True).items():
                a = agg.setdefault(k, {"ms": 0.0, "launches": 0})
                a["ms"] += tv["ms"]
                a["launches"] += tv["launches"]
        return agg

    # per-kernel device time from HIP events on each bead's stream, over the timed region
    agg = collect()
    iters = int(beads[0].observables.get("polar_iterations", 0)) if beads else 0
    mem_total, mem_tensor = beads[0].memory_usage() if beads else (0, 0)
    tiles = beads[0].tile_stats() if beads else {"tile_pairs": 0, "thole_stored": 0, "thole_far": 0, "beyond_cutoff": 0}

    # the same kernels with NOTHING else on the GPU: one extra, untimed pass, one bead at a time (HIP events again).
    # In the timed region up to 32 beads are in flight on 32 streams, so an event pair there brackets a kernel that shares
    # the GPU with other beads' kernels.
    iso = None
    if rank == 0:
        # two fresh contexts with the Jacobi contraction as two kernels (MPMC_JACOBI=split): k_dipole_iter_stream is the pure
        # HBM-streaming part, k_dipole_iter_far the pure fp64 part of the default single-launch kernel
        old = os.environ.get("MPMC_JACOBI")
        old1 = os.environ.get("MPMC_ONE_STREAM")
        os.environ["MPMC_JACOBI"] = "split"
        os.environ["MPMC_ONE_STREAM"] = "1"  # no side stream either: every kernel of these contexts runs alone
        try:
            iso_beads = []
            for b in mine[:2]:
                a = dict(atoms)
                a["pos"] = bead_positions(atoms["pos"], b)
                iso_beads.append(energy.System(a, basis, opts, device=local_rank))
        finally:
            for k_, v_ in (("MPMC_JACOBI", old), ("MPMC_ONE_STREAM", old1)):
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
        for s in iso_beads:
            s.energy()  # warm-up (uploads, buffers)
        for s in iso_beads:
            s.set_profiling(True)
        for _ in range(2):
            for s in iso_beads:
                s.energy()
        iso = {}
        for s in iso_beads:
            for k, tv in s.timings(reset=True).items():
                a = iso.setdefault(k, {"ms": 0.0, "launches": 0})
                a["ms"] += tv["ms"]
                a["launches"] += tv["launches"]
            s.close()
        # and the production (single-launch) contraction itself, alone on the GPU: one more context, default kernels, one stream
        old1 = os.environ.get("MPMC_ONE_STREAM")
        os.environ["MPMC_ONE_STREAM"] = "1"
        try:
            a1 = dict(atoms)
            a1["pos"] = bead_positions(atoms["pos"], mine[0])
            s1 = energy.System(a1, basis, opts, device=local_rank)
        finally:
            if old1 is None:
                os.environ.pop("MPMC_ONE_STREAM", None)
            else:
                os.environ["MPMC_ONE_STREAM"] = old1
        s1.energy()
        s1.set_profiling(True)
        for _ in range(2):
            s1.energy()
        iso_hybrid = s1.timings(reset=True)
        s1.close()
    if world > 1:
        dist.barrier()

    hybrid_default = not agg.get("dipole_far", {}).get("launches")
    try:  # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            pmc_traffic = json.load(f)
    except OSError:
        pmc_traffic = {}
    n_pairs_all = n * (n - 1) // 2
    n_pairs_stored = tiles["thole_stored"] * 4096
    n_pairs_far = tiles["thole_far"] * 4096

    # the Jacobi iterations of the beads of one rank run in lockstep: one launch carries `batch` systems (mpmc_last_batch_size)
    batch = beads[0].last_batch_size() if (beads and args.concurrency == "async") else 1

    def roofline_of(name, tv, label, hybrid=None, per_launch=1):
        hybrid = hybrid_default if hybrid is None else hybrid
        """roofline of one kernel class from its HIP-event time.  Algorithmic figures (DESIGN.md §3):
        dipole_iter (k_dipole_iter_stream): HBM -- 16 B per stored unordered pair + 80 B per atom (positions, dipoles in, field out)
        dipole_far  (k_dipole_iter_far)   : fp64 -- FLOP_PAIR_FAR per far-field pair
        pair        (k_pair_fused)        : fp64 -- FLOP_PAIR_SWEEP per pair (breakdowns next to the constants at the top of this file)"""
        avg_ms = tv["ms"] / max(tv["launches"], 1)
        sec = avg_ms * 1e-3
        if name == "dipole_iter" and args.solver == "dense":  # the reference's 3N x 3N layout, contraction on v_mfma_f64_16x16x4_f64
            n3 = 3 * ((n + 63) // 64 * 64)
            alg = 8.0 * n3 * n3
            ach = alg / sec / 1e9 if sec > 0 else 0.0
            issued = 2.0 * n3 * n3 * 16 / sec / 1e12 if sec > 0 else 0.0
            return {"bound": "hbm", "kernel": "k_dense_matvec", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "avg_launch_ms": avg_ms, "launches": tv["launches"], "algorithmic_bytes_per_launch": alg, "measured": label,
                    "mfma_side": {"issued_tflops": issued, "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": issued / FP64_VALU_PEAK_TFLOPS,
                                  "useful_fraction": 1.0 / 16.0},
                    "note": "dense (3N)^2 x 8 B matrix-vector product: the vector is replicated over the 16 rows of the MFMA's A operand, no operand reuse"}
        if name == "dipole_iter":
            alg = (16.0 * n_pairs_stored + n * 80.0) * per_launch
            ach = alg / sec / 1e9 if sec > 0 else 0.0
            hbm = {"bound": "hbm", "kernel": "k_dipole_iter_hybrid" if hybrid else "k_dipole_iter_stream", "achieved": ach, "peak": HBM_PEAK_GBS,
                   "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_ms, "launches": tv["launches"],
                   "algorithmic_bytes_per_launch": alg, "measured": label}
            tr = pmc_traffic.get(hbm["kernel"])
            if tr and tr.get("natoms") == n:
                hbm["traffic"] = tr["hbm_bytes_per_launch"]
                hbm["traffic_source"] = tr["source"]
            if not hybrid:
                return hbm
            # the single-launch form walks ALL tile pairs: the stored ones stream 16 B/pair, the far ones recompute the bare dipole
            # tensor.  fp64 issue is what binds it (measured on MI355X: sending every off-diagonal tile pair down the recompute path
            # leaves the launch time unchanged, dropping the HBM loads saves 8 %), so the compute roof is the primary
            # entry and the bytes are reported beside it.
            fl = (FLOP_PAIR_STORED * n_pairs_stored + FLOP_PAIR_FAR * n_pairs_far) * per_launch
            tf = fl / sec / 1e12 if sec > 0 else 0.0
            if hbm["traffic"] is not None:
                hbm["traffic"] *= per_launch
            out = {"bound": "mfma", "kernel": "k_dipole_iter_hybrid_b" if per_launch > 1 else "k_dipole_iter_hybrid", "systems_per_launch": per_launch, "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": tf / FP64_VALU_PEAK_TFLOPS, "traffic": hbm["traffic"], "avg_launch_ms": avg_ms, "launches": tv["launches"],
                   "algorithmic_flops_per_launch": fl, "measured": label,
                   "hbm_side": {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "traffic")},
                   "note": ("fp64 VALU bound (MI355X fp64 vector and matrix peaks are both 78.6 TFLOP/s; the kernel issues v_fma_f64).  In the default "
                            "(async) mode the beads run on independent streams and this launch shares the GPU with other beads' kernels, so its "
                            "duration in the timed region is about twice what it needs alone: `isolated` holds the same kernels one at a time, "
                            "MPMC_PI_LOCKSTEP=1 runs all beads' iterations in one launch per iteration (clean durations, 9 % lower whole-job rate).  "
                            "One launch per Jacobi "
                            "iteration over ALL tile pairs: 48 flop per streamed pair (16 B of stored tensor), 64 flop per recomputed far-field pair (FMA = 2).  "
                            "Sustained v_fma_f64 issue measured on this pool (tools/microbench_f64.hip): 59 TFLOP/s, and the kernel's instruction mix "
                            "(FMA 41 %, MUL 29 %, DPP/int 17 %, ADD/RNDNE 11 %, RSQ 2 %) runs at ~80 % of the rate that mix sustains.  "
                            "MPMC_JACOBI=split runs the two halves as separate kernels (k_dipole_iter_stream HBM-bound, k_dipole_iter_far fp64-bound).")}
            if hbm.get("traffic_source"):
                out["traffic_source"] = hbm["traffic_source"]
            return out
        flops = FLOP_PAIR_FAR * n_pairs_far if name == "dipole_far" else FLOP_PAIR_SWEEP * n_pairs_all
        ach = flops / sec / 1e12 if sec > 0 else 0.0
        return {"bound": "mfma", "kernel": "k_dipole_iter_far" if name == "dipole_far" else "k_pair_fused", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": avg_ms, "launches": tv["launches"],
                "algorithmic_flops_per_launch": flops, "measured": label,
                "note": "fp64 compute bound: MI355X fp64 matrix (v_mfma_f64) and vector peaks are both 78.6 TFLOP/s; the kernel issues v_fma_f64"}

    if rank == 0:
        evals = P * args.steps
        value = evals / dt
        solver_used = "dense" if args.solver == "dense" else ("compact" if mem_tensor > 0 else "matrix_free")
        label = f"HIP events on one bead's stream over the timed region ({args.concurrency}: {len(beads)} beads in flight on this GPU)"
        cand = [k for k in ("dipole_iter", "dipole_far", "pair") if agg.get(k, {}).get("launches")]
        dom = max(cand, key=lambda k: agg[k]["ms"]) if cand else "dipole_iter"
        roof = roofline_of(dom, agg.get(dom, {"ms": 0.0, "launches": 0}), label, per_launch=batch if dom == "dipole_iter" else 1)
        # classes launched once per bead are seen on the instrumented bead only, the lockstep launches cover all `batch` beads
        scaled = {k: v["ms"] * (1 if (batch > 1 and k == "dipole_iter") else max(batch, 1)) for k, v in agg.items()}
        roof["share_of_device_time"] = scaled[dom] / max(sum(scaled.values()), 1e-30) if cand else 0.0
        roof["tile_pairs"] = tiles
        roof["other_kernels"] = {k: roofline_of(k, agg[k], label, per_launch=batch if k == "dipole_iter" else 1) for k in cand if k != dom}
        if iso is not None:
            lab2 = ("HIP events, extra untimed pass after the timed region: one bead at a time (each kernel alone on the GPU), Jacobi contraction "
                    "as two kernels on one stream (MPMC_JACOBI=split MPMC_ONE_STREAM=1)")
            roof["isolated"] = {k: roofline_of(k, iso[k], lab2, hybrid=False) for k in ("dipole_iter", "dipole_far", "pair") if iso.get(k, {}).get("launches")}
            roof["isolated_kernel_ms"] = {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in iso.items() if tv["launches"]}
            # the production kernel with nothing else on the GPU: what it reaches when other beads' kernels do not stretch its duration
            ih = iso_hybrid.get("dipole_iter", {"ms": 0.0, "launches": 0})
            if ih["launches"] and not ("dipole_far" in iso_hybrid and iso_hybrid["dipole_far"]["launches"]):
                it_ms = ih["ms"] / ih["launches"]
                fl = FLOP_PAIR_STORED * n_pairs_stored + FLOP_PAIR_FAR * n_pairs_far
                roof["alone_on_the_gpu"] = {"kernel": "k_dipole_iter_hybrid", "avg_launch_ms": it_ms, "achieved": fl / (it_ms * 1e-3) / 1e12,
                                            "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (it_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                            "hbm_GBps": (16.0 * n_pairs_stored + n * 80.0) / (it_ms * 1e-3) / 1e9,
                                            "measured": "HIP events, extra untimed pass after the timed region, one bead, one stream"}
        cpu = cpu_baseline(args.cpu_baseline, {**atoms, "pos": bead_positions(atoms["pos"], 0)}, basis, opts, workdir) if world == 1 else None
        out = {
            "metric": "energy-evals/sec (10k-atom LJ+Ewald+polar box); 1/2/4/8-GPU scaling",
            "value": value, "unit": "energy-evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{P}-bead path-integral ensemble of the {n}-atom polarizable box (BASELINE configs[3] x configs[4]): "
                                   f"LJ+LRC, Ewald kmax {opts['ewald_kmax']}, Thole exponential damping, {iters} Jacobi iterations, polar_ewald",
                       "natoms": n, "beads": P, "beads_per_gpu": P // world, "polar_solver": solver_used, "combine": args.combine,
                       "parallelism": f"beads sharded round-robin over {world} GPU(s); one {'all_gather' if args.combine == 'gather' else 'all_reduce'} of 4 fp64 per bead per step"},
            "V_mean_K": v, "obs_rd_es_pol_vdw": [float(x) for x in obs],
            "kernel_ms": {k: round(tv["ms"] / max(tv["launches"], 1), 6) for k, tv in agg.items() if tv["launches"]},
            "device_bytes_per_bead": mem_total,
            "roofline": roof,
        }
        if args.host_positions:
            out["note_host_positions"] = "positions of every bead re-uploaded from host memory inside every timed step (PCIe-inclusive rate, not the headline)"
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    for s in beads:
        s.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
