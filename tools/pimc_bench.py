#!/usr/bin/env python3
"""Monte Carlo steps per second of the PI-NVT driver (examples/pimc_nvt.cpp) on the 10 000-atom LJ + Ewald box (BASELINE
configs[2] as a P-image path-integral system): full evaluations per move versus per-move delta energies (--trial).

usage: python tools/pimc_bench.py [P] [steps] [natoms] [polar]
"polar": the polarizable box (Thole iterative, 10 iterations) -- every trial is then a full evaluation of every image (a bead move
re-centres the whole chain, so no image keeps its coordinates).
Both runs use the same seed, so they make the same moves; their energy.dat rows must agree to 1e-9.
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mpmcxx_amd import build, gen_box  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
natoms = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
polar = len(sys.argv) > 4 and sys.argv[4] == "polar"
POLAR = "polarization on\npolar_damp_type exponential\npolar_damp 2.1304\npolar_iterative on\npolar_max_iter 10\npolar_ewald on\n"
wd = tempfile.mkdtemp(prefix="pimc_bench_")
L = 86.0 * (natoms / 10000.0) ** (1.0 / 3.0)
rows = gen_box.lattice_box(natoms, L, 13)
gen_box.write_pqr(os.path.join(wd, "box.pqr"), rows)
with open(os.path.join(wd, "pi.in"), "w") as f:
    f.write(f"""job_name pibox
ensemble pi_nvt
temperature 77.0
numsteps {steps}
corrtime {max(steps // 4, 1)}
seed 5
move_factor 0.002
rot_factor 1.0
bead_perturb_probability 0.5
PI_trial_chain_length 2
ewald_kmax 7
{POLAR if polar else ""}basis1 {L!r} 0.0 0.0
basis2 0.0 {L!r} 0.0
basis3 0.0 0.0 {L!r}
pqr_input box.pqr
""")
build.build_library()
exe = os.path.join(wd, "pimc_nvt")
libdir = os.path.join(ROOT, "mpmcxx_amd")
GPROF = os.environ.get("PIMC_BENCH_GPROF") == "1"  # host profile of the driver itself (library and HIP runtime time is not attributed)
OMP = os.environ.get("PIMC_BENCH_OPENMP") == "1"  # the images' enqueues by a few host threads (the facade's optional OpenMP loops)
subprocess.check_call(["g++", "-std=c++14", "-O2"] + (["-fopenmp"] if OMP else []) + (["-pg", "-fno-inline-small-functions", "-fno-inline-functions"] if GPROF else []) + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pimc_nvt.cpp"), "-L", libdir,
                       "-lmpmc_energy", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
res = {}
for mode in ("full", "trial"):
    out = os.path.join(wd, mode)
    os.makedirs(out)
    p = subprocess.run([exe, os.path.join(wd, "pi.in"), "-P", str(P), "-o", out] + (["--trial"] if mode == "trial" else []), stdout=subprocess.PIPE, text=True, check=True, cwd=out)
    if GPROF and os.path.exists(os.path.join(out, "gmon.out")):
        g = subprocess.run(["gprof", "-b", "-p", exe, os.path.join(out, "gmon.out")], stdout=subprocess.PIPE, text=True).stdout
        print(f"--- gprof flat profile, mode {mode} (first 14 rows)")
        print("\n".join(g.splitlines()[:19]))
    res[mode] = json.loads(p.stdout.strip().splitlines()[-1])
    res[mode]["rows"] = [ln.split() for ln in open(os.path.join(out, "pibox.energy.dat")) if not ln.startswith("#")]
worst = 0.0
for a, b in zip(res["full"]["rows"], res["trial"]["rows"]):
    for x, y in zip(a[1:7], b[1:7]):
        worst = max(worst, abs(float(x) - float(y)) / max(abs(float(y)), 1.0))
for mode in ("full", "trial"):
    r = res[mode]
    print(f"{mode:5s}: {natoms} atoms x {P} images, {r['steps']} steps in {r['seconds']:.2f} s = {r['steps_per_s']:.1f} MC steps/s "
          f"({r['energy_evals_per_s']:.0f} image evaluations/s), AR {r['AR']:.3f}; host waits: {r.get('wait_polls_seen')} polls seen, "
          f"{r.get('wait_polls_timed_out')} timed out, {r.get('wait_stream_syncs')} stream syncs, {r.get('wait_poll_yields')} yields; "
          f"{len(os.sched_getaffinity(0))} cores available, OpenMP {'on' if OMP else 'off'}")
print(f"largest relative difference between the two energy.dat files: {worst:.2e}; speed-up {res['trial']['steps_per_s'] / res['full']['steps_per_s']:.1f}x")
