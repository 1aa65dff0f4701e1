"""GPU (MI355X): DROP-IN PROOF.  The reference's own executable -- its Monte Carlo driver, move generation,
Metropolis test, averaging and file output, compiled in place into oracle/_ref/ -- runs with System::energy()
interposed (-Wl,--wrap) by oracle/ref_adapter.cpp, which calls the HIP path through the C ABI.
The run must reproduce the stock binary's output (goldens generated in the build container by the stock binary).

MPMC_WRAP_MODE=both additionally evaluates the original energy() on every call and aborts on a > 1e-9 relative
difference in any component, so every one of the thousands of configurations visited is a parity check."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

WRAPPED = os.path.join(util.ROOT, "oracle", "_ref", "mpmcxx_wrapped")


def run_case(tmp_path, case, infile, P, mode):
    if not os.path.exists(WRAPPED):
        pytest.skip("oracle/_ref/mpmcxx_wrapped not present (built only where /root/reference exists)")
    src = os.path.join(util.GOLDEN, case)
    for f in os.listdir(src):
        if not f.startswith("golden_"):
            shutil.copy(os.path.join(src, f), tmp_path)
    env = dict(os.environ, MPMC_WRAP_MODE=mode, OMP_NUM_THREADS=str(P))
    p = subprocess.run([WRAPPED, "-P", str(P), infile], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "calls served by libmpmc_energy.so" in p.stderr, p.stderr[-500:]
    return p


def rows(path):
    out = []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        out.append([float(x) for x in line.split()])
    return out


def compare_energy_dat(ours, gold, rel):
    a, b = rows(ours), rows(gold)
    assert len(a) == len(b)
    for ra, rb in zip(a, b):
        assert ra[0] == rb[0]
        for x, y in zip(ra[1:], rb[1:]):
            assert abs(x - y) <= rel * max(abs(y), 1.0) + 1.1e-6, (ra, rb)  # the file prints 6 decimals


@pytest.mark.parametrize("mode", ["gpu", "both"])
def test_pi001_argon_dimer_runs_on_the_hip_path(tmp_path, mode):
    """BASELINE config 1: sample-input/pi001-argon-dimer-2K (equilibrate.in, 2000 steps, -P 8, seed 1)."""
    p = run_case(tmp_path, "pi001", "equilibrate.in", 8, mode)
    compare_energy_dat(os.path.join(tmp_path, "ArAr2K.energy.dat"), os.path.join(util.GOLDEN, "pi001", "golden_energy.dat"), 1e-9)
    gold = open(os.path.join(util.GOLDEN, "pi001", "golden_final_averages.txt")).read().strip().splitlines()
    for line in gold[-4:]:  # final AR / total energy / kinetic energy lines, printed digits identical
        assert line in p.stdout, line
    n_calls = int([ln for ln in p.stderr.strip().splitlines() if "calls served" in ln][-1].split()[1])
    assert n_calls == 21000  # 8 initial + 2000*8 + 624 accepted*8


@pytest.mark.parametrize("mode", ["gpu", "both"])
def test_pi_polarizable_box_runs_on_the_hip_path(tmp_path, mode):
    """27 polarizable ions, P = 4 beads, 300 PI-NVT steps: LJ + Ewald + Thole through the stock PI driver."""
    p = run_case(tmp_path, "pi_ion27", "input.in", 4, mode)
    compare_energy_dat(os.path.join(tmp_path, "ion27.energy.dat"), os.path.join(util.GOLDEN, "pi_ion27", "golden_energy.dat"), 1e-9)
    gold = open(os.path.join(util.GOLDEN, "pi_ion27", "golden_final_averages.txt")).read().strip().splitlines()
    for line in gold[-4:]:
        assert line in p.stdout, line
    # per-molecule dipoles / fields written by the reference's own writers from atom->mu / ef_static / ef_induced
    for name in ("dipole", "field"):
        ours = open(os.path.join(tmp_path, f"ion27.{name}.dat")).read().split()
        ref = open(os.path.join(util.GOLDEN, "pi_ion27", f"golden_{name}.dat")).read().split()
        assert len(ours) == len(ref)
        for x, y in zip(ours, ref):
            assert abs(float(x) - float(y)) <= 1e-6 * max(abs(float(y)), 1.0) + 2e-6, (name, x, y)


def test_pi_1000_ion_box_stock_driver_checks_every_call(tmp_path):
    """1000 polarizable ions, P = 4, 12 PI-NVT steps of the STOCK driver with MPMC_WRAP_MODE=both: every energy() call is evaluated by the
    reference as well and the adapter aborts on any component differing by more than 1e-9."""
    p = run_case(tmp_path, "pi_ion1000", "input.in", 4, "both")
    compare_energy_dat(os.path.join(tmp_path, "ion1000.energy.dat"), os.path.join(util.GOLDEN, "pi_ion1000", "golden_energy.dat"), 1e-9)
    assert "MISMATCH" not in p.stderr


@pytest.mark.parametrize("mode", ["gpu", "both"])
@pytest.mark.parametrize("case,job", [("pi_h2", "h2pi"), ("pi_water64", "water64"), ("pi_frozen", "frozen"), ("pi_tri", "tri"), ("pi_gs", "gs"), ("pi_nopbc", "nopbc"), ("pi_wolf", "wolf"),
                                      ("pi_h2_orient", "h2or")])  # (orientational bead moves, per-image restart files)
def test_pi_boxes_through_the_stock_driver(tmp_path, case, job, mode):
    """Rigid molecules through the stock driver (rotation + translation moves, wrapall): 8 LJ diatomics, 64 three-site polarizable
    molecules with a neutral polarizable atom, and a frozen charged framework with mobile polar diatomics (intramolecular exclusions, the erf form of the field for chargeless partners, Ewald,
    Thole).  In mode `both` the reference evaluates every configuration too and the adapter aborts on a 1e-9 difference."""
    p = run_case(tmp_path, case, "input.in", 4, mode)
    compare_energy_dat(os.path.join(tmp_path, f"{job}.energy.dat"), os.path.join(util.GOLDEN, case, "golden_energy.dat"), 1e-9)
    assert "MISMATCH" not in p.stderr
    gold = open(os.path.join(util.GOLDEN, case, "golden_final_averages.txt")).read().strip().splitlines()
    for line in gold[-4:]:
        assert line in p.stdout, line


GIBBS_WRAPPED = os.path.join(util.ROOT, "oracle", "_ref", "ref_gibbs_traj_wrapped")


@pytest.mark.parametrize("case", ["gibbs_lj", "gibbs_water", "gibbs_water_polar"])
def test_gibbs_shaped_call_sequence_through_the_adapter(tmp_path, case):
    """The stock Gibbs_mc loop cannot run without MPI; its call sequence can: oracle/ref_gibbs_traj.cpp drives the reference's own
    pick_Gibbs_move / make_move_Gibbs / energy / boltzmann_factor_NVT_Gibbs / restore over TWO Systems (Gibbs.cpp:179-180), here linked
    with System::energy() interposed by the adapter.  Transfer moves change both boxes' atom counts between calls (the adapter sees another
    N and goes through mpmc_set_atoms, growing past the capacity hint), volume moves rescale both cells.  MPMC_WRAP_MODE=both: the reference
    evaluates every configuration as well and the adapter aborts on a 1e-9 difference in any component; the trajectory must be the one the
    unwrapped driver made in the build container (tests/golden/gibbs_*/trajectory.json: same decisions, energies to 1e-9)."""
    import json

    if not os.path.exists(GIBBS_WRAPPED):
        pytest.skip("oracle/_ref/ref_gibbs_traj_wrapped not present (built only where /root/reference exists)")
    src = os.path.join(util.GOLDEN, case)
    for f in ("input.in", "boxA.pqr", "boxB.pqr"):
        shutil.copy(os.path.join(src, f), tmp_path)
    with open(os.path.join(src, "trajectory.json")) as f:
        ref = json.load(f)
    steps = len(ref["steps"])
    p = subprocess.run([GIBBS_WRAPPED, "input.in", str(steps)], cwd=tmp_path, env=dict(os.environ, MPMC_WRAP_MODE="both", OMP_NUM_THREADS="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "MISMATCH" not in p.stderr and "calls served by libmpmc_energy.so" in p.stderr, p.stderr[-800:]
    ours = json.loads(p.stdout[p.stdout.rfind('\n{"initial') + 1:])
    assert len(ours["steps"]) == steps
    natoms_seen = set()
    for a, b in zip(ours["steps"], ref["steps"]):
        assert a["movetype"] == b["movetype"] and a["accepted"] == b["accepted"] and a["natoms"] == b["natoms"] and a["N"] == b["N"], (a["step"], a, b)
        for k in range(2):
            for key in ("final_energy", "energy", "volume"):
                x, y = a[key][k], b[key][k]
                assert (x == y) or (not np.isfinite(y) and not np.isfinite(x)) or abs(x - y) <= 1e-9 * max(abs(y), 1.0), (a["step"], key, x, y)
        natoms_seen.add(tuple(a["natoms"]))
    assert len(natoms_seen) > 1  # molecules did move between the boxes: mpmc_set_atoms saw several N per context
    n_calls = int([ln for ln in p.stderr.strip().splitlines() if "calls served" in ln][-1].split()[1])
    assert n_calls >= 2 * steps
