"""the path-integral NVT Monte Carlo driver (include/mpmc_pimc.hpp) against the stock reference binary.

The goldens (tests/golden/pi001 = the reference's sample-input/pi001-argon-dimer-2K, tests/golden/pi_ion27 = 27 polarizable
ions) were written by the UNMODIFIED reference executable (oracle/make_pi_golden.sh).  The driver uses the same random-number
stream and the same arithmetic, so a run must reproduce every printed digit of `energy.dat`, the acceptance rates and the final
bead geometries.
  * CPU: the driver with the oracle as the per-image evaluator (tests/cpp/pimc_check.cpp) -- pins the host logic without a GPU;
  * GPU: examples/pimc_nvt (the HIP path through the C++ facade) on the same inputs.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import util
from mpmcxx_amd import pqr

CASES = {
    # name: (input file, P, job name, expected acceptance: (AR, displace, bead))
    "pi001": ("equilibrate.in", 8, "ArAr2K"),
    "pi_ion27": ("input.in", 4, "ion27"),
    "pi_h2": ("input.in", 4, "h2pi"),  # 8 rigid diatomics (LJ sites): the quaternion rotation of PI_displace, rigid translation in the bead moves
    "pi_water64": ("input.in", 4, "water64"),  # 64 rigid 3-site polarizable molecules + a neutral atom: exclusions, rotation, Ewald, Thole together
    "pi_frozen": ("input.in", 4, "frozen"),  # frozen charged framework (27 sites) + 6 mobile polar diatomics: frozen pairs, only movable molecules are picked
    "pi_tri": ("input.in", 4, "tri"),  # triclinic cell; Jacobi iteration terminated by polar_precision
    "pi_nopbc": ("input.in", 4, "nopbc"),  # polar_ewald off (thole_field_nopbc), polar_gamma 1.03, dipole rrms
    "pi_wolf": ("input.in", 4, "wolf"),  # Wolf electrostatics, rd_lrc off
    "pi_gs": ("input.in", 4, "gs"),  # Gauss-Seidel sweeps, dipole rrms
    # orientational bead moves (sorbate_orientation_site / sorbate_bondlength / sorbate_reducedMass): the four images start from restart files
    # with scattered orientations, so that a quarter of the bead moves is accepted and every row depends on the orientation sampler
    "pi_h2_orient": ("input.in", 4, "h2or"),
    "pi_ion1000": ("input.in", 4, "ion1000"),  # 1000 polarizable ions, 12 steps: rows and acceptance rates only (no final geometries kept)
}
LIBDIR = os.path.join(util.ROOT, "mpmcxx_amd")
ORACLE = os.path.join(util.ROOT, "oracle")


def rows(path):
    return [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]


def golden_ar(name):
    """'OUTPUT: AR = 0.31200 (0.00000 I/ 0.00000 R/ 0.00000 D/ 0.34456 BEAD' of the last averages block"""
    line = [ln for ln in open(os.path.join(util.GOLDEN, name, "golden_final_averages.txt")) if "AR =" in ln][-1]
    t = line.replace("(", " ").replace("/", " ").split()
    return float(t[3]), float(t[8]), float(t[10])


@pytest.fixture(scope="module")
def pimc_check(tmp_path_factory):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    subprocess.check_call(["make", "-s", "-C", ORACLE, "oracle"])
    exe = str(tmp_path_factory.mktemp("pimc") / "pimc_check")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "tests", "cpp", "pimc_check.cpp"),
                           "-L", LIBDIR, "-lmpmc_energy", "-L", ORACLE, "-lmpmc_oracle", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ORACLE}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("trial", [False, True], ids=["full", "trial_moves"])
@pytest.mark.parametrize("name", list(CASES))
def test_driver_with_oracle_evaluator_reproduces_the_stock_binary(pimc_check, name, trial, tmp_path):
    if name == "pi_ion1000" and trial:
        pytest.skip("the 1000-atom case runs once on the CPU (15 s of oracle evaluations)")
    inp, P, job = CASES[name]
    out = subprocess.run([pimc_check, os.path.join(util.GOLDEN, name, inp), str(P), str(tmp_path)] + (["--trial"] if trial else []),
                         stdout=subprocess.PIPE, text=True, check=True)
    r = json.loads(out.stdout)
    ours, gold = rows(os.path.join(tmp_path, "energy.dat")), rows(os.path.join(util.GOLDEN, name, "golden_energy.dat"))
    assert ours == gold  # every printed digit of every row
    ar, ar_d, ar_b = golden_ar(name)
    assert f"{r['AR']:.5f}" == f"{ar:.5f}"
    nd, nb = r["accept_displace"] + r["reject_displace"], r["accept_bead"] + r["reject_bead"]
    assert f"{(r['accept_displace'] / nd if nd else 0.0):.5f}" == f"{ar_d:.5f}"
    assert f"{(r['accept_bead'] / nb if nb else 0.0):.5f}" == f"{ar_b:.5f}"
    for k in range(P if os.path.exists(os.path.join(util.GOLDEN, name, "golden_final-0000.pqr")) else 0):
        a = pqr.read_pqr(os.path.join(tmp_path, f"final-{k:04d}.pqr"))["pos"]
        b = pqr.read_pqr(os.path.join(util.GOLDEN, name, f"golden_final-{k:04d}.pqr"))["pos"]
        assert np.abs(a - b).max() <= 1.0e-6  # both files carry 6 decimals


def test_driver_refuses_what_it_does_not_cover(pimc_check, tmp_path):
    src = open(os.path.join(util.GOLDEN, "pi001", "equilibrate.in")).read()
    for bad, code in (("ensemble                       pi_nvt", "ensemble uvt"), ("PI_trial_chain_length          4", "PI_trial_chain_length 8")):
        assert bad in src
    p = tmp_path / "uvt.in"
    p.write_text(src.replace("ensemble                       pi_nvt", "ensemble uvt").replace("Ar-Ar-4A.pqr", os.path.join(util.GOLDEN, "pi001", "Ar-Ar-4A.pqr")))
    out = subprocess.run([pimc_check, str(p), "8", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["error"] == 4004  # unsupported_setting
    p = tmp_path / "chain.in"
    p.write_text(src.replace("PI_trial_chain_length          4", "PI_trial_chain_length 8").replace("Ar-Ar-4A.pqr", os.path.join(util.GOLDEN, "pi001", "Ar-Ar-4A.pqr")))
    out = subprocess.run([pimc_check, str(p), "8", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["error"] == 4001  # invalid_setting: chain length must be in [1, P-1]
    p = tmp_path / "fh.in"
    p.write_text(src.replace("Ar-Ar-4A.pqr", os.path.join(util.GOLDEN, "pi001", "Ar-Ar-4A.pqr")) + "\nfeynman_hibbs on\n")
    out = subprocess.run([pimc_check, str(p), "8", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["error"] == 3000  # as the reference: no Feynman-Hibbs corrections in a PI run
    p = tmp_path / "orient.in"
    p.write_text(src.replace("Ar-Ar-4A.pqr", os.path.join(util.GOLDEN, "pi001", "Ar-Ar-4A.pqr")) + "\nsorbate_bondlength Ar\n")
    out = subprocess.run([pimc_check, str(p), "8", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["error"] == 3000  # a sorbate_* key without its value
    out = subprocess.run([pimc_check, os.path.join(util.GOLDEN, "pi001", "equilibrate.in"), "6", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1 and json.loads(out.stdout)["error"] == 9003  # the Trotter number must be a power of two >= 4


@pytest.fixture(scope="module", params=[False, True], ids=["plain", "openmp"])
def pimc_nvt(tmp_path_factory, request):
    """the example driver, built without and with -fopenmp (the facade's loops over the images then run on a few host threads, as the
    reference's bead loop does: PathIntegral.cpp:759-775) -- both builds have to reproduce the stock binary"""
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    exe = str(tmp_path_factory.mktemp("pimc") / "pimc_nvt")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall"] + (["-fopenmp"] if request.param else []) + ["-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "examples", "pimc_nvt.cpp"),
                           "-L", LIBDIR, "-lmpmc_energy", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("trial", [False, True], ids=["full", "trial_moves"])
@pytest.mark.parametrize("name", list(CASES))
def test_pimc_on_the_hip_path_reproduces_the_stock_binary(pimc_nvt, name, trial, tmp_path):
    """trial_moves: every move goes through mpmc_trial_* (per-move delta energies for the LJ dimer, a full evaluation behind the same
    calls for the polarizable box): same trajectory, same rows."""
    inp, P, job = CASES[name]
    out = subprocess.run([pimc_nvt, os.path.join(util.GOLDEN, name, inp), "-P", str(P), "-o", str(tmp_path)] + (["--trial"] if trial else []),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    ours, gold = rows(os.path.join(tmp_path, f"{job}.energy.dat")), rows(os.path.join(util.GOLDEN, name, "golden_energy.dat"))
    assert len(ours) == len(gold)
    for a, b in zip(ours, gold):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert abs(float(x) - float(y)) <= 1e-9 * max(abs(float(y)), 1.0) + 1.1e-6, (a, b)  # 6 printed decimals
    ar, ar_d, ar_b = golden_ar(name)
    assert f"{r['AR']:.5f}" == f"{ar:.5f}" and f"{r['AR_displace']:.5f}" == f"{ar_d:.5f}" and f"{r['AR_bead']:.5f}" == f"{ar_b:.5f}"
    for k in range(P if os.path.exists(os.path.join(util.GOLDEN, name, "golden_final-0000.pqr")) else 0):
        a = pqr.read_pqr(os.path.join(tmp_path, f"{job}.final-{k:04d}.pqr"))["pos"]
        b = pqr.read_pqr(os.path.join(util.GOLDEN, name, f"golden_final-{k:04d}.pqr"))["pos"]
        assert np.abs(a - b).max() <= 1.0e-6


def test_host_side_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The driver, the C++ facade and the file readers (everything above the C ABI) under -fsanitize=address,undefined with the oracle
    as evaluator: any report aborts the run (GPU sanitizers are not available on the pool; the host side is where the pointer work is)."""
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    subprocess.check_call(["make", "-s", "-C", ORACLE, "oracle"])
    exe = str(tmp_path / "pimc_san")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
                           "-Werror", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "tests", "cpp", "pimc_check.cpp"),
                           "-L", LIBDIR, "-lmpmc_energy", "-L", ORACLE, "-lmpmc_oracle", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ORACLE}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    for name, trial in (("pi_ion27", False), ("pi001", True)):
        inp, P, _ = CASES[name]
        out = subprocess.run([exe, os.path.join(util.GOLDEN, name, inp), str(P), str(tmp_path)] + (["--trial"] if trial else []),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        assert rows(os.path.join(tmp_path, "energy.dat")) == rows(os.path.join(util.GOLDEN, name, "golden_energy.dat"))
