"""the Gibbs-ensemble (nvt_gibbs) Monte Carlo driver (include/mpmc_gibbs.hpp: GibbsNVT) against trajectories made by
the REFERENCE's own functions.

tests/golden/gibbs_*/trajectory.json come from oracle/ref_gibbs_traj.cpp (build container): the reference's object code for
pick_Gibbs_move / make_move_Gibbs (displacement, coupled volume change, particle transfer) / energy / boltzmann_factor_NVT_Gibbs / restore,
driven step by step (the stock Gibbs_mc loop dies in unrelated bookkeeping).
  * CPU: tests/cpp/gibbs_check.cpp (the driver with the oracle as evaluator): every move type, every accept / reject, every atom count and
    volume equal; trial energies and Boltzmann factors to 1e-12; final geometries to 1e-12 A.
  * GPU: examples/gibbs_nvt.cpp (the HIP path through the C++ facade, box 0 on device 0, box 1 on device 1 when present): the same
    decisions, energies to 1e-9."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import util

LIBDIR = os.path.join(util.ROOT, "mpmcxx_amd")
ORACLE = os.path.join(util.ROOT, "oracle")
CASES = ["gibbs_lj", "gibbs_water", "gibbs_water_polar"]


def golden(name):
    with open(os.path.join(util.GOLDEN, name, "trajectory.json")) as f:
        return json.load(f)


def parse(stdout):
    return json.loads(stdout[stdout.find('{"initial'):])


def compare(ours, ref, tol, pos_tol):
    assert len(ours["steps"]) == len(ref["steps"])
    for k in range(2):
        assert util.close(ours["initial_energy"][k], ref["initial_energy"][k], tol)
    assert ours["volume_probability"] == ref["volume_probability"]
    kinds = set()
    for a, b in zip(ours["steps"], ref["steps"]):
        assert a["movetype"] == b["movetype"] and a["accepted"] == b["accepted"] and a["natoms"] == b["natoms"], b["step"]
        assert a["N"] == b["N"], b["step"]
        kinds.add((tuple(b["movetype"]), tuple(b["accepted"])))
        for key in ("final_energy", "boltzmann_factor", "energy", "volume"):
            for x, y in zip(a[key], b[key]):
                if isinstance(y, float) and (math.isnan(y) or math.isinf(y)):
                    assert (math.isnan(x) and math.isnan(y)) or x == y, (b["step"], key)
                else:
                    assert abs(x - y) <= tol * max(abs(y), 1e-6 if key == "boltzmann_factor" else 1.0), (b["step"], key, x, y)
    for box in ("final_box_0", "final_box_1"):
        pa, pb = np.array(ours[box]["pos"]), np.array(ref[box]["pos"])
        assert pa.shape == pb.shape and np.abs(pa - pb).max() <= pos_tol, box
        assert np.abs(np.array(ours[box]["basis"]) - np.array(ref[box]["basis"])).max() <= pos_tol
        assert ours[box]["mol_id"] == ref[box]["mol_id"] and ours[box]["charge"] == ref[box]["charge"]
    return kinds


@pytest.fixture(scope="module")
def gibbs_check(tmp_path_factory):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    subprocess.check_call(["make", "-s", "-C", ORACLE, "oracle"])
    exe = str(tmp_path_factory.mktemp("gibbs") / "gibbs_check")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(util.ROOT, "include"),
                           os.path.join(util.ROOT, "tests", "cpp", "gibbs_check.cpp"), "-L", LIBDIR, "-lmpmc_energy", "-L", ORACLE, "-lmpmc_oracle",
                           f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ORACLE}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("name", CASES)
def test_driver_with_the_oracle_reproduces_the_reference_made_trajectory(gibbs_check, name):
    ref = golden(name)
    out = subprocess.run([gibbs_check, os.path.join(util.GOLDEN, name, "input.in")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    kinds = compare(parse(out.stdout), ref, 1e-12, 1e-12)
    # the trajectory really exercises every branch: displacements accepted and rejected per box, transfers both ways, volume exchanges
    moves = {k[0] for k in kinds}
    assert {(2, 2), (0, 1), (1, 0), (5, 5)} <= moves
    assert any(k[0] == (2, 2) and k[1] in ((0, 1), (1, 0)) for k in kinds)  # the two boxes decide independently
    assert any(k[0] in ((0, 1), (1, 0)) and k[1] == (0, 0) for k in kinds) and any(k[0] in ((0, 1), (1, 0)) and k[1] == (1, 1) for k in kinds)


def test_settings_reader_refuses_what_the_driver_does_not_mirror(gibbs_check, tmp_path):
    src = open(os.path.join(util.GOLDEN, "gibbs_lj", "input.in")).read()
    for extra, code in (("spinflip_probability 0.1\n", 4004), ("", None)):
        p = tmp_path / "x.in"
        text = src.replace("boxA.pqr", os.path.join(util.GOLDEN, "gibbs_lj", "boxA.pqr")).replace("boxB.pqr", os.path.join(util.GOLDEN, "gibbs_lj", "boxB.pqr"))
        if code is None:
            text = text.replace("transfer_probability 0.3\n", "")  # missing_setting, SimulationControl.Gibbs.cpp:112-115
            code = 4003
        p.write_text(text + extra)
        out = subprocess.run([gibbs_check, str(p), "3"], stdout=subprocess.PIPE, text=True)
        assert out.returncode == 1 and json.loads(out.stdout)["error"] == code


@pytest.fixture(scope="module")
def gibbs_nvt(tmp_path_factory):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    exe = str(tmp_path_factory.mktemp("gibbs") / "gibbs_nvt")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "examples", "gibbs_nvt.cpp"),
                           "-L", LIBDIR, "-lmpmc_energy", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_gibbs_example_compiles_with_plain_gxx(gibbs_nvt):
    assert os.path.exists(gibbs_nvt)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_driver_on_the_hip_path_reproduces_the_reference_made_trajectory(gibbs_nvt, name):
    ref = golden(name)
    out = subprocess.run([gibbs_nvt, os.path.join(util.GOLDEN, name, "input.in")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    compare(parse(out.stdout), ref, 1e-9, 1e-9)


def test_driver_is_clean_under_address_and_ub_sanitizers(tmp_path):
    """The Gibbs driver inserts and erases molecules in the middle of the atom list: run it, with the oracle as evaluator, under
    -fsanitize=address,undefined (any report aborts the run)."""
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    subprocess.check_call(["make", "-s", "-C", ORACLE, "oracle"])
    exe = str(tmp_path / "gibbs_san")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
                           "-Werror", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "tests", "cpp", "gibbs_check.cpp"),
                           "-L", LIBDIR, "-lmpmc_energy", "-L", ORACLE, "-lmpmc_oracle", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{ORACLE}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    for name, steps in (("gibbs_water", 120), ("gibbs_water_polar", 40)):
        out = subprocess.run([exe, os.path.join(util.GOLDEN, name, "input.in"), str(steps)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        ours, ref = parse(out.stdout), golden(name)
        assert [s["accepted"] for s in ours["steps"]] == [s["accepted"] for s in ref["steps"][:steps]]
