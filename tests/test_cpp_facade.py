"""The C++ host facade (include/mpmc_system.hpp): compiles against the C ABI with plain g++ (CPU test) and, on the
GPU, reproduces the reference's golden energies when driven the way the reference's MC loop drives System."""
import json
import os
import subprocess

import numpy as np
import pytest

import util

SRC = os.path.join(util.ROOT, "tests", "cpp", "facade_check.cpp")
LIBDIR = os.path.join(util.ROOT, "mpmcxx_amd")


def build(tmp_path):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    exe = os.path.join(tmp_path, "facade_check")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "include"), SRC, "-L", LIBDIR, "-lmpmc_energy",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def dump(name, path):
    atoms, basis, o = util.load_fixture(name)
    with open(path, "w") as f:
        f.write(f"{atoms['pos'].shape[0]}\n")
        for row in np.asarray(basis):
            f.write(" ".join(repr(float(x)) for x in row) + "\n")
        f.write(" ".join(str(int(o[k])) for k in ("rd_only", "rd_lrc", "polarization", "polar_iterative", "polar_ewald", "polar_max_iter", "polar_rrms", "ewald_kmax")))
        f.write(" " + " ".join(repr(float(o[k] or 0.0)) for k in ("polar_precision", "polar_gamma", "polar_damp", "ewald_alpha", "polar_ewald_alpha")) + "\n")
        for i in range(atoms["pos"].shape[0]):
            p = atoms["pos"][i]
            vals = [p[0], p[1], p[2], atoms["mass"][i], atoms["charge"][i], atoms["polarizability"][i], atoms["epsilon"][i], atoms["sigma"][i]]
            f.write(" ".join(repr(float(v)) for v in vals) + f" {int(atoms['mol_id'][i])} {int(atoms['frozen'][i])}\n")


def test_facade_compiles_and_links_with_plain_gxx(tmp_path):
    exe = build(str(tmp_path))
    assert os.path.exists(exe)
    # the text dump parses; without a GPU the facade throws the library's error code as an int (no CPU fallback)
    inp = os.path.join(tmp_path, "case.txt")
    dump("ion216_polar", inp)
    out = subprocess.run([exe, inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert "cannot read" not in out.stderr
    from mpmcxx_amd import energy

    if energy.device_count() == 0:
        assert out.returncode == 1 and json.loads(out.stdout.strip())["error"] == 101


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ion216_polar", "water64_polar", "lj64"])
def test_facade_reproduces_reference(tmp_path, name):
    exe = build(str(tmp_path))
    inp = os.path.join(tmp_path, "case.txt")
    dump(name, inp)
    out = subprocess.run([exe, inp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    g = util.golden(name)
    assert util.close(r["energy"], g["total"]) and util.close(r["rd"], g["rd"]) and util.close(r["lj"], g["rd"])
    if g["es"] != 0:
        assert util.close(r["es"], g["es"]) and util.close(r["coulombic"], g["es"])
    if g["polar"] != 0:
        assert util.close(r["polar"], g["polar"])
        assert util.max_rel(r["mu0"], g["mu"][:3]) < 1e-9
        assert r["iters"] == g["polar_iterations"]
    assert r["N"] == g["N"] and util.close(r["NU"], g["NU"])
    assert r["e_back"] == r["energy"] and r["e_trial"] != r["energy"]  # reject path restores the energy bit for bit
    assert r["pi_rd"] == r["pi_rd_check"]
    assert r["thrown"] == 4004  # reference unsupported_setting
