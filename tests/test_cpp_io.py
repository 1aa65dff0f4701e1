"""The C++ readers/writer of the reference's on-disk formats (include/mpmc_io.hpp): CPU tests compare them with the Python
readers and the reference's golden box data; the GPU test runs examples/energy_cli on the golden inputs."""
import json
import os
import subprocess

import numpy as np
import pytest

import util
from mpmcxx_amd import pqr

SRC = os.path.join(util.ROOT, "examples", "energy_cli.cpp")
LIBDIR = os.path.join(util.ROOT, "mpmcxx_amd")


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    exe = str(tmp_path_factory.mktemp("cli") / "energy_cli")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "include"), SRC, "-L", LIBDIR, "-lmpmc_energy",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("name", ["ion216_triclinic", "water64_polar", "ion216_alpha", "ion216_frozen", "lj64"])
def test_cpp_readers_match_python_readers_and_reference_box(cli, name):
    inp = os.path.join(util.GOLDEN, f"{name}.in")
    out = subprocess.run([cli, inp, "--parse"], stdout=subprocess.PIPE, text=True, check=True)
    r = json.loads(out.stdout)
    atoms, basis, o = pqr.load_case(inp)
    g = util.golden(name)
    assert r["n"] == g["natoms"] and r["volume"] == g["volume"] and r["cutoff"] == g["cutoff"]
    assert r["ewald_alpha"] == g["ewald_alpha"] and r["polar_ewald_alpha"] == g["polar_ewald_alpha"]
    a = np.array(r["atoms"])
    assert np.array_equal(a[:, 0:3], atoms["pos"]) and np.array_equal(a[:, 3], atoms["mass"]) and np.array_equal(a[:, 4], atoms["charge"])
    assert np.array_equal(a[:, 5], atoms["polarizability"]) and np.array_equal(a[:, 6], atoms["epsilon"]) and np.array_equal(a[:, 7], atoms["sigma"])
    assert np.array_equal(a[:, 8].astype(int), atoms["mol_id"]) and np.array_equal(a[:, 9].astype(int), atoms["frozen"])
    want = [o["rd_only"], o["rd_lrc"], o["polarization"], o["polar_iterative"], o["polar_ewald"], o["polar_max_iter"], o["polar_rrms"], o["ewald_kmax"]]
    assert r["options"][:8] == [int(x) for x in want]
    assert r["options"][8:] == [float(o["polar_precision"]), float(o["polar_gamma"]), float(o["polar_damp"])]
    assert r["unsupported"] == 0


def test_write_pqr_roundtrips(cli, tmp_path):
    inp = os.path.join(util.GOLDEN, "water64_polar.in")
    outp = str(tmp_path / "rt.pqr")
    subprocess.check_call([cli, inp, "--write", outp])
    a0 = pqr.read_pqr(os.path.join(util.GOLDEN, "water64_polar.pqr"))
    a1 = pqr.read_pqr(outp)
    assert np.array_equal(a0["mol_id"], a1["mol_id"]) and np.array_equal(a0["frozen"], a1["frozen"])
    assert np.allclose(a0["pos"], a1["pos"], atol=5e-7) and np.allclose(a0["charge"], a1["charge"], atol=5e-5 * 408.7816)
    assert np.allclose(a0["sigma"], a1["sigma"], atol=5e-6) and np.allclose(a0["polarizability"], a1["polarizability"], atol=5e-6)


def test_out_of_scope_keyword_sets_the_refusal_flag(cli, tmp_path):
    src = open(os.path.join(util.GOLDEN, "lj64.in")).read() + "spectre on\nrd_crystal on\n"
    p = tmp_path / "x.in"
    p.write_text(src.replace("lj64.pqr", os.path.join(util.GOLDEN, "lj64.pqr")))
    out = subprocess.run([cli, str(p), "--parse"], stdout=subprocess.PIPE, text=True, check=True)
    assert json.loads(out.stdout)["unsupported"] == 4 + 8  # MPMC_FLAG_RD_CRYSTAL | MPMC_FLAG_SPECTRE


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ion216_polar", "water64_polar", "ion216_triclinic", "lj1000"])
def test_energy_cli_reproduces_reference(cli, name):
    out = subprocess.run([cli, os.path.join(util.GOLDEN, f"{name}.in")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    g = util.golden(name)
    for k in ("total", "rd", "es", "polar", "es_real", "es_recip", "es_self"):
        assert util.close(r[k], g[k]), (k, r[k], g[k])
    assert r["n_lj_in_cutoff"] == g["n_lj_in_cutoff"]
    if g["polar"] != 0:
        assert util.max_rel(r["mu0"], g["mu"][:3]) < 1e-9 and r["polar_iterations"] == g["polar_iterations"]
