"""The two-box (Gibbs ensemble) side of the path.

CPU: mpmc_gibbs_boltzmann_factor against the REFERENCE's own boltzmann_factor_NVT_Gibbs (src/SimulationControl.Gibbs.cpp:358-522;
tests/golden/gibbs_bf.json from oracle/make_gibbs_golden.py: the reference's object code called on bare System objects).
GPU: the two boxes evaluated together (mpmc_gibbs_energy: SimulationControl.Gibbs.cpp:179-180) -- on one device, and box 0 -> device 0,
box 1 -> device 1 where two devices are visible -- through the move sequence of test_gpu_box_moves.py (volume exchange, particle transfer,
displacements, rejections), every evaluation checked against the oracle."""
import json
import math
import os

import numpy as np
import pytest

import util
from mpmcxx_amd import energy


def same(a, b):
    if math.isnan(b):
        return math.isnan(a)
    return a == b or abs(a - b) <= 2e-15 * abs(b)


def test_boltzmann_factor_matches_the_reference_function():
    with open(os.path.join(util.GOLDEN, "gibbs_bf.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 60
    kinds = set()
    for c in cases:
        rc, bf, en = energy.gibbs_boltzmann_factor(c["movetype"], c["temperature"], c["init_energy"], c["final_energy"], c["N"], c["volume"],
                                                   c["checkpoint_volume_0"], current=(-1.0, -1.0))
        ref = c["ref"]
        kinds.add((tuple(c["movetype"]), ref["status"]))
        if ref["status"] != 0:  # the reference throws (20000: the boxes disagree; 102: not a Gibbs move)
            assert rc == ref["status"], c
            continue
        assert rc == 0, c
        for k in range(2):
            assert same(bf[k], ref["boltzmann_factor"][k]), (c, bf)
            assert same(en[k], ref["energy"][k]), (c, en)
    assert len(kinds) >= 8  # every branch of the function is in the file


@pytest.mark.gpu
def test_two_boxes_together_equal_two_boxes_alone():
    a, basis, opts = util.load_fixture("ion216_polar")
    half = a["mol_id"] < 100
    b = {k: v[half].copy() for k, v in a.items()}
    A, B = energy.System(a, basis, opts), energy.System(b, basis, opts)
    ea, eb = energy.gibbs_energy(A, B)
    assert ea == A.energy() and eb == B.energy()
    with pytest.raises(energy.MpmcError):
        energy.gibbs_energy(A, A)
    A.close()
    B.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["ion216_polar", "water64_polar"])
def test_gibbs_boxes_on_two_devices(fixture):
    """box 0 -> device 0, box 1 -> device 1 (north_star: Gibbs dual-box energies shard across the GPUs of a node)"""
    if energy.device_count() < 2:
        pytest.skip("needs two GPUs")
    import test_gpu_box_moves as bm

    rng = np.random.default_rng(11)
    a, basis, opts = util.load_fixture(fixture)
    mols = np.unique(a["mol_id"])
    keep = np.isin(a["mol_id"], mols[: len(mols) // 2 // 2 * 2])
    b = {k: v[keep].copy() for k, v in a.items()}
    A, B = bm.Box(a, basis, opts), bm.Box(b, np.array(basis).copy(), opts)
    B.sys.close()
    B.sys = energy.System(B.atoms, B.basis, opts, device=1)
    for step in range(6):
        kind = ("volume", "transfer", "displace")[step % 3]
        if kind == "volume":
            va, vb = abs(np.linalg.det(A.basis)), abs(np.linalg.det(B.basis))
            f = np.exp((rng.random() - 0.5) * 0.1)
            A.scale_volume(f)
            B.scale_volume((vb + va - va * f) / vb)
        elif kind == "transfer":
            m = int(rng.choice(A.molecules()))
            gone = A.remove_molecule(m)
            B.insert_molecule(gone, (0.5 - rng.random(3)) @ B.basis, at_index=0)
        else:
            for bx in (A, B):
                bx.displace(int(rng.choice(bx.molecules())), rng.normal(scale=0.2, size=3))
        ea, eb = energy.gibbs_energy(A.sys, B.sys)
        from oracle import OracleSystem

        for bx, e in ((A, ea), (B, eb)):
            ref = OracleSystem(bx.atoms, bx.basis, bx.opts).energy()
            if np.isfinite(ref["energy"]):
                assert util.close(e, ref["energy"], 1e-9), (step, kind)
    A.close()
    B.close()


# ---- C++ facade (include/mpmc_gibbs.hpp, examples/gibbs_boxes.cpp) -------------------------------------------------------------------
@pytest.fixture(scope="module")
def gibbs_cli(tmp_path_factory):
    import subprocess

    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    libdir = os.path.join(util.ROOT, "mpmcxx_amd")
    exe = str(tmp_path_factory.mktemp("gibbs") / "gibbs_boxes")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "examples", "gibbs_boxes.cpp"),
                           "-L", libdir, "-lmpmc_energy", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_gibbs_facade_compiles_with_plain_gxx(gibbs_cli):
    assert os.path.exists(gibbs_cli)


@pytest.mark.gpu
def test_gibbs_facade_places_the_boxes_and_reproduces_the_reference_energies(gibbs_cli):
    import subprocess

    a, b = "ion216_polar", "water64_polar"
    out = subprocess.run([gibbs_cli, os.path.join(util.GOLDEN, f"{a}.in"), os.path.join(util.GOLDEN, f"{b}.in"), "--displace", "0.05", "-0.02", "0.03"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["devices"] == [0, 1 if energy.device_count() > 1 else 0]
    assert util.close(r["initial_energy"][0], util.golden(a)["total"]) and util.close(r["initial_energy"][1], util.golden(b)["total"])
    # the displaced boxes against the oracle, the factors against exp(-dE / T)
    from oracle import OracleSystem

    for k, name in enumerate((a, b)):
        atoms, basis, opts = util.load_fixture(name)
        first = atoms["mol_id"] == atoms["mol_id"][0]
        pos = atoms["pos"].copy()
        pos[first] += np.array([0.05, -0.02, 0.03])
        ref = OracleSystem(dict(atoms, pos=pos), basis, opts).energy()
        assert util.close(r["final_energy"][k], ref["energy"]), name
        assert util.close(r["boltzmann_factor"][k], math.exp(-(r["final_energy"][k] - r["initial_energy"][k]) / r["temperature"]), 1e-12)


@pytest.mark.gpu
def test_pi_ensemble_bound_to_an_rccl_communicator(tmp_path):
    """PathIntegralEnsemble::use_comm: the facade's cross-rank hook on mpmc_pi_gather_beads (one-rank communicator on a one-GPU box)"""
    import subprocess

    import test_cpp_io as tc

    libdir = os.path.join(util.ROOT, "mpmcxx_amd")
    exe = str(tmp_path / "energy_cli")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "include"), tc.SRC, "-L", libdir, "-lmpmc_energy",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    gold = os.path.join(util.GOLDEN, "pi_ion27")
    beads = sorted(f for f in os.listdir(gold) if f.endswith(".pqr") and "final" in f)
    if len(beads) < 2:
        beads = sorted(f for f in os.listdir(gold) if f.endswith(".pqr"))
    inp = [f for f in os.listdir(gold) if f.endswith(".in")][0]
    args = [os.path.join(gold, inp)]
    outs = []
    for mode in ("--pi", "--pi-rccl"):
        o = subprocess.run([exe] + args + [mode] + [os.path.join(gold, b) for b in beads], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=gold)
        assert o.returncode == 0, o.stdout + o.stderr
        outs.append(json.loads(o.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] and outs[0]["P"] == len(beads)
