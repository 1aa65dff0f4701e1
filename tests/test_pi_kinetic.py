"""the path-integral kinetic estimator (PI_calculate_kinetic / PI_chain_mass_length2,
reference src/SimulationControl.PathIntegral.cpp:806-965).

Pins:
  * tests/golden/pi000 -- the reference's OWN shipped sample output (sample-input/pi000-free-argon-2K): the four final bead
    geometries and the last row of Ar2K.energy.dat (kinetic column) that the reference wrote for exactly that state;
  * tests/golden/pi001, pi_ion27 -- stock-binary runs made here (oracle/make_pi_golden.sh), final beads + last energy row.
The geometries are printed with 6 decimals, so the recomputed estimator agrees to the rounding of the coordinates
(tolerance below), while oracle / library / C++ facade / Python host layer must agree with each other bit for bit.
"""
import json
import os
import subprocess

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from mpmcxx_amd import energy, pi, pqr
import oracle

CASES = {
    # name: (input file, bead file pattern, energy.dat, P)
    "pi000": ("equilibrate.in", "Ar2K.final-%04d.pqr", "Ar2K.energy.dat", 4),
    "pi001": ("equilibrate.in", "golden_final-%04d.pqr", "golden_energy.dat", 8),
    "pi_ion27": ("input.in", "golden_final-%04d.pqr", "golden_energy.dat", 4),
}


def load(name):
    inp, pat, en, P = CASES[name]
    d = os.path.join(util.GOLDEN, name)
    cfg = pqr.read_input(os.path.join(d, inp))
    beads = [pqr.read_pqr(os.path.join(d, pat % k)) for k in range(P)]
    row = [ln for ln in open(os.path.join(d, en)) if not ln.startswith("#")][-1].split()
    cols = dict(zip(["step", "energy", "coulombic", "rd", "polar", "vdw", "kinetic", "kin_temp", "N"], [float(x) for x in row[:9]]))
    return cfg, beads, cols, [os.path.join(d, pat % k) for k in range(P)], os.path.join(d, inp)


def coordinate_rounding_tolerance(beads, T, P):
    """|dK| for coordinates rounded to 1e-6 A: K = const - 0.5 w2 sum M d^2 / kB, d(d^2) <= 2 |d| 1e-6 sqrt(3) per link."""
    kB, hbar2, amu = 1.3806503e-23, 1.11211999e-68, 1.66053873e-27
    w2 = P / ((1.0 / (kB * T)) ** 2 * hbar2)
    pos = np.stack([b["pos"] for b in beads])
    d = np.linalg.norm(pos - np.roll(pos, -1, axis=0), axis=2)  # single-atom molecules in all three cases
    return float(0.5 * w2 / kB * np.sum(beads[0]["mass"][None, :] * amu * 1e-20 * 2 * d * 1e-6 * np.sqrt(3.0))) + 1e-6


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_kinetic_matches_the_reference_written_value(name):
    cfg, beads, cols, _, _ = load(name)
    T, P = cfg["options"]["temperature"], len(beads)
    pos = np.stack([b["pos"] for b in beads])
    k, chain = oracle.pi_kinetic(pos, beads[0]["mass"], beads[0]["mol_id"], beads[0]["frozen"], T)
    assert cols["N"] == len(set(beads[0]["mol_id"].tolist()))
    assert abs(k - cols["kinetic"]) <= coordinate_rounding_tolerance(beads, T, P), (k, cols["kinetic"])
    assert chain > 0


@pytest.mark.parametrize("name", list(CASES))
def test_library_and_python_host_layer_equal_the_oracle_bit_for_bit(name):
    cfg, beads, cols, _, _ = load(name)
    T, P = cfg["options"]["temperature"], len(beads)
    pos = np.stack([b["pos"] for b in beads])
    k_ref, chain_ref = oracle.pi_kinetic(pos, beads[0]["mass"], beads[0]["mol_id"], beads[0]["frozen"], T)
    coms = []
    for b in beads:
        c, m, mv = pi.molecule_coms(b["pos"], b["mass"], b["mol_id"], b["frozen"])
        coms.append(c)
    k, chain = pi.pi_calculate_kinetic(np.stack(coms), m, mv, P, T)
    assert chain == chain_ref and k == k_ref
    assert pi.pi_calculate_energy(k, -1.5) == k + -1.5


def test_frozen_molecules_and_multi_atom_molecules():
    """water64 (3-site molecules, mass-weighted COM) with half of the molecules frozen: frozen ones leave both N and the chain."""
    atoms, basis, opts = util.load_fixture("water64_polar")
    rng = np.random.default_rng(5)
    P, T = 6, 77.0
    atoms = dict(atoms)
    fr = np.zeros(len(atoms["mass"]), dtype=np.int32)
    mols = np.unique(atoms["mol_id"])
    for m_ in mols[::2]:
        fr[atoms["mol_id"] == m_] = 1
    atoms["frozen"] = fr
    pos = np.stack([atoms["pos"] + rng.normal(scale=0.03, size=atoms["pos"].shape) for _ in range(P)])
    k_ref, chain_ref = oracle.pi_kinetic(pos, atoms["mass"], atoms["mol_id"], fr, T)
    coms = []
    for s in range(P):
        c, m, mv = pi.molecule_coms(pos[s], atoms["mass"], atoms["mol_id"], fr)
        coms.append(c)
    assert mv.sum() == len(mols) - len(mols[::2])
    k, chain = pi.pi_calculate_kinetic(np.stack(coms), m, mv, P, T)
    assert chain == chain_ref and k == k_ref
    # classical limit: coincident images give exactly the equipartition term 1.5 N T P
    same = np.stack([coms[0]] * P)
    k0, chain0 = pi.pi_calculate_kinetic(same, m, mv, P, T)
    assert chain0 == 0.0 and k0 == pytest.approx(1.5 * mv.sum() * T * P, rel=1e-15)


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    from mpmcxx_amd import build as mbuild

    mbuild.build_library()
    exe = str(tmp_path_factory.mktemp("cli") / "energy_cli")
    libdir = os.path.join(util.ROOT, "mpmcxx_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "examples", "energy_cli.cpp"),
                           "-L", libdir, "-lmpmc_energy", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("name", list(CASES))
def test_cpp_facade_kinetic(cli, name):
    cfg, beads, cols, files, inp = load(name)
    T, P = cfg["options"]["temperature"], len(beads)
    out = subprocess.run([cli, inp, "--pi-kinetic"] + files, stdout=subprocess.PIPE, text=True, check=True)
    r = json.loads(out.stdout)
    pos = np.stack([b["pos"] for b in beads])
    k_ref, chain_ref = oracle.pi_kinetic(pos, beads[0]["mass"], beads[0]["mol_id"], beads[0]["frozen"], T)
    assert r["P"] == P and r["N"] == cols["N"]
    assert r["kinetic"] == k_ref and r["chain_mass_len2"] == chain_ref
    # single-chain variant (the Boltzmann-factor measure of one perturbed molecule) against the same arithmetic on molecule 0
    sel = beads[0]["mol_id"] == beads[0]["mol_id"][0]
    _, chain0 = oracle.pi_kinetic(pos[:, sel], beads[0]["mass"][sel], beads[0]["mol_id"][sel], beads[0]["frozen"][sel], T)
    assert r["chain0"] == chain0


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, beads, cols, _, _ = load("pi001")
        P = len(beads)
        coms = []
        for b in pi.beads_of_rank(P, rank, world):
            c, m, mv = pi.molecule_coms(beads[b]["pos"], beads[b]["mass"], beads[b]["mol_id"], beads[b]["frozen"])
            coms.append(c)
        out[rank] = pi.pi_calculate_kinetic(np.stack(coms), m, mv, P, cfg["options"]["temperature"], rank, world)
    finally:
        dist.destroy_process_group()


def test_kinetic_world2_gloo_equals_single_process():
    """beads sharded round-robin over 2 ranks: the ring of adjacent images crosses ranks; one all-gather of the COMs."""
    cfg, beads, cols, _, _ = load("pi001")
    pos = np.stack([b["pos"] for b in beads])
    k_ref, chain_ref = oracle.pi_kinetic(pos, beads[0]["mass"], beads[0]["mol_id"], beads[0]["frozen"], cfg["options"]["temperature"])
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 500) + 7
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    for r in range(2):
        assert out[r] == (k_ref, chain_ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pi001", "pi_ion27"])
def test_full_estimator_on_reference_final_beads(cli, name):
    """PI_calculate_energy (kinetic + HIP potential over the P beads) on the geometries the stock binary ended with, against
    the row the stock binary printed for that state.  Coordinates carry 6 decimals: the potential terms agree to ~1e-6 relative."""
    cfg, beads, cols, files, inp = load(name)
    out = subprocess.run([cli, inp, "--pi"] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    tolk = coordinate_rounding_tolerance(beads, cfg["options"]["temperature"], len(beads))
    assert abs(r["kinetic"] - cols["kinetic"]) <= tolk
    for ours, ref in (("rd", "rd"), ("es", "coulombic"), ("polar", "polar")):
        assert abs(r[ours] - cols[ref]) <= 2e-6 * abs(cols[ref]) + 0.02, (ours, r[ours], cols[ref])
    assert abs(r["energy"] - cols["energy"]) <= 2e-6 * abs(cols["energy"]) + 0.02 + tolk
