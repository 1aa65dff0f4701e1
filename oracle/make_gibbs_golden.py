#!/usr/bin/env python3
"""tests/golden/gibbs_bf.json: inputs and outputs of the REFERENCE's own boltzmann_factor_NVT_Gibbs
(src/SimulationControl.Gibbs.cpp:358), called through oracle/_ref/ref_gibbs_bf (build container only).
Data only: every case is a line of numbers in, a line of numbers out."""
import json
import os
import random
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BIN = os.path.join(HERE, "_ref", "ref_gibbs_bf")
INSERT, REMOVE, DISPLACE, ADIABATIC, SPINFLIP, VOLUME, BEADS = range(7)


def main():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    rng = random.Random(20261004)
    cases = []

    def case(ma, mb, bad=None):
        T = rng.choice([40.0, 77.0, 150.0, 298.15])
        ia, ib = rng.uniform(-5e4, -1e3), rng.uniform(-2e4, -1e2)
        fa, fb = ia + rng.gauss(0, 60), ib + rng.gauss(0, 60)
        if bad == "a":
            fa = rng.choice([float("inf"), float("nan")])
        if bad == "b":
            fb = rng.choice([float("inf"), float("nan"), float("-inf")])
        NA, NB = float(rng.randint(1, 400)), float(rng.randint(0, 300))
        VA, VB = rng.uniform(5e3, 8e4), rng.uniform(5e3, 2e5)
        ck = VA * rng.uniform(0.95, 1.05)
        cases.append({"movetype": [ma, mb], "temperature": T, "init_energy": [ia, ib], "final_energy": [fa, fb], "N": [NA, NB], "volume": [VA, VB],
                      "checkpoint_volume_0": ck})

    for _ in range(12):
        case(DISPLACE, DISPLACE)
        case(REMOVE, INSERT)
        case(INSERT, REMOVE)
        case(VOLUME, VOLUME)
    for ma, mb in ((DISPLACE, DISPLACE), (REMOVE, INSERT), (INSERT, REMOVE), (VOLUME, VOLUME)):
        case(ma, mb, bad="a")
        case(ma, mb, bad="b")
    for ma, mb in ((DISPLACE, VOLUME), (VOLUME, DISPLACE), (INSERT, INSERT), (REMOVE, REMOVE), (ADIABATIC, ADIABATIC), (BEADS, BEADS), (VOLUME, INSERT)):
        case(ma, mb)
    fmt = lambda x: repr(float(x)).replace("inf", "inf").replace("nan", "nan")
    text = "".join("%d %d %s %s %s %s %s %s %s %s %s %s\n" % (c["movetype"][0], c["movetype"][1], fmt(c["temperature"]), fmt(c["init_energy"][0]),
                                                             fmt(c["final_energy"][0]), fmt(c["init_energy"][1]), fmt(c["final_energy"][1]), fmt(c["N"][0]),
                                                             fmt(c["volume"][0]), fmt(c["N"][1]), fmt(c["volume"][1]), fmt(c["checkpoint_volume_0"])) for c in cases)
    out = subprocess.run([BIN], input=text, stdout=subprocess.PIPE, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == len(cases)
    for c, ln in zip(cases, out):
        bfa, bfb, ea, eb, status = ln.split()
        c["ref"] = {"boltzmann_factor": [float(bfa), float(bfb)], "energy": [float(ea), float(eb)], "status": int(status),
                    "note": "boltzmann_factor -1 = left untouched by the reference (sentinel written by the driver)"}
    with open(os.path.join(ROOT, "tests", "golden", "gibbs_bf.json"), "w") as f:
        json.dump({"generator": "oracle/make_gibbs_golden.py + oracle/_ref/ref_gibbs_bf (reference SimulationControl::boltzmann_factor_NVT_Gibbs)", "cases": cases}, f, indent=0)
        f.write("\n")
    print(len(cases), "cases")


def trajectories():
    """tests/golden/gibbs_*/: two-box inputs in the reference's own formats + the trajectory oracle/_ref/ref_gibbs_traj made for them
    (the reference's own pick_Gibbs_move / make_move_Gibbs / energy / boltzmann_factor_NVT_Gibbs / restore, driven step by step)."""
    import sys

    sys.path.insert(0, ROOT)
    from mpmcxx_amd import gen_box

    traj = os.path.join(HERE, "_ref", "ref_gibbs_traj")
    cases = {
        # name: (molecules A, molecules B, L, extra options, steps)
        "gibbs_water": (24, 10, 16.0, {"ewald_kmax": 5}, 200),
        "gibbs_water_polar": (20, 8, 16.0, dict(gen_box.POLAR_OPTS, ewald_kmax=5), 80),
        "gibbs_lj": (0, 0, 18.0, {"rd_only": "on"}, 150),
    }
    for name, (na, nb, L, extra, steps) in cases.items():
        d = os.path.join(ROOT, "tests", "golden", name)
        os.makedirs(d, exist_ok=True)
        if name == "gibbs_lj":  # single-site atoms: 40 in the dense box, 12 in the dilute one
            A = gen_box.lattice_box(40, L, 21, charged=False, alpha=0.0)
            B = gen_box.lattice_box(12, L, 22, charged=False, alpha=0.0)
        else:
            A = gen_box.molecular_box(na, L, 5, extra_neutral=False)
            B = gen_box.molecular_box(nb, L, 9, extra_neutral=False)
        gen_box.write_pqr(os.path.join(d, "boxA.pqr"), A)
        gen_box.write_pqr(os.path.join(d, "boxB.pqr"), B)
        opts = {"job_name": name, "ensemble": "nvt_gibbs", "temperature": 300.0, "numsteps": steps, "corrtime": 10, "seed": 7, "move_factor": 0.05,
                "rot_factor": 0.05, "transfer_probability": 0.3, "volume_probability": 0.1, "volume_change_factor": 0.25}
        opts.update(extra)
        lines = [f"{k} {gen_box._fmt(v)}" for k, v in opts.items()]
        lines += [f"basis1 {L!r} 0.0 0.0", f"basis2 0.0 {L!r} 0.0", f"basis3 0.0 0.0 {L!r}", "pqr_input boxA.pqr", "pqr_input_B boxB.pqr"]
        lines += [f"{k} off" for k in ("pop_histogram", "traj_output", "energy_output", "dipole_output", "field_output")]
        with open(os.path.join(d, "input.in"), "w") as f:
            f.write("\n".join(lines) + "\n")
        p = subprocess.run([traj, "input.in", str(steps)], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        txt = p.stdout
        res = json.loads(txt[txt.rfind("\n{\"initial") + 1:])
        res["generator"] = "oracle/make_gibbs_golden.py + oracle/_ref/ref_gibbs_traj (the reference's own move / energy / acceptance functions)"
        with open(os.path.join(d, "trajectory.json"), "w") as f:
            json.dump(res, f, indent=0, separators=(",", ":"))
            f.write("\n")
        for junk in os.listdir(d):
            if junk not in ("input.in", "boxA.pqr", "boxB.pqr", "trajectory.json"):
                os.remove(os.path.join(d, junk))
        import collections

        print(name, len(res["steps"]), "steps", dict(collections.Counter(tuple(s["movetype"]) for s in res["steps"])), "accepted", sum(s["accepted"][0] for s in res["steps"]))


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "trajectories":
        trajectories()
    else:
        main()
        trajectories()
