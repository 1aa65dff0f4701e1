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


if __name__ == "__main__":
    main()
