#!/usr/bin/env python3
"""Inputs of tests/golden/pi_h2_orient: the 8 rigid diatomics of pi_h2 with a DIFFERENT orientation in each of the 4 images.

The orientational bead moves of the reference (src/SimulationControl.PathIntegral.cpp:1559-1698) are rejected without exception when
all images start from one geometry (see include/mpmc_pimc.hpp, PI_NVT_boltzmann_factor); started from per-image restart files with
scattered orientations (`parallel_restarts on`) about half of them shorten the orientational chain and are accepted, so that the
stock binary's energy.dat rows depend on every digit of the orientation sampler and of Molecule::orient.  This script writes those
restart files (seeded, deterministic); oracle/make_pi_golden.sh then runs the stock binary on them.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "..", "tests", "golden")
rows = [ln.split() for ln in open(os.path.join(G, "pi_h2", "h2.pqr")) if ln.startswith("ATOM")]
rng = np.random.default_rng(20261004)
pos = np.array([[float(t[6]), float(t[7]), float(t[8])] for t in rows])
mol = np.array([int(t[5]) for t in rows])
for image in range(4):
    out = pos.copy()
    for m in np.unique(mol):
        k = np.where(mol == m)[0]
        c = out[k].mean(axis=0)  # equal masses
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.0, np.pi)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        out[k] = (out[k] - c) @ R.T + c + rng.normal(scale=0.05, size=3)  # (+ a small spread of the image's centre of mass)
    with open(os.path.join(G, "pi_h2_orient", f"h2or.restart-{image:04d}.pqr"), "w") as f:
        for t, p in zip(rows, out):
            f.write("ATOM  %5d %-4s %-3s %-1s %4d    %11.6f %11.6f %11.6f %9.5f %9.5f %8.5f %10.5f %8.5f %7.5f %7.5f\n" % (
                int(t[1]), t[2], t[3], t[4], int(t[5]), p[0], p[1], p[2], float(t[9]), float(t[10]), float(t[11]), float(t[12]), float(t[13]), 0.0, 0.0))
print("wrote 4 restart files")
