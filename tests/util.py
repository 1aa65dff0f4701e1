"""shared helpers of the test-suite (tests may use the oracle; the product package may not)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from mpmcxx_amd import pqr  # noqa: E402

REL_TOL = 1e-9  # BASELINE.json north_star: energies within 1e-9 relative of the reference CPU path

SMALL = ["ar2", "lj64", "ion64_es", "ion216_polar", "ion216_polar_nopbc", "ion216_triclinic", "ion216_frozen",
         "ion216_precision", "ion216_gamma", "ion216_alpha", "water64_polar", "lj1000", "ion1000_polar",
         "ion216_wolf", "water64_fh2", "water64_fh4", "ion216_fh4_polar", "ion216_gs", "water64_gs_precision", "ion1000_gs", "ion216_framework",
         "ion1000_triclinic"]
LARGE = ["ion10k_es", "ion10k_polar", "ion8000_triclinic"]

ENERGY_KEYS = [("energy", "total"), ("rd_energy", "rd"), ("coulombic_energy", "es"), ("polarization_energy", "polar"),
               ("es_real", "es_real"), ("es_recip", "es_recip"), ("es_self", "es_self"),
               ("lj_pairs", "lj_pairs"), ("lrc_pair", "lrc_pair"), ("lrc_self", "lrc_self")]
COUNT_KEYS = ["n_pairs", "n_intra", "n_rd_excluded", "n_es_excluded", "n_frozen", "n_lj_in_cutoff"]


def golden(name):
    with open(os.path.join(GOLDEN, f"{name}.json")) as f:
        return json.load(f)


def load_fixture(name):
    """(atoms, basis, options) parsed from the committed reference-format files."""
    return pqr.load_case(os.path.join(GOLDEN, f"{name}.in"))


def load_generated(name, tmpdir):
    """large boxes are regenerated deterministically instead of being committed as text."""
    from mpmcxx_amd import gen_box

    inp, _ = gen_box.materialize(name, str(tmpdir))
    return pqr.load_case(inp)


def close(a, b, tol=REL_TOL):
    if b == 0.0:
        return abs(a) <= tol
    return abs(a - b) <= tol * abs(b)


def assert_energies(res, g, rd_only, tol=REL_TOL, label="", wolf=False):
    bad = []
    for k_ours, k_gold in ENERGY_KEYS:
        if rd_only and k_gold in ("es", "es_real", "es_recip", "es_self", "polar"):
            continue
        if wolf and k_gold in ("es_real", "es_recip", "es_self"):
            continue  # with wolf on, coulombic() is coulombic_wolf(); the harness' Ewald component columns are not part of it
        if not close(res[k_ours], g[k_gold], tol):
            bad.append(f"{k_gold}: ours {res[k_ours]!r} ref {g[k_gold]!r}")
    assert not bad, f"{label} energy mismatch (tol {tol}): " + "; ".join(bad)


def assert_counts(res, g, rd_only, label=""):
    keys = list(COUNT_KEYS) + ([] if rd_only else ["n_es_in_cutoff"])
    bad = [f"{k}: ours {res[k]} ref {g[k]}" for k in keys if int(res[k]) != int(g[k])]
    assert not bad, f"{label} pair-count mismatch (must be bit-exact): " + "; ".join(bad)


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max()
    return float(np.abs(a - b).max() / scale) if scale > 0 else float(np.abs(a).max())
