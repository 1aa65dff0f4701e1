#!/usr/bin/env python3
"""Generate tests/golden/*  from the REFERENCE's own object code (build container only).

For every named fixture in mpmcxx_amd/gen_box.py this script
  1. writes NAME.in / NAME.pqr (the reference's on-disk formats) into tests/golden/,
  2. runs oracle/_ref/ref_harness (reference System::energy(), compiled in place from /root/reference/src
     by `make -C oracle ref`) on them,
  3. stores the harness JSON (energies, predicate counts, E0, mu, A-matrix spot blocks at %.17g) as
     tests/golden/NAME.json.
The 10k-atom boxes are not stored as PQR text (regenerated deterministically by gen_box.fixture); only
their energies and the per-atom E0 / mu / E_ind of every 157th atom (64 atoms) are recorded.

Fixtures are DATA (inputs + expected outputs).  No reference source text is written anywhere.
Usage: python oracle/make_golden.py [--large] [names...]
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from mpmcxx_amd import gen_box  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run_harness(in_path: str, extra):
    wd = os.path.dirname(in_path)
    p = subprocess.run([HARNESS, os.path.basename(in_path)] + extra, cwd=wd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"ref_harness failed on {in_path}:\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
    # the JSON object is everything after the last line that starts with '{'
    txt = p.stdout
    start = txt.rfind("\n{")
    return json.loads(txt[start + 1:])


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    large = "--large" in sys.argv
    names = args or (gen_box.SMALL_FIXTURES + (gen_box.LARGE_FIXTURES if large else []))
    if not os.path.exists(HARNESS):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLDEN, exist_ok=True)
    for name in names:
        is_large = name in gen_box.LARGE_FIXTURES
        if is_large:
            wd = tempfile.mkdtemp(prefix="golden_")
        else:
            wd = GOLDEN
        inp, pqr = gen_box.materialize(name, wd)
        rows, basis, opts = gen_box.fixture(name)
        n = len(rows)
        spots = [f"0,1", f"1,0", f"0,{n - 1}", f"{n // 2},{n // 3}"] if n > 3 else ["0,1", "1,0"]
        # small boxes: every atom's E0 / mu / E_ind and the update_com + wrap_all state; 10k boxes: a 64-atom sample (every 157th atom)
        extra = ["--sample-atoms", "157" if n > 5000 else "61"] if is_large else ["--dump-atoms", "--dump-com"]
        if opts.get("polarization") == "on" and not is_large:
            extra += ["--amatrix"] + spots
        res = run_harness(inp, extra)
        res["fixture"] = name
        res["generator"] = "mpmcxx_amd/gen_box.py:fixture + oracle/_ref/ref_harness (reference System::energy)"
        with open(os.path.join(GOLDEN, f"{name}.json"), "w") as f:
            json.dump(res, f, indent=0, separators=(",", ":"))
            f.write("\n")
        print(f"{name}: n={res['natoms']} total={res['total']!r} rd={res['rd']!r} es={res['es']!r} pol={res['polar']!r}")
        if is_large:
            shutil.rmtree(wd, ignore_errors=True)
    # stray files the reference writes into cwd
    for junk in os.listdir(GOLDEN):
        if junk.endswith((".dat", ".last", ".traj", ".csv")) or junk.startswith("t."):
            os.remove(os.path.join(GOLDEN, junk))


if __name__ == "__main__":
    main()
