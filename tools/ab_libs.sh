#!/bin/bash
# same-box A/B of two BUILDS of the library: tools/ab_libs.sh <label>=<path to .so> ...   (bench.py --steps 5, 32 beads in flight and serial, twice, interleaved)
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  label="${spec%%=*}"; lib="${spec#*=}"
  MPMC_ENERGY_LIB=$lib timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none > gpurun_out/abl_${label}_${rep}.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_${label}_${rep}.json"))
r=d["roofline"]
print("rep${rep} ${label}: %.1f evals/s  panel %.4f ms back to back (frac %.3f)  E %.13e" % (d["value"], r["avg_launch_ms"], r["frac"], d.get("energy_bead0", float("nan"))))
PY
done
done
