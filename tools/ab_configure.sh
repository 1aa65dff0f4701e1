#!/bin/bash
# same-box A/B of bench.py under lists of measurement switches: tools/ab_configure.sh "label:key=v key=v" ...  (three interleaved repeats, --steps 20)
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  label="${spec%%:*}"; kvs="${spec#*:}"; args=""
  for kv in $kvs; do args="$args --configure $kv"; done
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-baseline none --no-other-configs $args > gpurun_out/abc_${label}_${rep}.json 2>gpurun_out/abc_${label}_${rep}.err || { echo "rep${rep} ${label}: bench failed"; tail -3 gpurun_out/abc_${label}_${rep}.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abc_${label}_${rep}.json"))
r=d["roofline"]
print("rep${rep} %-14s: %.1f evals/s (pcie %.1f, four beads %.1f)  panel alone %.2f us  in-flight kernel ms %s" % ("${label}", d["value"], d.get("pcie_inclusive_value", 0),
      (r.get("four_beads_in_flight") or {}).get("evals_per_s", 0), r["avg_launch_ms"] * 1e3, {k: round(v, 4) for k, v in r["in_timed_region"]["kernel_ms"].items() if k in ("dipole_iter", "reduce", "pair")}))
PY
done
done
