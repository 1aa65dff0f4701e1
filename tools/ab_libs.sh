#!/bin/bash
# same-box A/B of two BUILDS of the library: tools/ab_libs.sh <label>=<path to .so> ...   (bench.py --steps 10, three times, interleaved).
# Prints the headline (32 beads in flight), the PCIe-inclusive rate, one system at a time for configs[2] / configs[3], and the two hot kernels
# alone on the GPU (back-to-back HIP events).  Older builds go under mpmcxx_amd/.abl/ (git-ignored, travels with gpurun).
mkdir -p gpurun_out
for rep in 1 2 3; do
for spec in "$@"; do
  label="${spec%%=*}"; lib="${spec#*=}"
  MPMC_ENERGY_LIB=$lib timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-baseline none > gpurun_out/abl_${label}_${rep}.json 2>gpurun_out/abl_${label}_${rep}.err || { echo "rep${rep} ${label}: bench failed"; tail -3 gpurun_out/abl_${label}_${rep}.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_${label}_${rep}.json"))
r=d["roofline"]
oc={o["workload"][:10]: o for o in (r.get("other_configs") or d.get("other_configs", []))}
pair=(r.get("other_kernels") or {}).get("pair", {})
print("rep${rep} %-10s: %.1f evals/s (pcie %.1f)  panel %.2f us (frac %.3f)  pair %.1f us (frac %.3f)  alone: cfg2 %.0f/s cfg3 %.0f/s  in flight: cfg2 %.0f/s" % (
    "${label}", d["value"], d.get("pcie_inclusive_value", 0), r["avg_launch_ms"] * 1e3, r["frac"], pair.get("avg_launch_ms", 0) * 1e3, pair.get("frac", 0),
    oc.get("configs[2]", {}).get("evals_per_s_alone", 0), oc.get("configs[3]", {}).get("evals_per_s_alone", 0), oc.get("configs[2]", {}).get("evals_per_s_in_flight", 0)))
PY
done
done
