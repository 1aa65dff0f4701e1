#!/bin/bash
# same-box A/B of execution variants (MI355X boxes differ by up to ~10 %, so variants are only comparable inside one gpurun call)
# usage: tools/ab_bench.sh "<label>:<ENV=...>" ...   e.g. tools/ab_bench.sh "hybrid:MPMC_JACOBI=hybrid" "split:"
mkdir -p gpurun_out
for rep in 1 2; do
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  for conc in async serial; do
    env $envs timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --concurrency $conc > gpurun_out/ab_${label}_${conc}_${rep}.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_${label}_${conc}_${rep}.json"))
k=d["kernel_ms"]
print("rep${rep} ${label:-default} ${conc}: %.1f evals/s  pair %.3f  iter %.3f far %.3f" % (d["value"], k.get("pair",0), k.get("dipole_iter",0), k.get("dipole_far",0)))
PY
  done
done
done
