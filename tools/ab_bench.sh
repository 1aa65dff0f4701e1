#!/bin/bash
# same-box A/B of execution variants (MI355X boxes differ by up to ~10 %, so variants are only comparable inside one gpurun call)
# usage: tools/ab_bench.sh "<label>:<key=value,key=value>" ...   (keys of mpmc_debug_configure, passed as bench.py --configure)
#   e.g. tools/ab_bench.sh "default:" "fused:pair_kernel=1" "nopanels:panels=0";   MPMC_ENERGY_LIB=<other build> selects another library
mkdir -p gpurun_out
for rep in 1 2; do
for spec in "$@"; do
  label="${spec%%:*}"; cfg="${spec#*:}"
  flags=""; for kv in ${cfg//,/ }; do flags="$flags --configure $kv"; done
  for conc in async serial; do
    timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-baseline none --no-extra-passes --concurrency $conc $flags > gpurun_out/ab_${label}_${conc}_${rep}.json 2>/dev/null
    python - <<PY
import json
d=json.load(open("gpurun_out/ab_${label}_${conc}_${rep}.json"))
k=d["kernel_ms"]
print("rep${rep} ${label:-default} ${conc}: %.1f evals/s  pair %.3f  iter %.3f" % (d["value"], k.get("pair",0), k.get("dipole_iter",0)))
PY
  done
done
done
