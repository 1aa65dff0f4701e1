#!/bin/bash
# samples GPU clocks / power (rocm-smi) while the benchmark runs: tells whether fp64-dense kernels hold the rated clock
# usage (through gpurun): tools/clock_probe.sh [bench args...]
mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showtemp > gpurun_out/smi_idle.txt 2>&1
python bench.py --steps 40 --warmup 2 --cpu-baseline none "$@" > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
pid=$!
sleep 14
for k in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp > gpurun_out/smi_load_$k.txt 2>&1
  sleep 0.7
done
wait $pid
grep -h -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" gpurun_out/smi_idle.txt | sed 's/^/idle: /'
for k in 1 2 3 4 5 6; do grep -h -E "sclk|Power \(|Average|Current Socket|junction|hotspot" gpurun_out/smi_load_$k.txt | sed "s/^/load$k: /"; done
python - <<'PY'
import json
d=json.load(open("gpurun_out/clock_probe_bench.json"))
print("bench:", d["value"], d["unit"], d["ms_per_step"], "ms/step")
PY
