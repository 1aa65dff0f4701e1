#!/bin/bash
# four torch-free ranks of bench.py sharing the one GPU (combine over the socket hub), and one process on the same ensemble: same potential, bit for bit
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root; mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --gpus 4 --combine-impl hub --force-device 0 --steps 5 --warmup 2 --cpu-baseline none > gpurun_out/r05_bench_four_ranks_hub.json 2> gpurun_out/r05_bench_four_ranks_hub.err; echo "four ranks rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-baseline none --no-extra-passes > gpurun_out/r05_bench_one_rank_short.json 2>/dev/null; echo "one rank rc=$?"
python3 - <<PY
import json
a=json.loads(open("gpurun_out/r05_bench_four_ranks_hub.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/r05_bench_one_rank_short.json").read().strip().splitlines()[-1])
print("four ranks:", a["value"], "evals/s, beads per rank", a["config"]["beads_per_gpu"], [(r["rank"], r["torch_imported"], len(r["rocm_libs"]["libamdhip64"])) for r in a["config"]["ranks"]])
print("one process:", b["value"], "| V_mean_K equal bit for bit:", a["V_mean_K"] == b["V_mean_K"], a["V_mean_K"])
PY
