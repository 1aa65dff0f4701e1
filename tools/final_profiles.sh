#!/bin/bash
# regenerates the judged artefacts of profiles/ in ONE gpurun call (same box for every number):
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh'   then   bash tools/final_profiles.sh --collect r01   (here, after the merge)
# 1. the driver's command (python bench.py)                                  -> final_bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no CPU leg)       -> final_bench_default_kernel_stats.csv + the JSON line
# 3. tools/profile.sh: one bead at a time, --stats and the separate PMC passes -> final_serial_kernel_stats.csv, final_pmc_serial_summary.md
# Every run writes into directories of its own (run id = start time), so earlier runs kept in gpurun_out/ never mix in.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
if [ "$1" = "--collect" ]; then
	tag=${2:-r01}
	id=$(cat $out/final_profiles.id)
	cp $out/final_bench_default.json profiles/${tag}_final_bench_default.json
	cp $out/final_bench_under_rocprof.json profiles/${tag}_final_bench_default_under_rocprof.json
	cp "$(find $out/prof_default_$id -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_bench_default_kernel_stats.csv
	cp "$(find $out/prof_final$id/stats -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_serial_kernel_stats.csv
	python3 tools/summarize_pmc.py $out/prof_final$id > profiles/${tag}_final_pmc_serial_summary.md
	exit 0
fi
mkdir -p $out
id=$(date +%s)
echo $id > $out/final_profiles.id
cd $root && timeout -k 10 400 python3 bench.py > $out/final_bench_default.json 2> $out/final_bench_default.err || { echo "bench failed"; exit 1; }
tail -c 600 $out/final_bench_default.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default_$id -- python3 $root/bench.py --cpu-baseline none > $out/final_bench_under_rocprof.json 2> $out/final_bench_under_rocprof.err) || { echo "rocprof default failed"; exit 1; }
cd $root && bash tools/profile.sh final$id
