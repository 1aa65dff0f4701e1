#!/bin/bash
# regenerates the judged artefacts of profiles/ in ONE gpurun call (same box for every number):
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh'   then   bash tools/final_profiles.sh --collect r03   (here, after the merge)
# 1. the driver's command (python bench.py)                                       -> <tag>_final_bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no CPU leg)            -> <tag>_final_bench_default_kernel_stats.csv + the JSON line
# 3. the 4-beads-per-GPU rehearsal of the 8-GPU run                               -> <tag>_final_bench_rehearsal_4_beads.json
# 4. tools/pmc_stalls.sh: ONE bead, one stream -- rocprofv3's kernel trace of every kernel alone on the GPU, the PMC groups in separate
#    passes (instruction mix, waits, LDS, FETCH_SIZE, WRITE_SIZE)                 -> <tag>_final_serial_kernel_stats.csv, <tag>_pmc_stalls.txt, <tag>_traffic.json
# Every run writes into directories of its own (run id = start time), so earlier runs kept in gpurun_out/ never mix in.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
if [ "$1" = "--collect" ]; then
	tag=${2:-r03}
	id=$(cat $out/final_profiles.id)
	cp $out/final_bench_default.json profiles/${tag}_final_bench_default.json
	cp $out/final_bench_under_rocprof.json profiles/${tag}_final_bench_default_under_rocprof.json
	cp $out/final_bench_rehearsal4.json profiles/${tag}_final_bench_rehearsal_4_beads.json
	cp "$(find $out/prof_default_$id -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_bench_default_kernel_stats.csv
	cp "$(find $out/pmc_stalls_final$id/st -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_serial_kernel_stats.csv
	cp $out/pmc_stalls_final$id/summary.txt profiles/${tag}_pmc_stalls.txt
	cp $out/pmc_stalls_final$id/traffic.json profiles/${tag}_traffic.json
	exit 0
fi
mkdir -p $out
id=$(date +%s)
echo $id > $out/final_profiles.id
cd $root && timeout -k 10 400 python3 bench.py > $out/final_bench_default.json 2> $out/final_bench_default.err || { echo "bench failed"; exit 1; }
tail -c 400 $out/final_bench_default.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default_$id -- python3 $root/bench.py --cpu-baseline none > $out/final_bench_under_rocprof.json 2> $out/final_bench_under_rocprof.err) || { echo "rocprof default failed"; exit 1; }
cd $root && timeout -k 10 200 python3 bench.py --cpu-baseline none --no-extra-passes --beads-per-gpu-rehearsal 4 --steps 20 > $out/final_bench_rehearsal4.json 2> $out/final_bench_rehearsal4.err || { echo "rehearsal failed"; exit 1; }
cd $root && TAG=final$id bash tools/pmc_stalls.sh > /dev/null 2>&1
grep "^trace" $out/pmc_stalls_final$id/summary.txt
