#!/bin/bash
# rocprofv3 passes of the bench workload on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats            per-kernel durations
#   2. --pmc <SQ counters>               VALU / LDS / wait mix           (own pass, kernel-trace only)
#   3. --pmc FETCH_SIZE, 4. --pmc WRITE_SIZE   HBM traffic (separate passes: TCC slots; FETCH_SIZE reads 1/2 of wide
#      coalesced reads on gfx950, MI355X_MICROARCH.md "HBM")
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r01}; shift
args=${@:---steps 2 --warmup 1 --beads 4 --concurrency serial --cpu-baseline none}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py $args > $out/stats.json 2> $out/stats.err || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/sq -- python3 $root/bench.py $args > $out/sq.json 2> $out/sq.err || echo "sq pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py $args > $out/fetch.json 2> $out/fetch.err || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py $args > $out/write.json 2> $out/write.err || echo "write pass failed"
find $out -name "*.csv" | head -20
