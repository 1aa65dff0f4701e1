#!/bin/bash
# regenerates the judged artefacts of profiles/ in ONE gpurun call (same box for every number) -- run ONCE per round, on the final build:
#   gpurun --timeout 1200 -- 'bash tools/final_profiles.sh'   then   bash tools/final_profiles.sh --collect r05   (here, after the merge)
# 1. the driver's command (python bench.py)                                       -> <tag>_final_bench_default.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no CPU leg)            -> <tag>_final_bench_default_kernel_stats.csv + the JSON line
# 3. tools/rehearsal_curve.sh: 32 .. 1 beads in flight, with and without events   -> <tag>_rehearsal_beads_per_gpu.txt
# 4. tools/pmc_stalls.sh: ONE bead, one stream -- rocprofv3's kernel trace of every kernel alone on the GPU, the PMC groups in separate
#    passes (instruction mix, waits, LDS, FETCH_SIZE, WRITE_SIZE)                 -> <tag>_final_serial_kernel_stats.csv, <tag>_pmc_stalls.txt, <tag>_traffic.json
# 5. tools/config_rates.py (BASELINE configs[1..3]), tools/solver_time.py, the launch list of one evaluation (tools/small_trace.py)
# 6. bench.py --solver dense (4 beads), and the two multi-rank launch modes rehearsed on the one GPU (gloo ranks / --launch inprocess)
# Every run writes into directories of its own (run id = start time), so earlier runs kept in gpurun_out/ never mix in.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
if [ "$1" = "--collect" ]; then
	tag=${2:-r05}
	id=$(cat $out/final_profiles.id)
	cp $out/final_bench_default.json profiles/${tag}_final_bench_default.json
	cp $out/final_bench_under_rocprof.json profiles/${tag}_final_bench_default_under_rocprof.json
	cp "$(find $out/prof_default_$id -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_bench_default_kernel_stats.csv
	cp "$(find $out/pmc_stalls_final$id/st -name '*kernel_stats.csv' | head -1)" profiles/${tag}_final_serial_kernel_stats.csv
	cp $out/pmc_stalls_final$id/summary.txt profiles/${tag}_pmc_stalls.txt
	cp $out/pmc_stalls_final$id/traffic.json profiles/${tag}_traffic.json
	cp $out/final_rehearsal_curve.txt profiles/${tag}_rehearsal_beads_per_gpu.txt
	cp $out/final_config_rates.txt profiles/${tag}_config_rates.txt
	cp $out/final_solver_time.txt profiles/${tag}_solver_time.txt
	cp $out/final_trace10k.txt profiles/${tag}_one_evaluation_timeline.txt
	cp $out/final_bench_dense.json profiles/${tag}_bench_dense_solver.json
	cp $out/final_bench_hub2.json profiles/${tag}_bench_two_ranks_one_gpu_hub.json
	cp $out/final_bench_cabi2.json profiles/${tag}_bench_two_ranks_one_gpu_cabi_fallback.json
	cp $out/final_bench_torchrun2.json profiles/${tag}_bench_two_ranks_one_gpu_under_torchrun.json
	cp $out/final_alone.txt profiles/${tag}_alone.txt
	cp $out/final_sweep_replicas.txt profiles/${tag}_sweep_replicas.txt
	cp $out/final_bench_inprocess2.json profiles/${tag}_bench_inprocess_two_device_slots.json
	exit 0
fi
mkdir -p $out
id=$(date +%s)
echo $id > $out/final_profiles.id
cd $root && timeout -k 10 400 python3 bench.py > $out/final_bench_default.json 2> $out/final_bench_default.err || { echo "bench failed"; exit 1; }
tail -c 300 $out/final_bench_default.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_default_$id -- python3 $root/bench.py --cpu-baseline none > $out/final_bench_under_rocprof.json 2> $out/final_bench_under_rocprof.err) || { echo "rocprof default failed"; exit 1; }
echo "rehearsal curve"; cd $root && bash tools/rehearsal_curve.sh > $out/final_rehearsal_curve.txt 2>&1 || { echo "rehearsal failed"; exit 1; }
cat $out/final_rehearsal_curve.txt
echo "pmc"; cd $root && TAG=final$id bash tools/pmc_stalls.sh > /dev/null 2>&1
grep "^trace" $out/pmc_stalls_final$id/summary.txt
echo "config rates"; cd $root && timeout -k 10 300 python3 tools/config_rates.py > $out/final_config_rates.txt 2>&1; cat $out/final_config_rates.txt
echo "solver time"; timeout -k 10 300 python3 tools/solver_time.py > $out/final_solver_time.txt 2>&1; cat $out/final_solver_time.txt
echo "timeline"; rm -rf $out/trace10k_$id; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace10k_$id -- python3 $root/tools/small_trace.py ion10k_polar 30 > $out/final_trace10k.log 2>&1)
cd $root && python3 tools/small_trace.py --report $out/trace10k_$id > $out/final_trace10k.txt 2>&1; head -3 $out/final_trace10k.txt
echo "dense"; timeout -k 10 300 python3 bench.py --solver dense --beads 4 --steps 3 --warmup 1 --cpu-baseline none --no-other-configs > $out/final_bench_dense.json 2> $out/final_bench_dense.err || echo "dense bench failed"
echo "two torch-free ranks on one GPU, started bare, combine over the socket hub"; timeout -k 10 300 python3 bench.py --gpus 2 --combine-impl hub --force-device 0 --steps 5 --warmup 2 --cpu-baseline none > $out/final_bench_hub2.json 2> $out/final_bench_hub2.err || echo "hub2 failed"
echo "the same with the default --combine-impl cabi: RCCL refuses two ranks on one device, every rank falls back together"; timeout -k 10 300 python3 bench.py --gpus 2 --force-device 0 --comm-init-timeout 60 --steps 5 --warmup 2 --cpu-baseline none > $out/final_bench_cabi2.json 2> $out/final_bench_cabi2.err || echo "cabi2 failed"
echo "the driver's form: two ranks under python -m torch.distributed.run"; timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --combine-impl hub --force-device 0 --steps 5 --warmup 2 --cpu-baseline none > $out/final_bench_torchrun2.json 2> $out/final_bench_torchrun2.err || echo "torchrun2 failed"
echo "one evaluation at a time"; timeout -k 10 300 python3 tools/alone_ab.py "default:" > $out/final_alone.txt 2>&1; cat $out/final_alone.txt
echo "sweep without fill and drain"; timeout -k 10 300 python3 tools/sweep_replicas.py > $out/final_sweep_replicas.txt 2>&1; cat $out/final_sweep_replicas.txt
echo "in-process, two device slots"; timeout -k 10 300 python3 bench.py --gpus 2 --launch inprocess --force-device 0 --steps 5 --warmup 2 --cpu-baseline none > $out/final_bench_inprocess2.json 2> $out/final_bench_inprocess2.err || echo "inprocess2 failed"
for f in final_bench_dense final_bench_hub2 final_bench_cabi2 final_bench_torchrun2 final_bench_inprocess2; do python3 -c "
import json,sys
d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['n_gpus'], d['config'].get('launch'), '|', d['config'].get('combine_impl'))"; done
