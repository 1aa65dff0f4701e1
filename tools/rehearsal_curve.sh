#!/bin/bash
# The per-GPU load of an N-GPU run of the 32-bead ensemble on ONE GPU: 32 / 16 / 8 / 4 / 2 / 1 beads in flight, the timed region without
# instrumentation (default since round 4) and with the per-launch HIP events of rounds 1-3 on bead 0's stream (--events-in-timed-region).
#   gpurun --timeout 900 -- 'bash tools/rehearsal_curve.sh > gpurun_out/rehearsal_curve.txt'
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
echo "beads_in_flight  evals/s(no events)  evals/s(events on bead 0)  ratio_to_32(no events)  ratio_to_32(events)"
base0=""; base1=""
for b in 32 16 8 4 2 1; do
	steps=$((640 / b)); [ $steps -gt 160 ] && steps=160
	v0=$(timeout -k 10 200 python3 bench.py --cpu-baseline none --no-extra-passes --beads-per-gpu-rehearsal $b --steps $steps --warmup 3 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])") || exit 1
	v1=$(timeout -k 10 200 python3 bench.py --cpu-baseline none --no-extra-passes --events-in-timed-region --beads-per-gpu-rehearsal $b --steps $steps --warmup 3 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])") || exit 1
	[ -z "$base0" ] && base0=$v0 && base1=$v1
	python3 -c "print(f'{$b:>3d}  {$v0:9.1f}  {$v1:9.1f}  {$v0/$base0:6.3f}  {$v1/$base1:6.3f}')"
done
