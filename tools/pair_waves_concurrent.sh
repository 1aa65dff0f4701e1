#!/bin/bash
# throughput regime (32 beads in flight, bench.py) of the pair sweep's one-wave and four-wave forms at the sizes around the threshold
#   gpurun --timeout 900 -- 'bash tools/pair_waves_concurrent.sh'
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out; mkdir -p $out; cd $root
for n in 3000 5000 7000 10000; do for w in 1 4 1 4; do
	timeout -k 10 200 python bench.py --natoms $n --cpu-baseline none --no-extra-passes --configure pair_kernel=1 --configure pair_waves=$w > $out/pwc.json 2> $out/pwc.err || { echo "bench failed n=$n w=$w"; tail -5 $out/pwc.err; exit 1; }
	python -c "
import json; b=json.loads(open('$out/pwc.json').read().strip().splitlines()[-1]); print('natoms $n  W=$w  %.1f evals/s' % b['value'])"
done; done
