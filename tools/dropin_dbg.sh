#!/bin/bash
# stress of the drop-in binary in MPMC_WRAP_MODE=both (every call checked against the reference): repeats the 1000-ion PI case
# usage (GPU box): bash tools/dropin_dbg.sh [repeats] [threads]
root=${GRAFT_REPO_ROOT:-$(pwd)}
n=${1:-30}; thr=${2:-4}
bad=0
for i in $(seq 1 $n); do
  d=$(mktemp -d)
  cp $root/tests/golden/pi_ion1000/input.in $root/tests/golden/pi_ion1000/*.pqr $d/
  (cd $d && MPMC_WRAP_MODE=both OMP_NUM_THREADS=$thr timeout 300 $root/oracle/_ref/mpmcxx_wrapped -P 4 input.in > out.log 2> err.log)
  rc=$?
  [ $((i % 10)) -eq 0 ] && echo "  ... $i runs, $bad failures"
  if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "run $i rc=$rc"; grep -h "ref_adapter" $d/err.log | cut -c1-300; fi
done
echo "runs $n, failures $bad"
