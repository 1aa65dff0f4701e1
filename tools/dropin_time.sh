#!/bin/bash
# wall time of the reference's own PI driver on the 1000-ion polarizable case: stock CPU energy() versus the wrapped binary on the HIP path
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
for mode in passthrough gpu; do
  d=$(mktemp -d); cp $root/tests/golden/pi_ion1000/input.in $root/tests/golden/pi_ion1000/ion1000.pqr $d/
  sed -i 's/numsteps 12/numsteps 60/; s/corrtime 4/corrtime 20/' $d/input.in
  s=$(date +%s.%N)
  (cd $d && MPMC_WRAP_MODE=$mode OMP_NUM_THREADS=4 $root/oracle/_ref/mpmcxx_wrapped -P 4 input.in > log.txt 2> err.txt)
  e=$(date +%s.%N)
  echo "$mode: $(python3 -c "print(round($e - $s, 2))") s for 60 PI-NVT steps (P = 4, 1000 polarizable ions); last row: $(tail -1 $d/ion1000.energy.dat | cut -c1-60); $(grep -h 'calls served' $d/err.txt)"
done
