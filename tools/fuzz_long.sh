#!/bin/bash
# a longer fuzz campaign with seeds of its own (through gpurun from the repo root): bash tools/fuzz_long.sh > gpurun_out/fuzz_long.log
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
rc_all=0
run() { echo "== $*"; "$@" | tail -n 1; local rc=${PIPESTATUS[0]}; echo "== exit code $rc"; [ $rc -ne 0 ] && rc_all=1; }
run timeout -k 10 280 python3 tools/fuzz.py 210000 3000
run timeout -k 10 280 python3 tools/fuzz.py 250000 2500 sweep
run timeout -k 10 280 python3 tools/fuzz_large.py 19000 200
run timeout -k 10 280 python3 tools/fuzz_large.py 19500 200 sweep
run timeout -k 10 280 python3 tools/fuzz_trial.py 19000 600
run timeout -k 10 280 python3 tools/fuzz_trial.py 19700 600 polar
run timeout -k 10 280 python3 tools/fuzz_large.py 29000 200 fused
run timeout -k 10 200 python3 tools/fuzz_state.py 19000 200 4
echo "== overall $rc_all"
exit $rc_all
