#!/bin/bash
# every fuzzer once, exit codes recorded (run through gpurun from the repo root): bash tools/fuzz_all.sh > gpurun_out/fuzz_all.log
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
rc_all=0
run() { echo "== $*"; "$@"; local rc=$?; echo "== exit code $rc"; [ $rc -ne 0 ] && rc_all=1; }
run timeout -k 10 240 python3 tools/fuzz.py 20000 1500
run timeout -k 10 240 python3 tools/fuzz_large.py 2000 120
run timeout -k 10 240 python3 tools/fuzz.py 40000 800 sweep
run timeout -k 10 240 python3 tools/fuzz_large.py 4000 100 sweep
run timeout -k 10 240 python3 tools/fuzz.py 60000 600 sweep split
run timeout -k 10 240 python3 tools/fuzz_large.py 6000 60 sweep split
run timeout -k 10 240 python3 tools/fuzz_large.py 7000 100 grid
run timeout -k 10 240 python3 tools/fuzz_large.py 7500 100 sweep grid
run timeout -k 10 240 python3 tools/fuzz_large.py 8000 100 fused
run timeout -k 10 240 python3 tools/fuzz.py 80000 400 fused
run timeout -k 10 200 python3 tools/fuzz_trial.py 3000 300
run timeout -k 10 240 python3 tools/fuzz_trial.py 5000 400 polar
run timeout -k 10 150 python3 tools/fuzz_state.py 3000 120 4
echo "== overall $rc_all"
exit $rc_all
