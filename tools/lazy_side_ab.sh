#!/bin/bash
# side stream created at first fork (lazy_side_stream=1, every main stream gets a hardware queue of its own in turn) x GPU_MAX_HW_QUEUES:
# rate of the headline job, hardware queues used and how many kernels run at a time (rocprofv3 kernel trace of a shorter run)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_lazy_side_queues.txt; : > $out
for rep in 1 2; do
 for q in 4 8 6 3 2 12; do
  r=$(GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --steps 15 --warmup 3 --cpu-baseline none --no-extra-passes --no-other-configs --configure lazy_side_stream=1 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "rep$rep lazy_side_stream=1 GPU_MAX_HW_QUEUES=$q: $r evals/s" | tee -a $out
 done
done
for q in 8 2; do
export GPU_MAX_HW_QUEUES=$q
rm -rf gpurun_out/prof_lazy
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lazy -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline none --no-extra-passes --no-other-configs --configure lazy_side_stream=1 > /dev/null 2>&1)
python3 - <<PY | tee -a gpurun_out/r04_lazy_side_queues.txt
import csv, glob, collections
f = glob.glob('gpurun_out/prof_lazy/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
q = collections.Counter(r['Queue_Id'] for r in rows)
print('GPU_MAX_HW_QUEUES=$q dispatches per queue:', dict(q))
ev = []
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows); rows = rows[n//2:]
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
lvl = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print('  kernels running at a time (second half of the trace):', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
PY
done
