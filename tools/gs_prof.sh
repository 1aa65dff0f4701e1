#!/bin/bash
# rocprofv3 kernel statistics of Gauss-Seidel evaluations of the 10 000-atom box: bash tools/gs_prof.sh   (through gpurun, from the repo root)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/gs_prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/gs_trace.py > $out/run.log 2> $out/run.err || { echo "rocprof failed"; tail -n 5 $out/run.err; exit 1; }
tail -n 1 $out/run.log
f=$(find $out -name '*kernel_stats.csv' | head -n 1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; exit 1; }
cp "$f" $root/gpurun_out/gs_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:8]:
    print("  %-34s calls %6s  avg %8.2f us  %5.1f %%" % (r["Name"].split("(")[0][-34:], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * int(r["TotalDurationNs"]) / tot))
PY
