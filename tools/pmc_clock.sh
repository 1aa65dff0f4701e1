#!/bin/bash
# shader clock actually held under each kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration, one PMC pass
# usage (through gpurun): tools/pmc_clock.sh
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_clock
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $out/p1 -- python3 $root/tools/kernel_ab.py "split:MPMC_JACOBI=split,MPMC_ONE_STREAM=1" "hyb:" > $out/p1.log 2> $out/p1.err || echo "pass failed"
python3 - <<PY
import csv, glob, collections
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void mpmc::","")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"]=="GRBM_GUI_ACTIVE":
            rows[k]["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("dur_us",[0]))):
    m={c: sum(x)/len(x) for c,x in v.items()}
    if "dur_us" in m and m["dur_us"]>5:
        print(f"{k[:60]:60s} dur {m['dur_us']:8.1f} us  GUI_ACTIVE {m.get('GRBM_GUI_ACTIVE',0):.4g} -> {m.get('GRBM_GUI_ACTIVE',0)/m['dur_us']/1e3:.3f} GHz  " + "  ".join(f"{c} {x:.4g}" for c,x in m.items() if c not in ('dur_us','GRBM_GUI_ACTIVE')))
PY
