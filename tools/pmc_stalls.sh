#!/bin/bash
# where the waves of the Jacobi / pair kernels spend their cycles: wait, VALU, LDS, VMEM, instruction fetch (two PMC passes, one bead)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_stalls${TAG:+_$TAG}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 $root/tools/kernel_ab.py "hyb:" > $out/p1.log 2> $out/p1.err || echo "pass 1 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/p2 -- python3 $root/tools/kernel_ab.py "hyb:" > $out/p2.log 2> $out/p2.err || echo "pass 2 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $out/p3 -- python3 $root/tools/kernel_ab.py "hyb:" > $out/p3.log 2> $out/p3.err || echo "pass 3 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 $root/tools/kernel_ab.py "hyb:" > $out/st.log 2> $out/st.err || echo "stats pass failed"
python3 - <<PY
import csv, glob, collections
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0].replace("void mpmc::","")
        if "pair_" in n or "panel" in n: print(f"trace: {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void mpmc::","")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in rows.items():
    if "hybrid" in k or "pair_fused" in k or "pair_sweep" in k or "panel" in k:
        m={c: sum(x)/len(x) for c,x in v.items()}
        wc=m.get("SQ_WAVE_CYCLES",1)
        print(k[:50])
        for c,x in sorted(m.items()):
            print(f"   {c:28s} {x:12.4g}  {x/wc:8.3f} of wave cycles")
PY
