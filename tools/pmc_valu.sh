#!/bin/bash
# instruction counts per launch of the pair sweep for a list of variants ("label:key=v,..."; MPMC_ENERGY_LIB selects another build):
#   gpurun -- 'bash tools/pmc_valu.sh "split:pair_split=1" "nosplit:pair_split=0"'
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_valu
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
	label="${spec%%:*}"
	KAB_REPS=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $out/$label -- python3 $root/tools/kernel_ab.py "$spec" > $out/$label.log 2> $out/$label.err || echo "pass $label failed"
	python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/$label/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void mpmc::", "").replace("mpmc::", "")
        if n.startswith(("k_pair_", "k_dipole_iter_panel", "k_dipole_update_panel")):
            rows[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, m in rows.items():
    a = {c: sum(x) / len(x) for c, x in m.items()}
    print("$label %-42s VALU %.4g  SALU %.4g  waves %.0f  FMA %.4g MUL %.4g ADD %.4g  LDS %.4g  busy %.4g  VALU/wave %.0f" % (
        n[:42], a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_SALU", 0), a.get("SQ_WAVES", 0), a.get("SQ_INSTS_VALU_FMA_F64", 0), a.get("SQ_INSTS_VALU_MUL_F64", 0),
        a.get("SQ_INSTS_VALU_ADD_F64", 0), a.get("SQ_INSTS_LDS", 0), a.get("SQ_BUSY_CU_CYCLES", 0), a.get("SQ_INSTS_VALU", 0) / max(a.get("SQ_WAVES", 1), 1)))
PY
done
