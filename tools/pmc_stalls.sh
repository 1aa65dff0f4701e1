#!/bin/bash
# Per-launch counters of the two dominant kernels (Jacobi contraction, pair sweep), ONE bead of the 10 000-atom box, one stream (each kernel
# alone on the GPU): where the waves spend their cycles, the instruction mix, LDS conflicts, HBM bytes, and rocprofv3's own kernel trace.
# Every --pmc group is its own pass with --kernel-trace only (gpurun refuses --pmc combined with other trace domains); FETCH_SIZE and
# WRITE_SIZE in separate passes, FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md "HBM").
#   gpurun --timeout 900 -- 'TAG=r03 bash tools/pmc_stalls.sh'      -> gpurun_out/pmc_stalls_r03/{summary.txt,traffic.json}
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_stalls${TAG:+_$TAG}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
	local name=$1; shift
	timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 $root/tools/kernel_ab.py "prod:" > $out/$name.log 2> $out/$name.err || echo "pass $name failed"
}
run p1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run p2 SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_ACTIVE_INST_SCA
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run p4 FETCH_SIZE
run p5 WRITE_SIZE
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 $root/tools/kernel_ab.py "prod:" > $out/st.log 2> $out/st.err || echo "stats pass failed"
python3 - > $out/summary.txt <<PY
import csv, glob, collections, json
def short(n): return n.split("(")[0].replace("void mpmc::","").replace("mpmc::","")
want = ("k_dipole_iter", "k_pair_", "k_dipole_update", "k_build_panels")
trace = {}
for f in glob.glob("$out/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Name"])
        if n.startswith(want):
            trace[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3)
            print(f"trace: {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {}
for k, v in rows.items():
    if not k.startswith(want): continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k[:60])
    for c, x in sorted(m.items()):
        print(f"   {c:28s} {x:12.4g}  {x/wc:8.3f} of wave cycles")
    if k.startswith(("k_dipole_iter_panel", "k_pair_sweep", "k_pair_fused")) and "SQ_INSTS_VALU" in m:
        fl = 64.0 * (2 * m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0))
        e = {"natoms": 10000, "valu_wave_insts_per_launch": m["SQ_INSTS_VALU"], "fma_f64": m.get("SQ_INSTS_VALU_FMA_F64"), "mul_f64": m.get("SQ_INSTS_VALU_MUL_F64"),
             "add_f64": m.get("SQ_INSTS_VALU_ADD_F64"), "trans_f64": m.get("SQ_INSTS_VALU_TRANS_F64"),
             "executed_flops_per_launch": fl, "executed_flops_what": "64 lanes x (2 FMA + MUL + ADD + TRANS) fp64 wave-instructions (PMC SQ_INSTS_VALU_*_F64; masked lanes count as executed)",
             "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": m.get("SQ_LDS_IDX_ACTIVE")}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e.update(FETCH_SIZE_KB=m["FETCH_SIZE"], WRITE_SIZE_KB=m["WRITE_SIZE"], fetch_correction=2.0, hbm_bytes_per_launch=(2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0)
        if k in trace:
            e.update(trace_calls=trace[k][0], trace_avg_launch_ms=trace[k][1] / 1e3, trace_min_launch_ms=trace[k][2] / 1e3)
        e["source"] = "tools/pmc_stalls.sh: rocprofv3 --kernel-trace --pmc <group> in separate passes (FETCH_SIZE / WRITE_SIZE each on its own, FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md), --kernel-trace --stats for the durations; one bead, one stream, 10 000 atoms"
        traffic[k.split("<")[0]] = e
json.dump(traffic, open("$out/traffic.json", "w"), indent=1, sort_keys=True)
PY
cat $out/summary.txt
