#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile.sh: per kernel name, mean duration and mean counter values
per dispatch.  usage: tools/summarize_pmc.py gpurun_out/prof_<tag> > profiles/<name>.md"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name.replace("void mpmc::", "").replace("mpmc::", "")


def main(root):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = defaultdict(lambda: defaultdict(list))
    for sub in ("sq", "fetch", "write"):
        for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted(dur, key=lambda k: -sum(dur[k]))
    cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
            "FETCH_SIZE", "WRITE_SIZE"]
    print("| kernel | calls | mean us | total ms | " + " | ".join(cols) + " |")
    print("|---|---|---|---|" + "---|" * len(cols))
    for k in names:
        row = [k, str(len(dur[k])), f"{sum(dur[k]) / len(dur[k]):.1f}", f"{sum(dur[k]) / 1e3:.2f}"]
        for c in cols:
            v = ctr[k].get(c)
            row.append(f"{sum(v) / len(v):.4g}" if v else "")
        print("| " + " | ".join(row) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
