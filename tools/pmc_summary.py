#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (tools/pmc_passes.sh output) into one table: counter sums per kernel
per dispatch (averaged over dispatches).  Usage: python tools/pmc_summary.py gpurun_out/<dir> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("mhk::"):
            continue
        c = row["Counter_Name"]
        acc[k][c] += float(row["Counter_Value"])
        ndisp[k][c].add(row["Dispatch_Id"])
out = {k: {c: v / max(len(ndisp[k][c]), 1) for c, v in cs.items()} for k, cs in acc.items()}
for k, cs in sorted(out.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s %18.0f" % (c, v))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
