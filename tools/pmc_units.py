#!/usr/bin/env python3
"""Summarise `rocprofv3 --pmc <counters> --output-format csv -d DIR` runs: per kernel, the mean of each counter per launch.
usage: python tools/pmc_units.py DIR [kernel-substring]"""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "k_mmq"
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].replace("mi355x::", "")[:48], r.get("Grid_Size", ""))
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k, json.dumps({c: round(v[1] / v[0], 1) for c, v in sorted(cs.items())}))
