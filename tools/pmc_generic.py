#!/usr/bin/env python3
"""pmc_generic.py DIR — per-kernel averages of whatever counters a `rocprofv3 --pmc ... --output-format csv` run collected (the top kernels by launches x value of the first counter)."""
import csv, glob, json, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("mi355x::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r.get("Dispatch_Id"))
out = []
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    n = len(calls[k]); out.append({"kernel": k, "launches": n, **{c: round(x/n, 1) for c, x in v.items()}})
print(json.dumps(out, indent=1))
