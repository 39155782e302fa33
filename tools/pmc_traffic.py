#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc FETCH_SIZE --output-format csv` run into per-launch HBM read traffic for the dominant kernels.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE is in KiB units of 64-B requests and reports exactly 1/2 of the bytes
of a wide coalesced streaming read -> bytes = FETCH_SIZE * 1024 * 2."""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = (r["Kernel_Name"].replace("mi355x::", "")[:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
out = []
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    out.append({"kernel": k[0], "grid": k[1], "launches": n, "fetch_size_raw_per_launch": v / n,
                "hbm_read_bytes_per_launch_corrected": v / n * 1024 * 2})
print(json.dumps(out, indent=1))
