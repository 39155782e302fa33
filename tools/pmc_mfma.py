#!/usr/bin/env python3
"""pmc_mfma.py DIR — matrix-core occupancy per kernel from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv`:
busy = SQ_VALU_MFMA_BUSY_CYCLES per launch (summed over the chip's 1024 SIMDs; 32 per 32x32x16 bf16 MFMA of one wave, MI355X_MICROARCH.md), clk = GRBM_GUI_ACTIVE / 8
(rocprofv3 sums the 8 XCDs) = the launch's length in shader clocks; share = busy / (1024 * clk). Only kernels with > 0.5 % of all busy cycles are listed."""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("mi355x::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen: seen.add(key); calls[k] += 1
tot = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in agg.values()) or 1.0
out = []
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if busy < 0.005*tot: continue
    n = calls[k]
    out.append({"kernel": k, "launches": n, "mfma_busy_cycles_per_launch": round(busy/n), "clocks_per_launch": round(gui/8/n),
                "mfma_busy_share": round(busy/(1024.0*gui/8), 4) if gui else None})
print(json.dumps(out, indent=1))
