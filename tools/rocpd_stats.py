"""rocpd_stats.py — per-kernel totals from the rocpd sqlite database rocprofv3 writes (`rocprofv3 --kernel-trace -d DIR -- ...`).
usage: python tools/rocpd_stats.py DIR_OR_DB [out.csv]"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
dbs = [src] if os.path.isfile(src) else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
rows = {}
for db in dbs:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    for name, n, tot, mn, mx in con.execute(
            f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name"):
        r = rows.setdefault(name, [0, 0, 1 << 62, 0]); r[0] += n; r[1] += tot; r[2] = min(r[2], mn); r[3] = max(r[3], mx)
total = sum(r[1] for r in rows.values())
lines = ["name,calls,total_ns,avg_ns,pct,min_ns,max_ns"]
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{name}",{r[0]},{r[1]},{r[1]/r[0]:.0f},{100.0*r[1]/total:.2f},{r[2]},{r[3]}')
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print("\n".join(l[:230] for l in lines[:25]))
