"""rocpd_stats.py — per-kernel totals from a rocprofv3 run's rocpd SQLite database (rocprofv3 --kernel-trace writes <prefix>_results.db on this
image; --output-format csv is not always there): the table rocprofv3 --stats would print.  python tools/rocpd_stats.py DB [out.csv] [min_start_frac]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if rows and frac > 0:
    t0, t1 = rows[0][1], rows[-1][2]
    rows = [r for r in rows if r[1] >= t0 + frac*(t1 - t0)]
agg = collections.OrderedDict()
for n, s, e in rows:
    a = agg.setdefault(n, [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
tot = sum(a[1] for a in agg.values())
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    lines.append(f'"{n}",{a[0]},{a[1]},{a[1]/a[0]:.1f},{100.0*a[1]/tot:.2f},{a[2]},{a[3]}')
if len(sys.argv) > 2 and sys.argv[2] != "-":
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for l in lines[:28]:
    print(l[:230])
