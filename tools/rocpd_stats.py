"""Kernel-stats summary (CSV) from a rocprofv3 rocpd sqlite database.
usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else cols[0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = re.sub(r'\(.*', '', n)
    a = agg.setdefault(n, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}')
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
