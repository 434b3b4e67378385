"""Per-launch timeline of the last potrf call in a rocprofv3 rocpd database: duration of every
launch and the idle gap before it.   usage: python tools/rocpd_timeline.py results.db"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else cols[0]
rows = sorted(cur.execute(f"select start, end, {name_col} from kernels").fetchall())
# the last finalize kernel ends the last call; walk back to the previous finalize
fin = [i for i, r in enumerate(rows) if 'finalize' in r[2]]
lo = fin[-2] + 1 if len(fin) > 1 else 0
sel = [r for r in rows[lo:fin[-1] + 1] if 'potrf' in r[2]]
t0 = sel[0][0]
tot = {'syrk': 0, 'panel': 0, 'gap': 0, 'finalize': 0}
prev_end = None
for s, e, n in sel:
    k = 'syrk' if 'syrk' in n else 'panel' if 'panel' in n else 'finalize'
    gap = (s - prev_end) if prev_end is not None else 0
    tot[k] += e - s; tot['gap'] += max(gap, 0)
    print(f"{(s - t0) / 1e3:9.1f} us  {k:8s} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:6.1f}")
    prev_end = e
print({k: round(v / 1e3, 1) for k, v in tot.items()}, "total", round((sel[-1][1] - t0) / 1e3, 1))
