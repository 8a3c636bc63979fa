"""per-kernel duration summary of a rocprofv3 results .db (the default output of ROCm 7.2's rocprofv3): name, calls, total / mean /
min / max in microseconds, sorted by total.  usage: python3 tools/rocprof_stats.py results.db [csv-out]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for n, c, t, a, mn, mx in rows:
    lines.append(f'"{n}",{c},{t},{a:.1f},{mn},{mx},{100.0 * t / tot:.2f}')
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for n, c, t, a, mn, mx in rows[:40]:
    print(f"{t/1e3:10.1f} us {100.0*t/tot:5.1f}%  x{c:<5d} avg {a/1e3:9.1f} min {mn/1e3:9.1f} max {mx/1e3:9.1f}  {n[:90]}")
