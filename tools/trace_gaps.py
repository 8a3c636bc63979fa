"""dev aid: GPU idle time inside the timed steps of bench.py from a rocprofv3 --kernel-trace CSV (x_kernel_trace.csv): the union of all
kernel intervals (any stream) against the wall time they span, and the largest gaps with the kernels on either side.
usage: python tools/trace_gaps.py x_kernel_trace.csv [n_last_steps=4]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# take the last 60 % of the trace (warm-up and CPU baseline excluded)
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * 0.4
rows = [r for r in rows if r[0] >= t0]
busy, gaps = 0, []
cur_s, cur_e, last = rows[0][0], rows[0][1], rows[0][2]
for s, e, n, q in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last = n if e >= cur_e else last
busy += cur_e - cur_s
wall = rows[-1][1] - rows[0][0]
print(f"wall {wall / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {100 * (1 - busy / wall):.1f} %, {len(rows)} kernels, {len(gaps)} gaps")
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else ">=20us"] += g
print({k: f"{v / 1e6:.2f} ms" for k, v in hist.items()})
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  gap {g / 1e3:8.1f} us   after {a}   before {b}")
