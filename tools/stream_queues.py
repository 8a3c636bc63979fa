"""which hardware queue each HIP stream's kernels went to, from a rocprofv3 --kernel-trace results .db (the
stream-count cliff).  usage: python3 tools/stream_queues.py results.db [label]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = db.execute("select stream_id, queue_id, count(*), sum(end - start), min(start), max(end) from kernels group by stream_id, queue_id order by 1, 2").fetchall()
t0 = min(r[4] for r in rows); t1 = max(r[5] for r in rows)
print(f"{label}: {len(set(r[0] for r in rows))} streams with kernels on {len(set(r[1] for r in rows))} hardware queues, trace span {(t1 - t0) / 1e6:.1f} ms")
for sid, qid, n, busy, a, b in rows:
    top = db.execute("select name, count(*) from kernels where stream_id = ? and queue_id = ? group by name order by sum(end - start) desc limit 2", (sid, qid)).fetchall()
    names = "; ".join(f"{nm[:48]} x{c}" for nm, c in top)
    print(f"   stream {sid:3d} -> queue {qid:3d}: {n:6d} kernels, {busy / 1e6:8.1f} ms busy   [{names}]")
# overlap between queues: time during which kernels of two different queues were both running (sweep)
ev = []
for qid, st, en in db.execute("select queue_id, start, end from kernels"):
    ev.append((st, 1, qid)); ev.append((en, -1, qid))
ev.sort()
active, last, both = {}, None, 0
for t, d, q in ev:
    if last is not None and sum(1 for v in active.values() if v > 0) >= 2:
        both += t - last
    active[q] = active.get(q, 0) + d
    last = t
print(f"   kernels of two or more queues in flight at once: {both / 1e6:.1f} ms of the span")
