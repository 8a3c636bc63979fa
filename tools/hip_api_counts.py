"""HIP runtime API calls per step from a rocprofv3 --hip-trace results .db: which synchronisation calls the step makes (DESIGN.md
section 7: what PyTorch's engine inserts on the "AccumulateGrad node's stream does not match" path).  usage: hip_api_counts.py results.db steps"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
views = [r[0] for r in db.execute("select name from sqlite_master where type = 'view'")]
src = "regions" if "regions" in views else None
if src is None:
    print("no regions view; views:", views); sys.exit(1)
cols = [c[1] for c in db.execute(f"pragma table_info('{src}')")]
rows = db.execute(f"select name, count(*), sum(end - start) from {src} group by name order by 2 desc").fetchall()
print(f"HIP API calls over the whole run (/{steps:g} steps incl. warm-up and set-up), {len(rows)} distinct:")
for n, c, t in rows:
    if any(k in n for k in ("Synchronize", "WaitEvent", "EventRecord", "EventQuery", "LaunchKernel", "Memcpy", "Memset", "StreamCreate", "EventCreate", "Malloc", "ModuleLaunch", "ExtLaunch")):
        print(f"   {n:44s} {c:8d}  ({c / steps:8.1f} per step)  {t / 1e6:9.2f} ms in the call")
