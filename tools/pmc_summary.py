"""summarise rocprofv3 --pmc csv output: per kernel name, mean of each counter over dispatches (optionally only the dispatches with
the largest grid).  usage: python3 tools/pmc_summary.py DIR [DIR ...] [--filter substr]"""
import csv, sys, collections, glob, os
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if flt and flt not in k:
                continue
            acc[(k[:70], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(acc.items()):
    print(f"== {k}  grid={g}")
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
