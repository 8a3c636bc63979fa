#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE CSVs of tools/pmc_step_kernels.py -> JSON: per kernel and shape, HBM bytes per launch
(2 * FETCH_SIZE KiB + WRITE_SIZE KiB: gfx950 counts half of a wide coalesced read, MI355X_MICROARCH.md) next to the algorithmic
bytes and their ratio.   usage: parse_pmc_kernels.py FETCH_DIR WRITE_DIR out.json"""
import collections, csv, glob, json, re, sys
fetch_dir, write_dir, out = sys.argv[1:4]
e = 2.0
T = {"text": 1536 * 512, "vision": 1536 * 196}
D = 768
# algorithmic bytes per launch as multiples of one [T, d] bf16 tensor, by (kernel substring, variant tag)
ALG = {
    "rmsnorm_fwd_packed": 2.0, "rmsnorm_stats": 1.0,
    "rmsnorm_fwd_pooled<POOL=2>": 2.0, "rmsnorm_fwd_pooled<POOL=2,stats>": 1.0, "rmsnorm_fwd_pooled<gelu_in>": 1.0,
    "attn_fwd": 4.0, "attn_bwd_dq": 5.0, "attn_bwd_dkv": 6.0, "attn_bwd1": 8.0,      # single pass: q|k|v, dO, O read once, dq|dk|dv written
}

def load(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    keep = ("rmsnorm", "attn_fwd_kernel", "attn_bwd")
    rows = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in keep):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows

fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
assert [r[1] for r in fe] == [r[1] for r in wr], "the two passes launched different kernel sequences"
# launches of one kernel name are split into classes of equal traffic (the text launches move 2.6x the bytes of the vision ones,
# and the variants of a template that the demangler prints alike -- with / without the residual gradient, with / without a stored
# activation -- differ by whole tensors): a class = launches within 6 % of each other
agg = collections.OrderedDict()
half = len(fe) // 2            # tools/pmc_step_kernels.py issues the whole text sequence first, then the same sequence at the vision shape
for idx, ((i, name, f), (_, _, w)) in enumerate(zip(fe, wr)):
    m = re.search(r"(rmsnorm_\w+?_kernel|attn_\w+?_kernel)(I[A-Za-z0-9_]*?E(?=v|E)|<[^(]*>)?", name)
    key = (m.group(0) if m else name)[:90]
    agg.setdefault((key, "text" if idx < half else "vision"), []).append((2 * f * 1024, w * 1024))
res = collections.OrderedDict()
for (key, tag), ls in agg.items():
    classes = []
    for f, w in ls:
        for c in classes:
            if abs((f + w) - c["tot"]) <= 0.06 * c["tot"]:
                c["n"] += 1; c["f"] += f; c["w"] += w
                break
        else:
            classes.append({"tot": f + w, "n": 1, "f": f, "w": w})
    for c in sorted(classes, key=lambda c: -c["tot"]):
        f, w = c["f"] / c["n"], c["w"] / c["n"]
        res[f"{key} [{tag}, {round((f + w) / 1e9, 3)} GB]"] = {
            "launches": c["n"], "fetch_bytes_corrected": round(f), "write_bytes": round(w), "hbm_bytes": round(f + w),
            "in_units_of_one_token_tensor": round((f + w) / (T[tag] * D * e), 3),
            "read_tensors": round(f / (T[tag] * D * e), 3), "written_tensors": round(w / (T[tag] * D * e), 3)}
import hashlib, os
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "meant_amd", "csrc")
_sha = {f: hashlib.sha256(open(os.path.join(_csrc, f), "rb").read()).hexdigest()[:16] for f in ("norm.hip", "attn_bf16.hip", "attn_bwd1.hip")}
json.dump({"source_sha16": _sha,      # bench.py flags the table as older than the kernels when a source has moved on since
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_step_kernels.py; bytes = 2 * FETCH_SIZE "
                     "* 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads); one token tensor = T x 768 bf16 "
                     "(text T = 786432: 1.208 GB; vision T = 301056: 0.462 GB)", "kernels": res}, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
