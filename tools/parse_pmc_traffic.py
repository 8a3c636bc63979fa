#!/usr/bin/env python3
"""Turn the two rocprofv3 counter CSVs of tools/pmc_nt256.py into profiles/<name>.json:
   shape "M,N,K" -> HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE reports half of a wide coalesced read; both in KiB)."""
import csv, glob, json, sys, collections
sys.path.insert(0, ".")
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
dram_dir = sys.argv[4] if len(sys.argv) > 4 else None      # optional third pass: TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
import importlib.util, os as _os
_spec = importlib.util.spec_from_file_location("pmc_nt256", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pmc_nt256.py"))
# (the launch list lives in pmc_nt256.py; importing it runs nothing: its launches sit under __main__)
def _launches():
    src = open(_spec.origin).read()
    ns = {}
    exec(src[src.index("LAUNCHES = ["):src.index("if __name__")], ns)
    return ns["LAUNCHES"]
LAUNCHES = _launches()
SHAPES = LAUNCHES

def per_dispatch(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    vals = []
    for r in csv.DictReader(open(f)):
        if ("gemm_bf16_nt256p_kernel" in r["Kernel_Name"] or "gemm_bf16_nt256s_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    return [v for _, v in vals]

fe, wr = per_dispatch(fetch_dir, "FETCH_SIZE"), per_dispatch(write_dir, "WRITE_SIZE")
assert len(fe) == len(wr) == 3 * len(SHAPES), (len(fe), len(wr))
rq = per_dispatch(dram_dir, "TCC_EA0_RDREQ_sum") if dram_dir else None
rd = per_dispatch(dram_dir, "TCC_EA0_RDREQ_DRAM_sum") if dram_dir else None
res = {}
for i, (M, N, K, how) in enumerate(SHAPES):
    f = sum(fe[3 * i:3 * i + 3]) / 3 * 1024 * 2
    w = sum(wr[3 * i:3 * i + 3]) / 3 * 1024
    alg = (M * K + N * K + M * N) * 2
    key = f"{M},{N},{K}" if (how == "rot" or N != 2304) else f"{M},{N},{K},plain_epilogue"   # "M,N,K": as the step launches that shape
    res[key] = {"hbm_bytes": f + w, "fetch_bytes_corrected": f, "write_bytes": w, "algorithmic_bytes": alg,
                           "ratio": round((f + w) / alg, 3)}
    if rq and rd and len(rq) == len(fe):
        # share of the L2's read requests that the fabric sent on to DRAM (the rest were Infinity-Cache hits): FETCH_SIZE
        # counts both (MI355X_MICROARCH.md, HBM section)
        q, dd = sum(rq[3 * i:3 * i + 3]), sum(rd[3 * i:3 * i + 3])
        res[key]["read_requests_to_dram_share"] = round(dd / q, 3) if q else None
import hashlib, os
_src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "meant_amd", "csrc", "gemm_bf16.hip")
json.dump({"kernel": "gemm_bf16_nt256p_kernel (ping-pong streaming NT GEMM; nt_pp = 0 launches gemm_bf16_nt256s_kernel)", "kernel_source": "meant_amd/csrc/gemm_bf16.hip",
           "kernel_source_sha16": hashlib.sha256(open(_src, "rb").read()).hexdigest()[:16],   # bench.py flags the table as stale when the source moves on
            "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
           "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads)", "shapes": res},
          open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
