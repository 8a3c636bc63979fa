"""per-kernel summary of the two SQ passes of tools/pmc_sq.sh: mean over dispatches (grouped by kernel name and grid size) of
   waves resident per SIMD = 4 * SQ_WAVE_CYCLES / (kernel cycles * 1024 SIMDs)      [SQ_WAVE_CYCLES counts quad-cycles]
   wait / stall / active   = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint buckets of a wave's life)
   MFMA pipe utilisation   = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024)
   LDS                     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
   kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.  (MI355X_MICROARCH.md, rocprofv3 PMC slots.)"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, g), cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if "GRBM_GUI_ACTIVE" not in d or "SQ_WAVE_CYCLES" not in d or d["SQ_WAVE_CYCLES"] <= 0:
        continue
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    rows.append((cyc * len(cs["GRBM_GUI_ACTIVE"]), k, g, d, cyc, len(cs["GRBM_GUI_ACTIVE"])))
for _, k, g, d, cyc, n in sorted(rows, reverse=True)[:24]:
    wc = d["SQ_WAVE_CYCLES"]
    lds = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"] if d.get("SQ_LDS_IDX_ACTIVE") else 0.0
    print(f"{k[:110]}  grid={g} x{n}")
    print(f"    kernel cycles {cyc:.4g} | waves/SIMD resident {4 * wc / cyc / 1024:.2f} | wait {d['SQ_WAIT_ANY'] / wc:.2f} stall {d['SQ_WAIT_INST_ANY'] / wc:.2f} "
          f"active {d['SQ_ACTIVE_INST_ANY'] / wc:.2f} (VALU {d['SQ_ACTIVE_INST_VALU'] / wc:.2f}, LDS {d['SQ_ACTIVE_INST_LDS'] / wc:.2f}) | MFMA pipe {d['SQ_VALU_MFMA_BUSY_CYCLES'] / cyc / 1024:.2f} | "
          f"LDS bank conflicts {lds:.3f} | per wave: {d['SQ_INSTS_VALU'] / d['SQ_WAVES']:.0f} VALU {d['SQ_INSTS_MFMA'] / d['SQ_WAVES']:.0f} MFMA {d['SQ_INSTS_SALU'] / d['SQ_WAVES']:.0f} SALU {d['SQ_INSTS_LDS'] / d['SQ_WAVES']:.0f} LDS")
