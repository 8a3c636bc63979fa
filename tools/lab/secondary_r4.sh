# GPU box: secondary figures + the fp32-tier line + the one-rank RCCL pair, one call: bash tools/lab/secondary_r4.sh
R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_secondary.sh r04 > /dev/null 2>&1
L=$R/gpurun_out/sec_r04.log
run() { echo "== $*" >> $L; "$@" 2>/dev/null | tail -1 >> $L; }
run python bench.py --no-cpu-baseline --fp32-batch 8
run python bench.py --no-cpu-baseline
MEANT_REDUCE_ALWAYS=1 run python bench.py --no-cpu-baseline
run python bench.py --no-cpu-baseline
MEANT_REDUCE_ALWAYS=1 run python bench.py --no-cpu-baseline
python - <<'PY'
import json, os
for ln in open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "sec_r04.log")):
    if ln.startswith("=="): print(ln.strip()); continue
    try:
        d = json.loads(ln); print("   ", d.get("value"), d.get("unit"), d.get("ms_per_step"), "ms", "frac", (d.get("roofline") or {}).get("frac"))
    except Exception: print("   ", ln.strip()[:300])
PY
