# secondary figures quoted in DESIGN.md section 6 (GPU box): bash tools/run_secondary.sh TAG -> gpurun_out/sec_TAG.log
R=$GRAFT_REPO_ROOT; TAG=$1; L=$R/gpurun_out/sec_$TAG.log; : > $L
run() { echo "== $*" >> $L; "$@" 2>/dev/null | tail -1 >> $L; }
cd $R
run python bench.py --no-cpu-baseline --with-optimizer --forward-only
run python bench.py --no-cpu-baseline --heads 8
run python bench.py --no-cpu-baseline --encoders 12 --batch-per-gpu 32
run python bench.py --no-cpu-baseline --encoders 12 --steps 4 --warmup 2
run python bench.py --no-cpu-baseline --encoders 12 --micro-batches 2 --steps 4 --warmup 2
run python bench.py --no-cpu-baseline --model meant_vqa
run python bench.py --no-cpu-baseline --model meant_vqa --batch-per-gpu 1024
run python bench.py --no-cpu-baseline --model meant_vision
run python bench.py --no-cpu-baseline --from-host u8
run python tools/bench_variants.py
run python tools/bench_mlm.py
run python tools/bench_timesformer.py --batch 32
cat $L | cut -c1-700
