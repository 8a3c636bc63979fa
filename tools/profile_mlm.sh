# per-kernel profile of the MLM pretrainer step: rocprofv3 --kernel-trace --stats
# usage (GPU box): bash tools/profile_mlm.sh TAG [bench_mlm args]   -> gpurun_out/prof_TAG/{stats.csv,bench.log}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
mkdir -p $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o x -- python3 $R/tools/bench_mlm.py "$@" > $R/gpurun_out/prof_$TAG/bench.log 2>&1
python3 $R/tools/rocprof_stats.py $R/gpurun_out/prof_$TAG/x_results.db $R/gpurun_out/prof_$TAG/stats.csv | head -45
tail -1 $R/gpurun_out/prof_$TAG/bench.log | cut -c1-300
