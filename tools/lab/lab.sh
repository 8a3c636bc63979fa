cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset MEANT_LIB_PATH; else export MEANT_LIB_PATH=$R/tools/lab/lib_$v.so; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/lab_$v -o x -- python3 $R/tools/probe_attn2.py 6 > $R/gpurun_out/lab_$v.log 2>&1
  echo "== $v"; grep "G=" $R/gpurun_out/lab_$v.log
  python3 $R/tools/rocprof_stats.py $R/gpurun_out/lab_$v/x_results.db | head -4
done
