# lab (GPU box), third part: HIP API calls of the step (what autograd's stream-mismatch path inserts), with and without the one-rank RCCL group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stream_cliff; mkdir -p $O
for tag in plain reduce; do
  if [ $tag = reduce ]; then export MEANT_REDUCE_ALWAYS=1 GPU_MAX_HW_QUEUES=8 MEANT_LANG_PRIORITY=0; fi
  rocprofv3 --hip-trace -d $O/hip_$tag -o x -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/hip_$tag.log 2>&1
  echo "== $tag: $(grep '^{' $O/hip_$tag.log | cut -c1-160)"
  python3 $R/tools/hip_api_counts.py $O/hip_$tag/x_results.db 16
  rm -f $O/hip_$tag/x_results.db
done
