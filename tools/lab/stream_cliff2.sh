# lab (GPU box), second part: the one-rank RCCL rehearsal (MEANT_REDUCE_ALWAYS=1) under the knobs that change the stream -> hardware-queue map
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stream_cliff; mkdir -p $O
line() { grep '^{' $1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], 'samples/s', d['ms_per_step'], 'ms', 'reduce' if d['config']['grad_allreduce'] else '')"; }
run() { tag=$1; shift; env "$@" python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/$tag.log 2>/dev/null; echo "$tag [$*]: $(line $O/$tag.log)"; }
for rep in 1 2 3; do
  run plain_$rep X=1
  run reduce_$rep MEANT_REDUCE_ALWAYS=1
  run reduce_q8_$rep MEANT_REDUCE_ALWAYS=1 GPU_MAX_HW_QUEUES=8
  run reduce_noprio_$rep MEANT_REDUCE_ALWAYS=1 MEANT_LANG_PRIORITY=0
  run reduce_q8_noprio_$rep MEANT_REDUCE_ALWAYS=1 GPU_MAX_HW_QUEUES=8 MEANT_LANG_PRIORITY=0
  run reduce_onestream_$rep MEANT_REDUCE_ALWAYS=1 MEANT_TWO_STREAMS=0
done
for cfg in "reduce_q8 GPU_MAX_HW_QUEUES=8" "reduce_noprio MEANT_LANG_PRIORITY=0" "reduce_q8_noprio GPU_MAX_HW_QUEUES=8 MEANT_LANG_PRIORITY=0"; do
  set -- $cfg; tag=$1; shift
  for kv in "$@"; do export $kv; done
  MEANT_REDUCE_ALWAYS=1 rocprofv3 --kernel-trace -d $O/trace_$tag -o x -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/trace_$tag.log 2>&1
  for kv in "$@"; do unset ${kv%%=*}; done
  python3 $R/tools/stream_queues.py $O/trace_$tag/x_results.db "one-rank RCCL, $* ($(line $O/trace_$tag.log))"
  rm -f $O/trace_$tag/x_results.db
done
