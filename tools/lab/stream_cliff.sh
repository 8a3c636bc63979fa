# lab (GPU box): the stream-count cliff of DESIGN.md section 7.  bench.py with 0 / 1 / 2 / 4 extra HIP streams created before the model runs:
# (a) plain runs, alternating, for the step time; (b) one rocprofv3 --kernel-trace run each for the stream -> hardware-queue map.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stream_cliff; mkdir -p $O
line() { grep '^{' $1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], 'samples/s', d['ms_per_step'], 'ms', 'reduce' if d['config']['grad_allreduce'] else '')"; }
for rep in 1 2; do
  for n in 0 1 2 4; do
    python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --extra-streams $n > $O/plain_${n}_$rep.log 2>/dev/null
    echo "extra streams $n (run $rep): $(line $O/plain_${n}_$rep.log)"
  done
done
for n in 0 1 2; do
  GPU_MAX_HW_QUEUES=8 python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --extra-streams $n > $O/q8_$n.log 2>/dev/null
  echo "GPU_MAX_HW_QUEUES=8, extra streams $n: $(line $O/q8_$n.log)"
done
for n in 0 1 2; do
  MEANT_LANG_PRIORITY=0 python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --extra-streams $n > $O/noprio_$n.log 2>/dev/null
  echo "language stack without stream priority, extra streams $n: $(line $O/noprio_$n.log)"
done
for rep in 1 2 3; do
  python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/pair_plain_$rep.log 2>/dev/null
  echo "plain (pair $rep): $(line $O/pair_plain_$rep.log)"
  MEANT_REDUCE_ALWAYS=1 python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/pair_reduce_$rep.log 2>$O/pair_reduce_$rep.err
  echo "MEANT_REDUCE_ALWAYS=1, one-rank RCCL group (pair $rep): $(line $O/pair_reduce_$rep.log)"
done
for n in 0 1 2; do
  rocprofv3 --kernel-trace -d $O/trace_$n -o x -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --extra-streams $n > $O/trace_$n.log 2>&1
  python3 $R/tools/stream_queues.py $O/trace_$n/x_results.db "extra streams $n ($(line $O/trace_$n.log))"
  rm -f $O/trace_$n/x_results.db
done
MEANT_REDUCE_ALWAYS=1 rocprofv3 --kernel-trace -d $O/trace_reduce -o x -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/trace_reduce.log 2>&1
python3 $R/tools/stream_queues.py $O/trace_reduce/x_results.db "one-rank RCCL ($(line $O/trace_reduce.log))"
rm -f $O/trace_reduce/x_results.db
