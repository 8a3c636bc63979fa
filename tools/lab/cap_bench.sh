# GPU box: the headline step with the NT GEMM's grid capped (MEANT_NT_GRID_CAP): does leaving CUs to the other stream pay?
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
for cap in 0 240 224 208 192; do
  MEANT_NT_GRID_CAP=$cap python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('cap $cap:', d['value'], 'samples/s', d['ms_per_step'], 'ms  frac', d['roofline']['frac'], ' sharing', d['roofline'].get('achieved_while_sharing_cus_with_second_stream'))"
done; done
