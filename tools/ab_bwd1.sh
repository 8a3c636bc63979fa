cd $GRAFT_REPO_ROOT
for v in 1 0; do
  export MEANT_ATTN_BWD1=$v
  echo "== attn_bwd1=$v"
  python tools/bench_mlm.py 2>/dev/null | tail -1 | cut -c1-200
  python bench.py --no-cpu-baseline --model meant_vqa 2>/dev/null | tail -1 | cut -c1-140
  python bench.py --no-cpu-baseline --model meant_vision 2>/dev/null | tail -1 | cut -c1-140
  python tools/bench_variants.py 2>/dev/null | tail -1 | cut -c1-160
  python bench.py --no-cpu-baseline --encoders 12 --batch-per-gpu 32 2>/dev/null | tail -1 | cut -c1-140
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-140
done
