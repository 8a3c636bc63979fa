cd $GRAFT_REPO_ROOT
export MEANT_FUSE_NORM_LINEAR=1
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --encoders 12 --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c56-130; }
run A=1
run MEANT_LANG_PRIORITY=0
run MEANT_TWO_STREAMS=0
run PYTORCH_HIP_ALLOC_CONF=expandable_segments:True
run MEANT_ATTN_BWD1=0
run A=1
