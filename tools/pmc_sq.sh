# SQ counters (occupancy, wait / stall / active split, MFMA pipe utilisation, LDS bank conflicts) of every kernel a program launches:
# two rocprofv3 --pmc passes (8 SQ slots per pass on gfx950) + GRBM_GUI_ACTIVE, summarised per kernel by tools/pmc_sq_summary.py.
# usage (GPU box): bash tools/pmc_sq.sh TAG tools/probe_attn2.py 2      -> gpurun_out/sq_TAG.txt
# (the program goes directly after `--`: no env / bash -c hop under the profiler)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
PROG=$R/$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sq_${TAG}_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sq_${TAG}_$i -o x -- python3 $PROG "$@" > $R/gpurun_out/sq_${TAG}_$i.log 2>&1
done
python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sq_${TAG}_1 $R/gpurun_out/sq_${TAG}_2 > $R/gpurun_out/sq_${TAG}.txt
cat $R/gpurun_out/sq_${TAG}.txt
