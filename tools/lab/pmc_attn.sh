cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_$n
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -o x -- python3 $R/tools/probe_attn2.py 2 > $R/gpurun_out/pmc_$n.log 2>&1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_SQ_WAVE_CYCLES $R/gpurun_out/pmc_GRBM_GUI_ACTIVE --filter attn_ > $R/gpurun_out/pmc_summary.txt
