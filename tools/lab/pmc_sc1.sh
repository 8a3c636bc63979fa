cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MEANT_LIB_PATH=$R/tools/lab/lib_SC1.so
bash $R/tools/pmc_nt256.sh r04sc1 > $R/gpurun_out/r04sc1_pmc_nt.log 2>&1
tail -70 $R/gpurun_out/r04sc1_pmc_nt.log | grep -E "ratio|\"[0-9]+,[0-9]+,[0-9]+\"|dram_share" | paste - - - | cut -c1-200
