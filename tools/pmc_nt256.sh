# HBM-side traffic of the streaming NT GEMM per shape: rocprofv3 PMC passes (separate, as MI355X_MICROARCH.md prescribes) over
# tools/pmc_nt256.py, reduced by tools/parse_pmc_traffic.py.  usage (GPU box): bash tools/pmc_nt256.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmcnt_${TAG}_$n
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcnt_${TAG}_$n -- python3 $R/tools/pmc_nt256.py > $R/gpurun_out/pmcnt_${TAG}_$n.log 2>&1
done
cd $R && python3 tools/parse_pmc_traffic.py gpurun_out/pmcnt_${TAG}_FETCH_SIZE gpurun_out/pmcnt_${TAG}_WRITE_SIZE gpurun_out/pmcnt_${TAG}.json gpurun_out/pmcnt_${TAG}_TCC_EA0_RDREQ_sum | tail -60
