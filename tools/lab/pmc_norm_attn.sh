# GPU box: the two PMC passes over tools/pmc_step_kernels.py (norm / attention HBM traffic) + reduction: bash tools/lab/pmc_norm_attn.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z); rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o x -- python3 $R/tools/pmc_step_kernels.py > $d.log 2>&1
done
python3 $R/tools/parse_pmc_kernels.py $R/gpurun_out/${TAG}_pmc_fetch_size $R/gpurun_out/${TAG}_pmc_write_size $R/gpurun_out/${TAG}_hbm_traffic.json > /dev/null 2> $R/gpurun_out/${TAG}_parse.err
grep -E "in_units|\"[a-z_0-9 ]+ \[" $R/gpurun_out/${TAG}_hbm_traffic.json | paste - - | cut -c1-150 | head -40
