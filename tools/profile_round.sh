# one call that collects what profiles/ holds for a build (GPU box): bash tools/profile_round.sh TAG
#   gpurun_out/TAG_pmc_fetch, TAG_pmc_write (+ TAG_hbm_traffic.json), gpurun_out/sq_TAG.txt, gpurun_out/prof_TAG/, gpurun_out/TAG_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/${TAG}_pmc_$(echo $c | tr A-Z a-z); rm -rf $d
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o x -- python3 $R/tools/pmc_step_kernels.py > $d.log 2>&1
done
python3 $R/tools/parse_pmc_kernels.py $R/gpurun_out/${TAG}_pmc_fetch_size $R/gpurun_out/${TAG}_pmc_write_size $R/gpurun_out/${TAG}_hbm_traffic.json > /dev/null 2> $R/gpurun_out/${TAG}_parse.err
echo "== traffic"; grep -A8 "attn_bwd1" $R/gpurun_out/${TAG}_hbm_traffic.json | grep -E "attn_bwd1|in_units" 
bash $R/tools/pmc_sq.sh $TAG tools/probe_attn2.py 2 > /dev/null 2>&1
echo "== sq"; grep -A1 "attn_" $R/gpurun_out/sq_$TAG.txt | cut -c1-400
bash $R/tools/profile_step.sh $TAG > $R/gpurun_out/${TAG}_profile_step.log 2>&1
echo "== step"; head -14 $R/gpurun_out/${TAG}_profile_step.log | cut -c1-160
cd $R && python3 bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
echo "== bench"; tail -1 $R/gpurun_out/${TAG}_bench.json | cut -c1-400
