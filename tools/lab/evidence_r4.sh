# GPU box: the counter evidence of round 4 in one call: bash tools/lab/evidence_r4.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_nt256.sh r04 > $R/gpurun_out/r04_pmc_nt.log 2>&1
tail -70 $R/gpurun_out/r04_pmc_nt.log | grep -E "ratio|\"[0-9]+,[0-9]+,[0-9]+\"|dram_share" | paste - - - | cut -c1-200
bash $R/tools/pmc_sq.sh r04gemm tools/probe_pp.py 3 786432x768x768 786432x2304x768 786432x768x2304 > /dev/null 2>&1
echo "== sq gemm"; grep -B1 -A3 "nt256p\|tn256p" $R/gpurun_out/sq_r04gemm.txt | cut -c1-600 | head -40
