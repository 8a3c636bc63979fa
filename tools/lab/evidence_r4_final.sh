# GPU box: everything profiles/r04_* holds for the final build, in one call: bash tools/lab/evidence_r4_final.sh
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1; tail -30 gpurun_out/r04_profile_round.log | cut -c1-300
bash tools/pmc_nt256.sh r04f > gpurun_out/r04f_pmc_nt.log 2>&1; tail -70 gpurun_out/r04f_pmc_nt.log | grep -E "ratio|\"[0-9]+,[0-9]+,[0-9]+\"" | paste - - | cut -c1-120
cd $R
python tools/probe_pp.py 10 > gpurun_out/r04_probe_pp.txt 2>&1; grep -v amdgpu gpurun_out/r04_probe_pp.txt
python tools/probe_tn_pp.py > gpurun_out/r04_probe_tn.txt 2>&1; grep -v amdgpu gpurun_out/r04_probe_tn.txt
python tools/probe_epi_modes.py 10 > gpurun_out/r04_probe_epi.txt 2>&1; grep -v amdgpu gpurun_out/r04_probe_epi.txt
