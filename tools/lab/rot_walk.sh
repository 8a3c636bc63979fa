# GPU box: the rotary projection with its panels walked one position range after the other (default) against memory order (lib_walk0)
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_gemm_stream.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
echo "== range walk (default)"; python tools/probe_epi_modes.py 10 2>&1 | grep -v amdgpu | grep rotary
echo "== memory order"; MEANT_LIB_PATH=$R/tools/lab/lib_walk0.so python tools/probe_epi_modes.py 10 2>&1 | grep -v amdgpu | grep rotary
done
bash tools/pmc_nt256.sh r04j > gpurun_out/r04j_pmc_nt.log 2>&1; tail -90 gpurun_out/r04j_pmc_nt.log | grep -E "ratio|\"[0-9]+,[0-9]+,[0-9]+" | paste - - | cut -c1-140 | head -3
