# GPU box: the LDS-free epilogue (default: full lines, no barrier behind it; lib_bar: with the barrier; lib_half: 64-byte segments, with
# the barrier) against the patch epilogue (lib_patch), with stamps
R=$GRAFT_REPO_ROOT; cd $R
P="python tools/probe_pp.py 10"; E="python tools/probe_epi_modes.py 10"
echo "== swap epilogue, full lines, no barrier (default)"; $P 2>&1 | grep -v amdgpu; $E 2>&1 | grep -v amdgpu
echo "== swap epilogue, full lines, barrier"; MEANT_LIB_PATH=tools/lab/lib_bar.so $E 2>&1 | grep -v amdgpu
echo "== swap epilogue, 64-byte segments, barrier"; MEANT_LIB_PATH=tools/lab/lib_half.so $E 2>&1 | grep -v amdgpu
echo "== patch epilogue"; MEANT_LIB_PATH=tools/lab/lib_patch.so $E 2>&1 | grep -v amdgpu
echo "== stamps, swap epilogue"; python tools/lab/stamp_pp.py 2>&1 | grep -v amdgpu
echo "== stamps, patch epilogue"; PROBE_LIB=lib_PPSTAMPP.so python tools/lab/stamp_pp.py 2>&1 | grep -v amdgpu
timeout -k 10 900 python -m pytest tests/test_gpu_gemm_stream.py tests/test_gpu_ops.py tests/test_gpu_norm_linear.py -x -q -m gpu 2>&1 | tail -5
