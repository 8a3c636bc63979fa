#!/usr/bin/env python3
"""Launch the streaming NT GEMM (gemm_bf16_nt256p_kernel) once per shape the bench step uses (plain epilogue), 3 times each, so that
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) can attribute HBM traffic per shape.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python tools/pmc_nt256.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE

SHAPES = [(786432, 2304, 768), (786432, 768, 768), (786432, 768, 2304),
          (301056, 768, 1024), (301056, 2304, 768), (301056, 768, 768), (301056, 768, 2304)]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st))
    torch.cuda.synchronize()
    del x, w, y
print("done")
