#!/usr/bin/env python3
"""fixed (prologue+epilogue) vs per-K cost of the NT GEMM: time at several K for the same M, N"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE
from tools.bench_kernels import timeit, st
dev = "cuda"
M, N = 393216, 768
for K in (64, 384, 768, 1536, 3072):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st()))
    t = timeit(f)
    print(f"K={K:5d}: {t*1e3:7.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s   per-K64-step {t*1e6/(K/64):7.2f} us")
