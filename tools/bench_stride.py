#!/usr/bin/env python3
"""Does the row stride of the activation operand matter (L2 channel hot-spotting at 1536-byte rows)?"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE
from tools.bench_kernels import timeit, st

dev = "cuda"
M, N, K = 786432, 768, 768
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
b = torch.randn(N, device=dev)
for ldx, ldy in [(768, 768), (832, 768), (768, 832), (832, 832), (800, 800), (776, 776), (896, 896), (1024, 1024)]:
    xb = torch.randn(M, ldx, device=dev).bfloat16()
    yb = torch.empty(M, ldy, device=dev, dtype=torch.bfloat16)
    f = lambda: check(lib.meant_linear_fwd(xb.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), None, 0, yb.data_ptr(), ldy, None, M, N, K, EPI_NONE, BF16, st()))
    t = timeit(f)
    print(f"ldx={ldx:5d} ldy={ldy:5d}: {t*1e3:7.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s")
    del xb, yb
