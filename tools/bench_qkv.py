import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE
from tools.bench_kernels import timeit, st
dev = "cuda"
for (M, N, K) in [(786432, 2304, 768), (786432, 768, 2304), (301056, 2304, 768)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st()))
    t = timeit(f)
    print(f"M={M} N={N} K={K}: {t*1e3:.3f} ms {2*M*N*K/t/1e12:.1f} TFLOP/s")
    del x, w, y
