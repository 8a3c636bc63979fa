"""lab: the streaming NT GEMM at shrinking M (A and C footprints from 1.2 GB down to 50 MB, i.e. inside the 256 MB Infinity Cache across\nrepeated launches): does the rate depend on where the A tiles come from?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(786432, 768, 768), (131072, 768, 768), (65536, 768, 768), (32768, 768, 768), (786432, 2304, 768), (65536, 2304, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st))
    for _ in range(5): f()
    torch.cuda.synchronize()
    n = max(10, int(786432 / M) * 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"M={M} N={N} K={K}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s  (A {M*K*2/1e6:.0f} MB, C {M*N*2/1e6:.0f} MB)", flush=True)
