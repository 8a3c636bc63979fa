#!/usr/bin/env python3
"""dev probe: the pooled RMSNorm pair of the encoder tail at the step's text shape (gelu(pre) formed on load, dropout 0.5, pooled
gradient): meant_rmsnorm_fwd_pooled / meant_rmsnorm_bwd_pooled"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from meant_amd._lib import lib, check
dev = torch.device("cuda")
rows, d, S = 786432, 768, 512
G = rows // S
st = torch.cuda.current_stream().cuda_stream
pre = torch.randn(rows, d, device=dev).bfloat16(); g = torch.ones(d, device=dev); r = torch.empty(rows, device=dev)
pooled = torch.empty(G, d, device=dev); dyp = torch.randn(G, d, device=dev); dx = torch.empty_like(pre); ds = torch.empty(d, device=dev)
wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timeit(lambda: check(lib.meant_rmsnorm_fwd_pooled(pre.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 0, 1, 1e-8, 0.5, 77, 1, st)))
print(f"fwd pooled, gelu on load      {t:.3f} ms", flush=True)
t = timeit(lambda: check(lib.meant_rmsnorm_bwd_pooled(dyp.data_ptr(), 1, None, g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, S, 1e-8, 0.5, 77,
                                                      None, 0, pre.data_ptr(), 1, ws.data_ptr(), wsb, st)))
print(f"bwd pooled dy, gelu on load   {t:.3f} ms", flush=True)
