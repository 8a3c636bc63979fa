#!/usr/bin/env python3
"""dev probe: ablations of the pooled RMSNorm forward at the text shape: gelu on load on / off, dropout on / off, pooled output vs
pooled input (+ y written), and the plain packed forward for reference"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from meant_amd._lib import lib, check
dev = torch.device("cuda")
rows, d, S = 786432, 768, 512
G = rows // S
st = torch.cuda.current_stream().cuda_stream
pre = torch.randn(rows, d, device=dev).bfloat16(); g = torch.ones(d, device=dev); r = torch.empty(rows, device=dev)
y = torch.empty_like(pre); pooled = torch.empty(G, d, device=dev)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for gelu, p in ((1, 0.5), (1, 0.0), (0, 0.5), (0, 0.0)):
    t = timeit(lambda: check(lib.meant_rmsnorm_fwd_pooled(pre.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 0, gelu, 1e-8, p, 77, 1, st)))
    print(f"pooled output, gelu_on_load={gelu} dropout={p}: {t:.3f} ms  ({rows*d*2/t/1e9:.2f} TB/s of the one tensor read)", flush=True)
t = timeit(lambda: check(lib.meant_rmsnorm_fwd_pooled(pre.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), pooled.data_ptr(), rows, d, S, 1, 0, 1e-8, 0.0, 0, 1, st)))
print(f"pooled input, y written: {t:.3f} ms", flush=True)
t = timeit(lambda: check(lib.meant_rmsnorm_fwd_pooled(pre.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 1, 0, 1e-8, 0.0, 0, 1, st)))
print(f"pooled input, statistics + means only: {t:.3f} ms", flush=True)
t = timeit(lambda: check(lib.meant_rmsnorm_stats(pre.data_ptr(), r.data_ptr(), rows, d, 1e-8, 1, st)))
print(f"statistics only (packed, 4 waves / SIMD): {t:.3f} ms", flush=True)
for p in (0.0, 0.5):
    t = timeit(lambda: check(lib.meant_rmsnorm_fwd(pre.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows, d, 1e-8, p, 77, 1, st)))
    print(f"plain packed forward, dropout={p}: {t:.3f} ms", flush=True)
