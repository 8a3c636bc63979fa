#!/usr/bin/env python3
"""dev probe: the plain packed RMSNorm forward / backward (with and without dropout, with the GELU derivative) and the statistics
pass at the step's text shape, one line per build (MEANT_LIB_PATH)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from meant_amd._lib import lib, check
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rows, d = 786432, 768
x = torch.randn(rows, d, device=dev).bfloat16(); g = torch.ones(d, device=dev); r = torch.empty(rows, device=dev)
y = torch.empty_like(x); dy = torch.randn(rows, d, device=dev).bfloat16(); dx = torch.empty_like(x); ds = torch.empty(d, device=dev)
wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
out = [os.path.basename(os.environ.get("MEANT_LIB_PATH", "tree"))]
for p in (0.0, 0.5):
    t = timeit(lambda: check(lib.meant_rmsnorm_fwd(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows, d, 1e-8, p, 77, 1, st)))
    out.append(f"fwd p={p}: {t:.3f}")
    t = timeit(lambda: check(lib.meant_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, 1e-8, p, 77,
                                                   None, None, 1, ws.data_ptr(), wsb, st)))
    out.append(f"bwd p={p}: {t:.3f}")
t = timeit(lambda: check(lib.meant_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, 1e-8, 0.5, 77,
                                               dy.data_ptr(), x.data_ptr(), 1, ws.data_ptr(), wsb, st)))
out.append(f"bwd + dres + gelu' p=0.5: {t:.3f}")
t = timeit(lambda: check(lib.meant_rmsnorm_stats(x.data_ptr(), r.data_ptr(), rows, d, 1e-8, 1, st)))
out.append(f"stats: {t:.3f}")
print("  ".join(out), flush=True)
