#!/usr/bin/env python3
"""dev probe: the gelu-on-load pooled norm pair at the step's text and vision shapes, one line per build
(MEANT_LIB_PATH=tools/lab/lib_<tag>.so): forward with gelu + dropout / gelu only / neither, backward with gelu + dropout"""
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
out = [os.path.basename(os.environ.get("MEANT_LIB_PATH", "tree"))]
for rows, S in ((786432, 512), (301056, 196)):
    d = 768; G = rows // S
    pre = torch.randn(rows, d, device=dev).bfloat16(); g = torch.ones(d, device=dev); r = torch.empty(rows, device=dev)
    pooled = torch.empty(G, d, device=dev); dyp = torch.randn(G, d, device=dev); dx = torch.empty_like(pre); ds = torch.empty(d, device=dev)
    wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    for gelu, p in ((1, 0.5), (1, 0.0), (0, 0.0)):
        t = timeit(lambda: check(lib.meant_rmsnorm_fwd_pooled(pre.data_ptr(), g.data_ptr(), None, r.data_ptr(), pooled.data_ptr(), rows, d, S, 0, gelu, 1e-8, p, 77, 1, st)))
        out.append(f"f{gelu}{int(p*10)}={t:.3f}")
    for p in (0.5, 0.0):
        t = timeit(lambda: check(lib.meant_rmsnorm_bwd_pooled(dyp.data_ptr(), 1, None, g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, S, 1e-8, p, 77,
                                                              None, 0, pre.data_ptr(), 1, ws.data_ptr(), wsb, st)))
        out.append(f"b{int(p*10)}={t:.3f}")
    out.append("|")
print(" ".join(out), flush=True)
