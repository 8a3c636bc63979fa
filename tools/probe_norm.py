"""dev probe: RMSNorm forward / backward variants at the step's text shape (rows = 128*12*512, d = 768), bf16"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
rows, d = 786432, 768
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn(rows, d, device=dev).bfloat16()
pre = torch.randn(rows, d, device=dev).bfloat16(); dres = torch.randn(rows, d, device=dev).bfloat16()
g = torch.ones(d, device=dev); y = torch.empty_like(x); r = torch.empty(rows, device=dev)
dx = torch.empty_like(x); ds = torch.empty(d, device=dev)
wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
GB = rows * d * 2 / 1e9
for name, p in [("fwd", 0.0), ("fwd dropout .5", 0.5)]:
    t = timeit(lambda: check(lib.meant_rmsnorm_fwd(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows, d, 1e-8, p, 1234, 1, st)))
    print(f"{name:28s} {t:.3f} ms  {2*GB/t:6.2f} TB/s", flush=True)
for name, p, dr, gp, nb in [("bwd", 0.0, None, None, 3), ("bwd +dres", 0.0, dres, None, 4), ("bwd +gelu", 0.0, None, pre, 4), ("bwd +gelu +dropout", 0.5, None, pre, 4),
                            ("bwd dropout", 0.5, None, None, 3)]:
    t = timeit(lambda: check(lib.meant_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, 1e-8, p, 1234,
                                                   dr.data_ptr() if dr is not None else None, gp.data_ptr() if gp is not None else None, 1, ws.data_ptr(), wsb, st)))
    print(f"{name:28s} {t:.3f} ms  {nb*GB/t:6.2f} TB/s", flush=True)
# the MLM step's size (64 x 512 tokens): fixed costs of a launch show here
rows2 = 32768
for name, dr in [("bwd 32k rows", None), ("bwd +dres 32k rows", dres)]:
    t = timeit(lambda: check(lib.meant_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows2, d, 1e-8, 0.0, 1234,
                                                   dr.data_ptr() if dr is not None else None, None, 1, ws.data_ptr(), wsb, st)), n=50)
    print(f"{name:28s} {t*1e3:.1f} us", flush=True)
t = timeit(lambda: check(lib.meant_rmsnorm_fwd(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows2, d, 1e-8, 0.0, 1234, 1, st)), n=50)
print(f"{'fwd 32k rows':28s} {t*1e3:.1f} us", flush=True)
