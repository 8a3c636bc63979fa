"""dev probe: cost of the NT GEMM epilogue variants at the step's shapes (plain / bias / residual / gelu+preact / rotary qkv)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
M = 786432
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
st = torch.cuda.current_stream().cuda_stream
for (N, K) in [(768, 768), (3072, 768), (768, 3072)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) / K ** 0.5
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    for name, args in [("plain", (None, None, None, 0)), ("bias", (b, None, None, 0)), ("bias+residual", (b, res, None, 0)),
                       ("bias+gelu", (b, None, None, 1)), ("bias+gelu+preact", (b, None, pre, 1))]:
        bb, rr, pp, epi = args
        def run():
            check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), bb.data_ptr() if bb is not None else None,
                                       rr.data_ptr() if rr is not None else None, N, y.data_ptr(), N,
                                       pp.data_ptr() if pp is not None else None, M, N, K, epi, 1, st), "lin")
        t = timeit(run)
        print(f"N={N} K={K} {name:18s} {t:.3f} ms  {fl/t/1e9:7.1f} TF", flush=True)
    del x, w, y, res, pre
# fused qkv projection with rotary (S=512, H=12, Dh=64, R=48)
K, H, Dh, R, S = 768, 12, 64, 48, 512
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(3 * H * Dh, K, device=dev, dtype=torch.bfloat16) / K ** 0.5
b = torch.randn(3 * H * Dh, device=dev)
qkv = torch.empty(M, 3 * H * Dh, device=dev, dtype=torch.bfloat16)
tabs = [torch.randn(S, R, device=dev) for _ in range(4)]
fl = 2.0 * M * 3 * H * Dh * K
for name, rot in [("qkv no rotary", 0), ("qkv rotary", 1)]:
    def run():
        check(lib.meant_qkv_proj_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), qkv.data_ptr(), M, K, S, H, Dh, R if rot else 0,
                                     *([t_.data_ptr() for t_ in tabs] if rot else [None] * 4), 1, st), "qkv")
    t = timeit(run)
    print(f"N=2304 K=768 {name:18s} {t:.3f} ms  {fl/t/1e9:7.1f} TF", flush=True)
