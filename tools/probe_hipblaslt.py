"""reference point only (not used by the product): torch.matmul (hipBLASLt) vs the shipped NT kernel at the step's shapes"""
import torch, time, sys
sys.path.insert(0, ".")
from meant_amd import ops
from meant_amd._lib import lib, check
dev = torch.device("cuda")
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, N, K) in [(786432, 768, 768), (786432, 2304, 768), (786432, 3072, 768), (786432, 768, 3072), (301056, 2304, 768), (301056, 3072, 768), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_blas = bench(lambda: torch.matmul(x, w.t(), out=y))
    st = torch.cuda.current_stream().cuda_stream
    def mine():
        check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st), "lin")
    t_mine = bench(mine)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: hipBLASLt {fl/t_blas/1e12:7.1f} TF ({t_blas*1e3:.3f} ms)   nt256 {fl/t_mine/1e12:7.1f} TF ({t_mine*1e3:.3f} ms)", flush=True)
    del x, w, y
