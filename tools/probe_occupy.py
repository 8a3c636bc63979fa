"""dev probe: NT GEMM time with N CUs pinned by another stream's long-running kernel (static vs dynamic tile walk)"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "occupy.so"))
occ.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device("cuda")
M, N, K = 786432, 768, 768
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm():
    check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, s1.cuda_stream), "lin")
for nb in (0, 8, 16, 32, 64):
    torch.cuda.synchronize()
    for _ in range(2): gemm()
    torch.cuda.synchronize()
    if nb: occ.occupy_launch(nb, 20000.0, s2.cuda_stream)          # 20 ms of pinned CUs
    torch.cuda.synchronize() if nb == 0 else None
    import time; time.sleep(0.002)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        e0.record()
        for _ in range(5): gemm()
        e1.record()
    torch.cuda.synchronize()
    print(f"{nb:3d} CUs pinned: {e0.elapsed_time(e1)/5:.3f} ms per GEMM (ideal {1.03*256/(256-nb):.3f})", flush=True)
