"""dev probe: time the NT bf16 GEMM entry point at a few shapes (MEANT_NT_W4 selects the kernel variant)"""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
shapes = [(786432, 768, 768), (786432, 2304, 768), (786432, 768, 3072), (8192, 8192, 8192)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st), "lin")
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    ref = (x[:4096].float() @ w.float().t())
    err = (y[:4096].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M={M} N={N} K={K}: {2.0*M*N*K/t/1e12:7.1f} TF ({t*1e3:.3f} ms)  relerr {err:.2e}", flush=True)
    del x, w, y
