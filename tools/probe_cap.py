"""NT GEMM throughput against the number of CUs it may take (option nt_grid_cap): the chip is power-limited under MFMA load, so
fewer CUs run at a higher clock -- how much throughput do the last 32 / 64 / 96 CUs really add?  Optionally with an HBM-bound kernel
(a bf16 copy) on a second stream, to see what the two together deliver."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
M, N, K = 786432, 768, 768
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
b = torch.randn(N, device=dev, dtype=torch.float32)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
src = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
dst = torch.empty_like(src)
side = torch.cuda.Stream()
def gemm():
    check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st), "lin")
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
copy_ms = timed(lambda: dst.copy_(src))
print(f"copy alone: {copy_ms:.3f} ms = {2 * src.numel() * 2 / copy_ms / 1e9:.2f} TB/s")
for cap in (0, 224, 192, 160, 128, 96, 64):
    check(lib.meant_set_option(b"nt_grid_cap", cap), "opt")
    g = timed(gemm)
    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dst.copy_(src)
        gemm()
        torch.cuda.current_stream().wait_stream(side)
    t = timed(both)
    print(f"cap {cap or 256:3d} CUs: GEMM alone {g:.3f} ms = {2.0 * M * N * K / g / 1e9:7.1f} TF   GEMM + copy on a second stream {t:.3f} ms (in sequence {g + copy_ms:.3f})")
check(lib.meant_set_option(b"nt_grid_cap", 0), "opt")
