"""dev probe: the fp32 (f32-MFMA) GEMM at the shapes of the bf16 tier's fp32 tail: temporal-encoder Linears and weight compositions"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd import ops
dev = torch.device("cuda")
for (M, N, K) in [(1536, 1536, 1536), (2304, 768, 768), (768, 768, 2304), (128, 1536, 1536), (1536, 4608, 1536)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    run = lambda: ops._gemm_f32(A, B, C, M, N, K, (K, 1), (1, K), (N, 1))
    run(); torch.cuda.synchronize()
    err = (C - A @ B.t()).abs().max().item() / (A @ B.t()).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"f32 GEMM M={M} N={N} K={K}: {2.0*M*N*K/t/1e12:6.1f} TF ({t*1e6:.0f} us) relerr {err:.1e}", flush=True)
