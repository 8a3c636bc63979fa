#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels through the C ABI (GPU box only).
usage: python tools/bench_kernels.py [gemm|attn|norm|all]"""
import math
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, F32, EPI_NONE, EPI_GELU, EPI_RESIDUAL


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def st():
    return torch.cuda.current_stream().cuda_stream


def bench_gemm():
    dev = "cuda"
    print("== bf16 NT GEMM  y[M,N] = x[M,K] w[N,K]^T + b")
    for (M, N, K) in [(6144 * 16, 768, 768), (6144 * 16, 2304, 768), (2352 * 16, 768, 1024), (6144 * 128, 768, 768), (8192, 8192, 8192), (4096, 4096, 4096)]:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st()))
        t = timeit(f)
        print(f"  M={M:7d} N={N:5d} K={K:5d}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s   ({(M*K+M*N)*2/t/1e9:6.0f} GB/s act)")
        ref = (x[:256].float() @ w.float().t() + b)
        err = (y[:256].float() - ref).abs().max().item()
        assert err < 0.1, err
    print("== bf16 TN GEMM  dW[N,K] += dY[M,N]^T X[M,K], db += colsum(dY)")
    for (M, N, K) in [(6144 * 16, 768, 768), (6144 * 16, 2304, 768), (6144 * 128, 768, 768)]:
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        f = lambda: check(lib.meant_linear_bwd_dw(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, BF16, None, 0, st()))
        t = timeit(f)
        print(f"  M={M:7d} N={N:5d} K={K:5d}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TFLOP/s")
        dw.zero_(); db.zero_(); f(); torch.cuda.synchronize()
        refb = dy.float().sum(0)
        assert ((db - refb).abs().max() / refb.abs().max()).item() < 1e-3
        ref = dy[:, :64].float().t() @ x.float()
        rel = ((dw[:64] - ref).abs().max() / ref.abs().max()).item()
        assert rel < 1e-2, rel


def bench_norm():
    dev = "cuda"
    print("== RMSNorm fwd / bwd (bf16)")
    for rows, d in [(6144 * 128, 768), (2352 * 128, 768)]:
        x = torch.randn(rows, d, device=dev).bfloat16()
        g = torch.ones(d, device=dev)
        y = torch.empty_like(x); r = torch.empty(rows, device=dev)
        f = lambda: check(lib.meant_rmsnorm_fwd(x.data_ptr(), g.data_ptr(), y.data_ptr(), r.data_ptr(), rows, d, 1e-8, 0.0, 0, BF16, st()))
        t = timeit(f)
        print(f"  fwd rows={rows} d={d}: {t*1e3:7.3f} ms  {2*rows*d*2/t/1e9:7.0f} GB/s")
        dx = torch.empty_like(x); ds = torch.empty(d, device=dev)
        wsb = lib.meant_rmsnorm_bwd_ws(rows, d); ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        f = lambda: check(lib.meant_rmsnorm_bwd(y.data_ptr(), x.data_ptr(), g.data_ptr(), r.data_ptr(), dx.data_ptr(), ds.data_ptr(), rows, d, 1e-8, 0.0, 0, BF16, ws.data_ptr(), wsb, st()))
        t = timeit(f)
        print(f"  bwd rows={rows} d={d}: {t*1e3:7.3f} ms  {3*rows*d*2/t/1e9:7.0f} GB/s")


def bench_attn():
    dev = "cuda"
    print("== flash attention (bf16) fwd / bwd, Dh=64 H=12")
    for (G, S, causal) in [(12 * 32, 512, 1), (12 * 32, 196, 0)]:
        H, Dh = 12, 64
        D = H * Dh
        qkv = torch.randn(G * S, 3 * D, device=dev).bfloat16()
        o = torch.empty(G * S, D, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(G, H, S, 2, device=dev)
        mask = torch.ones(G, S, device=dev) if causal else None
        scale = 1 / math.sqrt(D)
        wsb = lib.meant_attn_ws(G, S, H, Dh, BF16); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
        f = lambda: check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mask.data_ptr() if mask is not None else None, G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st()))
        t = timeit(f)
        fl = 4 * G * H * S * S * Dh
        print(f"  fwd G={G} S={S} causal={causal}: {t*1e3:7.3f} ms  {fl/t/1e12:6.1f} TFLOP/s (full-square count)  {4*G*S*D*2/t/1e9:6.0f} GB/s")
        do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
        f = lambda: check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mask.data_ptr() if mask is not None else None, dqkv.data_ptr(), G, S, H, Dh, scale, causal, None, None, None, None, 0, BF16, ws.data_ptr(), wsb, st()))
        t = timeit(f)
        print(f"  bwd G={G} S={S} causal={causal}: {t*1e3:7.3f} ms  {2.5*fl/t/1e12:6.1f} TFLOP/s (full-square count)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("norm", "all"):
        bench_norm()
    if what in ("attn", "all"):
        bench_attn()
