#!/usr/bin/env python3
"""dev probe: the streaming NT GEMM with and without the residual epilogue at the step's text / vision shapes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from meant_amd import ops
dev = torch.device("cuda")
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (786432, 301056):
    x = torch.randn(M, 768, device=dev).bfloat16(); w = torch.randn(768, 768, device=dev) * 0.03; b = torch.randn(768, device=dev)
    res = torch.randn(M, 768, device=dev).bfloat16()
    with torch.no_grad():
        t0 = timeit(lambda: ops.linear(x, w, b))
        t1 = timeit(lambda: ops.linear(x, w, b, res))
    fl = 2 * M * 768 * 768 / 1e9
    print(f"M={M}: plain {t0:.3f} ms ({fl / t0:.0f} TFLOP/s)  residual {t1:.3f} ms ({fl / t1:.0f} TFLOP/s)  +{(t1 / t0 - 1) * 100:.0f} %", flush=True)
