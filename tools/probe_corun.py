#!/usr/bin/env python3
"""Can an HBM-bound kernel run UNDER the streaming GEMM?  The GEMM holds every CU with one 8-wave workgroup (160 KiB of LDS, 2 x 224
VGPRs per SIMD lane), which leaves 64 VGPRs per lane and no LDS: a kernel that needs no more than that can be co-resident as a
third wave per SIMD, one that needs more waits for the GEMM's workgroups to leave.  Times, on two HIP streams, the (786432, N, 768)
GEMM and an elementwise cast (12-22 VGPRs, no LDS) alone and together.
    python tools/probe_corun.py [--n 768] [--iters 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=768)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from meant_amd import ops
    dev = torch.device("cuda")
    M = 786432
    x = torch.randn(M, 768, device=dev).bfloat16()
    w = (torch.randn(args.n, 768, device=dev) * 0.03)
    src = torch.randn(M, 768, device=dev).bfloat16()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def gemm():
        with torch.no_grad():
            for _ in range(args.iters):
                ops.linear(x, w)

    def copy():
        for _ in range(args.iters):
            ops.cast(src, torch.float32)

    def timed(fa, fb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(2):
        timed(gemm, copy)
    ta, tb, tab = timed(gemm, None), timed(None, copy), timed(gemm, copy)
    gb = args.iters * M * 768 * 6 / 1e9
    print(f"GEMM alone {ta:.2f} ms ({args.iters * 2 * M * args.n * 768 / ta / 1e9:.0f} TFLOP/s) | cast alone {tb:.2f} ms ({gb / tb:.0f} GB/s) | "
          f"together {tab:.2f} ms (sum {ta + tb:.2f}, max {max(ta, tb):.2f})")


if __name__ == "__main__":
    main()
