#!/usr/bin/env python3
"""probe_corun.py with a THIN co-runner (tools/lab/thin.hip: persistent, one 4-wave workgroup per CU, no LDS, <= 64 VGPRs): does an
HBM-bound pass written to fit beside the streaming GEMM's workgroup overlap with it?
    python tools/probe_corun2.py [--blocks 256]"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--n", type=int, default=768)
    args = ap.parse_args()
    from meant_amd import ops
    thin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "thin.so"))
    thin.thin_copy_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda")
    M = 786432
    x = torch.randn(M, 768, device=dev).bfloat16()
    w = (torch.randn(args.n, 768, device=dev) * 0.03)
    src = torch.randn(M, 768, device=dev).bfloat16()
    dst = torch.empty_like(src)
    n16 = src.numel() * 2 // 16
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def gemm():
        with torch.no_grad():
            for _ in range(args.iters):
                ops.linear(x, w)

    def copy():
        for _ in range(args.iters):
            thin.thin_copy_launch(src.data_ptr(), dst.data_ptr(), n16, args.blocks, torch.cuda.current_stream().cuda_stream)

    def timed(fa, fb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fb:
            with torch.cuda.stream(sb):
                fb()
        if fa:
            with torch.cuda.stream(sa):
                fa()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(2):
        timed(gemm, copy)
    ta, tb, tab = timed(gemm, None), timed(None, copy), timed(gemm, copy)
    gb = args.iters * M * 768 * 4 / 1e9
    print(f"blocks {args.blocks}: GEMM alone {ta:.2f} ms ({args.iters * 2 * M * args.n * 768 / ta / 1e9:.0f} TFLOP/s) | thin copy alone {tb:.2f} ms "
          f"({gb / tb * 1e3:.0f} GB/s) | together {tab:.2f} ms (sum {ta + tb:.2f}, max {max(ta, tb):.2f})")


if __name__ == "__main__":
    main()
