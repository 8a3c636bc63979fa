#!/usr/bin/env python3
"""Divided space-time attention backbone (src/meant/timesformer_pytorch.py) on one MI355X: forward +
backward of meant_amd.TimeSformer.meant_forward at MEANT's image geometry, bf16 tier.
    python tools/bench_timesformer.py [--batch 16] [--depth 1] [--frames 12] [--steps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--depth", type=int, default=1)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import meant_amd as M
    dev = torch.device("cuda")
    m = M.TimeSformer(dim=768, num_frames=args.frames, num_classes=2, image_size=224, patch_size=16, channels=4, depth=args.depth,
                      heads=12, dim_head=64).to(dev)
    m.compute_dtype = torch.bfloat16
    video = torch.randn(args.batch, args.frames, 4, 224, 224, device=dev)

    grad = None

    def step():
        nonlocal grad
        for p in m.parameters():
            p.grad = None
        x = m.meant_forward(video)
        if grad is None:                                 # a fixed upstream gradient: the timed region is the backbone's fwd + bwd only
            grad = torch.randn_like(x) / x.numel() ** 0.5
        x.backward(grad)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    n, f, d = 196, args.frames, 768
    T = args.batch * (1 + f * n)
    flops_layer = 2 * T * d * (2 * 3 * d + 2 * d + 8 * d + 4 * d) + 4 * args.batch * d * (n * f * (f + 1) + f * n * (n + 1))
    flops = 3 * (2 * T * 1024 * d + args.depth * flops_layer)
    print(json.dumps({"what": "TimeSformer.meant_forward fwd+bwd, bf16", "batch": args.batch, "frames": f, "depth": args.depth,
                      "ms_per_step": round(dt * 1e3, 2), "videos_per_s": round(args.batch / dt, 1), "tflops_algorithmic": round(flops / dt / 1e12, 1)}))


if __name__ == "__main__":
    main()
