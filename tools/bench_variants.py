#!/usr/bin/env python3
"""The model variants of SURVEY.md 8(d) other than the headline (secondary figures; bench.py's line stays the metric):
  C2  meant_vision(768, 4, 224, 224, 16, lag=1, 2 classes, 12 heads, E=1)        images (B, 1, 4, 224, 224)
  C5  meant_vqa(768, 768, 4, 224, 224, 16, lag=1, 3129 classes, V=64001, 12 heads, E=1)   tweets (B, 512), images (B, 4, 224, 224)
  T   meant_tweet(768, 4, lag=12, 2 classes, V=64001, 12 heads, E=1)             tweets (B, 12, 512)
forward + CE on the probabilities + backward, bf16 tier, train mode, one MI355X.
    python tools/bench_variants.py [--heads 12] [--steps 5]
Prints one JSON line per variant."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def timeit(step, steps):
    for _ in range(10):                                  # the first steps of a process also load code objects and ramp the clocks
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import meant_amd as M
    from meant_amd.train import cross_entropy_on_probs
    dev = torch.device("cuda")
    rs = np.random.RandomState(0)
    H, V, d = args.heads, 64001, 768

    def run(name, model, inputs, ncls, B, gflop_fwd):
        model = model.to(dev).train()
        model.compute_dtype = torch.bfloat16
        target = torch.from_numpy(rs.randint(0, ncls, (B,))).to(dev)

        def step():
            for p in model.parameters():
                p.grad = None
            loss = cross_entropy_on_probs(model(*inputs), target)
            loss.backward()
            return loss
        dt = timeit(step, args.steps)
        print(json.dumps({"variant": name, "heads": H, "batch": B, "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(B / dt, 1),
                          "tflops_algorithmic": round(3 * gflop_fwd * B / dt / 1e3, 1)}), flush=True)
        del model
        torch.cuda.empty_cache()

    # per-sample forward GFLOP (patch-embed 2 N P d, vision layer 16 d^2 N + 4 N^2 d, language layer 16 d^2 S + 4 S^2 d)
    N, P, S = 196, 1024, 512
    g_patch, g_vis, g_lang = 2 * N * P * d / 1e9, (16 * d * d * N + 4 * N * N * d) / 1e9, (16 * d * d * S + 4 * S * S * d) / 1e9
    B = 256
    img = torch.randn(B, 1, 4, 224, 224, device=dev, dtype=torch.bfloat16)
    run("C2 meant_vision", M.meant_vision(d, 4, 224, 224, 16, 1, 2, num_heads=H, num_encoders=1, channels=4), (img,), 2, B, g_patch + g_vis)
    del img
    B = 128
    tw = torch.from_numpy(rs.randint(0, V, (B, S))).to(dev)
    img = torch.randn(B, 4, 224, 224, device=dev, dtype=torch.bfloat16)
    mask = torch.ones(B, S, device=dev)
    mask[:, 400:] = 0
    run("C5 meant_vqa", M.meant_vqa(d, d, 4, 224, 224, 16, 1, 3129, torch.nn.Embedding(V, d), num_heads=H, num_encoders=1), (tw, img, mask), 3129, B,
        g_patch + g_vis + g_lang)
    del img
    B, L = 128, 12
    tw = torch.from_numpy(rs.randint(0, V, (B, L, S))).to(dev)
    mask = torch.ones(B, L, S, device=dev)
    mask[:, :, 400:] = 0
    run("meant_tweet lag=12", M.meant_tweet(d, 4, L, 2, torch.nn.Embedding(V, d), num_heads=H, num_encoders=1), (tw, mask), 2, B, L * g_lang)


if __name__ == "__main__":
    main()
