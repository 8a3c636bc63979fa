#!/usr/bin/env python3
"""MLM pretrainer step (pretrain_mlm.py:74-88,:144-166) on one MI355X: forward + CrossEntropyLoss over the
V = 64001 vocabulary + backward, bf16 tier, synthetic token ids with 15 % of the positions labelled.
    python tools/bench_mlm.py [--encoders 12] [--heads 12] [--batch 64] [--seq 512] [--steps 5]
Prints one JSON line (secondary figure; the headline metric stays bench.py's)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoders", type=int, default=12)
    ap.add_argument("--heads", type=int, default=12, help="12 -> Dh = 64 (flash path); the reference's default 8 -> Dh = 96 (zero-padded to the 128-wide kernels)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--vocab", type=int, default=64001)
    args = ap.parse_args()
    import meant_amd as M
    from transformers import RobertaConfig, RobertaForMaskedLM
    dev = torch.device("cuda")
    cfg = RobertaConfig(vocab_size=args.vocab, hidden_size=768, num_hidden_layers=1, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=args.seq + 2, pad_token_id=1, type_vocab_size=1, layer_norm_eps=1e-5)
    rob = RobertaForMaskedLM._from_config(cfg)
    model = M.meant_language_pretrainer(args.encoders, 768, rob.roberta.embeddings, rob.lm_head, text_dim=768, num_heads=args.heads).to(dev)
    model.compute_dtype = torch.bfloat16
    model.train()
    rs = np.random.RandomState(0)
    B, S, V = args.batch, args.seq, args.vocab
    ids = torch.from_numpy(rs.randint(2, V, (B, S))).to(dev)
    mask = torch.ones(B, S, device=dev)
    labels = torch.full((B, S), -100, dtype=torch.int64)
    pick = torch.from_numpy(rs.rand(B, S) < 0.15)
    labels[pick] = torch.from_numpy(rs.randint(2, V, int(pick.sum())))
    labels = labels.to(dev)

    # gradients go where meant_amd.train.TrainStep puts them: flat fp32 buckets, written by the backward kernels directly
    from meant_amd.parallel import GradReducer
    red = GradReducer(model.parameters(), direct_grads=True)

    def step():
        red.prepare()
        loss = model.loss(ids, mask, labels)
        loss.backward()
        red.wait()
        return loss

    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    assert torch.isfinite(loss).item()
    T = B * S
    flops = 3 * (args.encoders * (16 * 768 * 768 * T + 4 * S * 768 * T) + 2 * 768 * 768 * T + 2 * 768 * V * T)
    print(json.dumps({"what": "MLM pretrainer fwd + CE(V) + bwd, bf16", "encoders": args.encoders, "heads": args.heads, "batch": B, "seq": S,
                      "vocab": V, "ms_per_step": round(dt * 1e3, 2), "tokens_per_s": round(T / dt, 1),
                      "tflops_algorithmic": round(flops / dt / 1e12, 1), "loss": round(loss.item(), 4)}))


if __name__ == "__main__":
    main()
