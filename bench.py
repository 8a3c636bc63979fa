#!/usr/bin/env python3
"""Headline benchmark: samples/sec, forward+backward, full MEANT (tweet + image), lag=12, d=768, 12 heads,
512-token text, 224x224 p=16 patches, E=1 encoder layer, bf16 compute (fp32 master weights, fp32
statistics/softmax/accumulators), synthetic inputs, random-init weights -- BASELINE.json configs[2]/[3].

    python bench.py --gpus 1 --steps K --warmup W                       # one MI355X
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                            # N ranks, RCCL grad all-reduce

`--model meant_vqa` (BASELINE.json configs[4]: image + text, no lag axis, 3129 classes) and `--model meant_vision`
(configs[1]: images only, lag 1) run the same step with the same JSON contract on those classes; the default is the
headline `meant`.

A step = one pass of the hot path over one batch: forward (train mode: the languageEncoder's Dropout(0.5)
is live, fused into the RMSNorm kernel), cross-entropy on the probabilities (in_loop_train.py:232),
backward to every parameter gradient (incl. the 64001x768 embedding), and for N>1 the bucketed
gradient all-reduce overlapped with backward.  No optimizer step (the metric is fwd+bwd).
Weak scaling: 128 samples per GPU (global batch 1024 at 8 GPUs, BASELINE.json configs[3]).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- the dominant kernel (gemm_bf16_nt256p_kernel, MFMA-bound): algorithmic FLOPs of its launches / their
                  HIP-event durations.  The timed steps run the two encoder stacks on two HIP streams, where a launch
                  shares the CUs with the other stream's kernels and has no duration of its own; the events are
                  therefore taken in `roofline.timed_in` extra single-stream steps right after the timed region (the
                  figure measured inside it is reported beside it).  `roofline.others` carries the same for the
                  other hot kernels: dW GEMM and attention against the MFMA peak, RMSNorm / attention against HBM;
  cpu_baseline -- the CPU oracle (a port: the reference is Python and cannot travel) timed on the host cores on a bounded
                  sample of the same workload (rank 0, N=1 only): batch 1 and batch 8 on all the box's threads, batch 1
                  on one thread.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# A gradient collective beside the backward pass (WORLD_SIZE > 1, or the one-GPU rehearsal MEANT_REDUCE_ALWAYS=1): give every HIP
# stream a hardware queue of its own BEFORE the runtime starts.  On the default four queues the reducer's launch stream, RCCL's own
# streams and the step's compute streams alias each other by creation order (rocprofv3 --kernel-trace: main and a stack's stream on
# one queue); measured on one MI355X with a one-rank RCCL group: 38.3 ms per step with eight queues against 39.5 ms with four
# (plain step without collectives: 37.7 ms; profiles/r04_stream_queues.txt).
if int(os.environ.get("WORLD_SIZE", "1") or "1") > 1 or os.environ.get("MEANT_REDUCE_ALWAYS") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16 peak, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0         # HBM3E, MI355X_MICROARCH.md (~8 TB/s; ~6.3 TB/s is what a streaming kernel reaches)
V, D, H, L, S, IMG, P, C, NCLS = 64001, 768, 12, 12, 512, 224, 16, 4, 2


MODEL = "meant"                # --model: meant (configs[2]/[3], the headline) | meant_vqa (configs[4]) | meant_vision (configs[1])


def _shape():
    """(lag, text tokens per lag step, patches per lag step, classes, width of the fused feature) of the benchmarked model"""
    n = (IMG // P) ** 2
    if MODEL == "meant":
        return L, S, n, NCLS, 2 * D
    if MODEL == "meant_vqa":
        return 1, S, n, 3129, 2 * D
    return 1, 0, n, NCLS, D        # meant_vision


def flops_per_sample(E: int) -> float:
    """SURVEY.md 8(d), the work of the reference's own graph, fwd+bwd = 3 x fwd, full-square attention.  Per lag step:
    patch-embed 2 n P d; a vision layer 16 d^2 n + 4 n^2 d; a language layer 16 d^2 S + 4 S^2 d; then the temporal encoder
    (`meant`: 2 D_t^2 (3 lag + 3); the single-modality / vqa classes: their own small heads).  `meant` at lag 12:
    fwd = 3.88 + 91.26 E GFLOP."""
    lag, s, n, ncls, dt = _shape()
    fwd = lag * (2.0 * n * (C * P * P) * D + E * (16.0 * D * D * n + 4.0 * n * n * D) + (E * (16.0 * D * D * s + 4.0 * s * s * D) if s else 0.0))
    fwd += 2.0 * dt * dt * (3 * lag + 3) if MODEL != "meant_vqa" else 0.0
    fwd += 2.0 * dt * ncls
    return 3.0 * fwd


def flops_per_sample_executed(E: int) -> float:
    """what the kernels here execute: the encoder's Linear(d,d) that feeds q/k/v is composed into the projection
    (meant_amd.modules.COMPOSE_PRE_LINEAR), which removes 2 d^2 FLOPs per token per layer in forward and twice
    that in backward, and the stacks' final Linear commutes with the mean-pool; everything else as above."""
    import meant_amd.modules as mm
    lag, s, n, _, _ = _shape()
    per_linear = 3.0 * 2.0 * D * D * lag * (s + n)                       # one Linear(d, d) on every token: fwd + dX + dW
    saved = E * per_linear if mm.COMPOSE_PRE_LINEAR else 0.0
    # the last Linear of the last layer of each stack is evaluated on the pooled features (meant_amd.modules.POOL_LAST_LINEAR:
    # mean_s(h W^T + b + x) = mean_s(h) W^T + b + mean_s(x)); what replaces it is S (resp. 196) times smaller
    if mm.POOL_LAST_LINEAR:
        saved += per_linear - 3.0 * 2.0 * D * D * lag * (2 if s else 1)
    return flops_per_sample(E) - saved


def _newest_profile(pattern):
    """newest tracked counter table under profiles/ matching the glob (names are rNN_..., so the sort order is the round order)"""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return hits[-1] if hits else None


def _dominant_kernel():
    from meant_amd import _lib
    return "gemm_bf16_nt256p_kernel" if _lib.get_option("nt_pp") else "gemm_bf16_nt256s_kernel"


class GemmTimer:
    """HIP-event timing of every launch of the dominant kernel (gemm_bf16_nt256p_kernel: the streaming bf16 NT GEMM behind
    Linear forward, the fused q|k|v projection and every input-gradient) during the timed steps.  Events are recorded on the stream the kernel is
    launched on (torch's current stream, which the C ABI receives)."""

    def __init__(self):
        self.recs = []
        self.other = []          # (kind, e0, e1, flops, bytes) of the other hot kernels
        self.enabled = False
        self.others_enabled = False     # the other kernels are timed in the single-stream steps only

    def install(self):
        from meant_amd import ops
        lib = ops.lib
        timer = self

        def wrap(name, flops_of, key_of, kind="nt", bytes_of=None):
            orig = getattr(lib, name)

            def call(*a):
                if not timer.enabled or (kind != "nt" and not timer.others_enabled):
                    return orig(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = orig(*a)
                e1.record()
                if kind == "nt":
                    timer.recs.append((e0, e1, flops_of(a), key_of(a)))
                else:
                    timer.other.append((kind, e0, e1, flops_of(a) if flops_of else 0.0, bytes_of(a) if bytes_of else 0.0))
                return rc
            return call

        class LibProxy:
            def __getattr__(self_, n):
                return getattr(lib, n)
        proxy = LibProxy()
        # the launches that gemm_bf16_nt_launch routes to the streaming 256 x 256 kernel (gemm_bf16_nt256p_kernel): bf16, M >= 1024, M % 256 == 0, N % 256 == 0,
        # K % 64 == 0, K >= 128, and at least half a chip of 256 x 256 tiles (fewer go to the 128 x 128 kernel)
        ncu = torch.cuda.get_device_properties(0).multi_processor_count

        def nt256(M, N, K, dtype):
            ok = dtype == 1 and M >= 1024 and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K >= 128 and (M // 256) * (N // 256) * 2 >= ncu
            return 2.0 * M * N * K if ok else 0.0
        # meant_linear_fwd(x, ldx, w, bias, res, ldr, y, ldy, pre, M, N, K, epi, dtype, stream)
        proxy.meant_linear_fwd = wrap("meant_linear_fwd", lambda a: nt256(a[9], a[10], a[11], a[13]),
                                      lambda a: (a[9], a[10], a[11], 2.0 * a[9] * a[10] * ((1 if a[4] else 0) + (1 if a[8] else 0))))
        # meant_linear_bwd_dx(dy, lddy, wT, dx, lddx, M, N, K, dtype, stream): C[M,K] = dy[M,N] wT[K,N]^T
        proxy.meant_linear_bwd_dx = wrap("meant_linear_bwd_dx", lambda a: nt256(a[5], a[7], a[6], a[8]), lambda a: (a[5], a[7], a[6], 0.0))
        # meant_qkv_proj_fwd(x, ldx, w, bias, qkv, M, K, S, H, Dh, R, qa, qb, ka, kb, dtype, stream)
        proxy.meant_qkv_proj_fwd = wrap("meant_qkv_proj_fwd", lambda a: nt256(a[5], 3 * a[8] * a[9], a[6], a[15]),
                                        lambda a: (a[5], 3 * a[8] * a[9], a[6], 0.0))
        # the other hot kernels, with their algorithmic work: FLOPs (2 m n k; attention full-square:
        # 4 S^2 Dh per (group, head) forward, 2.5 x that backward) and / or HBM bytes (every operand once)
        es = 2.0                                                                   # bf16 activations
        # meant_linear_bwd_dw(dy, lddy, x, ldx, dw, db, M, N, K, dtype, ws, wsb, stream)
        proxy.meant_linear_bwd_dw = wrap("meant_linear_bwd_dw", lambda a: 2.0 * a[6] * a[7] * a[8] if (a[9] == 1 and a[6] >= 4096 and a[7] % 256 == 0 and a[8] % 256 == 0) else 0.0,
                                         None, "dw_gemm (gemm_bf16_tn256p_kernel)")
        # meant_attn_fwd(qkv, o, lse, km, G, S, H, Dh, scale, causal, dtype, ws, wsb, stream)
        proxy.meant_attn_fwd = wrap("meant_attn_fwd", lambda a: 4.0 * a[4] * a[6] * a[5] * a[5] * a[7], None, "attn_fwd",
                                    lambda a: 4.0 * a[4] * a[5] * a[6] * a[7] * es)
        # meant_attn_bwd(qkv, o, do, lse, km, dqkv, G, S, H, Dh, ...)
        # (the whole call: one single-pass kernel at the step's shapes (csrc/attn_bwd1.hip), the dQ + dK/dV pair otherwise; FLOPs are the
        # full-square 5-product count of SURVEY 8d either way; bytes: 9 token tensors as SURVEY 8d prices it)
        proxy.meant_attn_bwd = wrap("meant_attn_bwd", lambda a: 10.0 * a[6] * a[8] * a[7] * a[7] * a[9], None, "attn_bwd",
                                    lambda a: 9.0 * a[6] * a[7] * a[8] * a[9] * es)
        # meant_rmsnorm_fwd(x, scale, y, rinv, rows, d, ...): read x, write y
        proxy.meant_rmsnorm_fwd = wrap("meant_rmsnorm_fwd", None, None, "rmsnorm_fwd", lambda a: 2.0 * a[4] * a[5] * es)
        # meant_rmsnorm_fwd_pooled(x, scale, y, rinv, pooled, rows, d, group_rows, pool_input, ...): read x (+ write y when pool_input)
        proxy.meant_rmsnorm_fwd_pooled = wrap("meant_rmsnorm_fwd_pooled", None, None, "rmsnorm_fwd_pooled",
                                              lambda a: (2.0 if a[8] else 1.0) * a[5] * a[6] * es)
        # meant_rmsnorm_bwd(dy, x, scale, rinv, dx, dscale, rows, d, eps, p, seed, dres, gelu_pre, ...): read dy, x (+ dres, gelu_pre), write dx
        proxy.meant_rmsnorm_bwd = wrap("meant_rmsnorm_bwd", None, None, "rmsnorm_bwd",
                                       lambda a: (3.0 + (1 if a[11] else 0) + (1 if a[12] else 0)) * a[6] * a[7] * es)
        # meant_rmsnorm_bwd_pooled(dy, dy_pooled, x, scale, rinv, dx, dscale, rows, d, group_rows, eps, p, seed, dres, dres_pooled, gelu_pre, ...):
        # write dx; read x (unless formed from gelu_pre), dy / dres unless pooled, gelu_pre
        proxy.meant_rmsnorm_bwd_pooled = wrap("meant_rmsnorm_bwd_pooled", None, None, "rmsnorm_bwd_pooled",
                                              lambda a: (1.0 + (1 if a[2] else 0) + (0 if a[1] else 1) + (1 if (a[13] and not a[14]) else 0) + (1 if a[15] else 0)) * a[7] * a[8] * es)
        ops.lib = proxy

    def others_summary(self):
        """per kind: launches, mean ms, achieved TFLOP/s and GB/s of the algorithmic work, fractions of the peaks"""
        agg = {}
        for kind, e0, e1, f, b in self.other:
            if f <= 0 and b <= 0:
                continue
            t = e0.elapsed_time(e1) * 1e-3
            a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += t; a[2] += f; a[3] += b
        out = {}
        for kind, (n, t, f, b) in agg.items():
            ent = {"launches": n, "avg_ms": round(t / n * 1e3, 4)}
            if f > 0:
                ent["tflops"] = round(f / t / 1e12, 1)
                ent["mfma_frac"] = round(f / t / 1e12 / PEAK_BF16_TFLOPS, 4)
            if b > 0:
                ent["gb_per_s"] = round(b / t / 1e9, 1)
                ent["hbm_frac"] = round(b / t / 1e9 / PEAK_HBM_GBS, 4)
            out[kind] = ent
        return out

    @staticmethod
    def pmc_ratios():
        """measured HBM bytes (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes: profiles/r03_c_hbm_traffic_norm_attention.json,
        produced by tools/pmc_step_kernels.py + tools/parse_pmc_kernels.py) over the algorithmic bytes that `others` prices each
        kernel at, per kind and shape.  A lookup in tracked counter files, like `traffic` -- None when they are missing."""
        path = _newest_profile("r*_hbm_traffic_norm_attention.json")
        if path is None:
            return None
        doc = json.load(open(path))
        k = doc["kernels"]
        # the table was measured on one version of the kernels' sources: say so when a source has moved on since
        import hashlib
        stale = None
        if doc.get("source_sha16"):
            stale = any(not os.path.exists(os.path.join(ROOT, "meant_amd", "csrc", f)) or
                        hashlib.sha256(open(os.path.join(ROOT, "meant_amd", "csrc", f), "rb").read()).hexdigest()[:16] != h
                        for f, h in doc["source_sha16"].items())
        def units(sub, tag, lo, hi):
            for name, v in k.items():
                if sub in name and f"[{tag}," in name and lo <= v["in_units_of_one_token_tensor"] <= hi:
                    return v["in_units_of_one_token_tensor"]
            return None
        out = {}
        for tag in ("text", "vision"):
            ent = {}
            # algorithmic figures in token tensors: norm fwd 2, bwd 3 (+1 with the residual gradient or the GELU pre-activation),
            # attention fwd 4, bwd 9 (SURVEY 8d prices ONE pass; the two-pass scheme here reads q, k, v, dO twice: 12)
            for kind, sub, alg, lo, hi in (("rmsnorm_fwd", "rmsnorm_fwd_packed", 2.0, 1.9, 2.2), ("rmsnorm_bwd (+dres)", "rmsnorm_bwd_packed_kernelIDF16bLi3ELi0ELb0E", 4.0, 3.9, 4.2),
                                           ("rmsnorm_bwd", "rmsnorm_bwd_packed_kernelIDF16bLi3ELi0ELb0E", 3.0, 2.9, 3.2), ("attn_fwd", "attn_fwd_kernel", 4.0, 3.5, 6.0)):
                u = units(sub, tag, lo, hi)
                if u is not None:
                    ent[kind] = round(u / alg, 3)
            one = units("attn_bwd1", tag, 7.0, 14.0)
            if one:                                          # the single-pass backward: q|k|v, dO, O read once, dq|dk|dv written = 8 token tensors
                ent["attn_bwd (single pass)"] = round(one / 8.0, 3)
            dq, dkv = units("attn_bwd_dq", tag, 5.0, 9.0), units("attn_bwd_dkv", tag, 5.0, 9.0)
            if dq and dkv:
                ent["attn_bwd (dq + dkv, option attn_bwd1 = 0)"] = round((dq + dkv) / 9.0, 3)
                ent["attn_bwd two-pass vs its minimum of 12"] = round((dq + dkv) / 12.0, 3)
            out[tag] = ent
        return {"measured_over_algorithmic_hbm_bytes": out, "source": os.path.relpath(path, ROOT),
                "table_older_than_kernel_sources": stale}

    def summary(self, recs=None):
        tot_t, tot_f, n = 0.0, 0.0, 0
        for e0, e1, f, _ in (self.recs if recs is None else recs):
            if f <= 0:
                continue
            tot_t += e0.elapsed_time(e1) * 1e-3
            tot_f += f
            n += 1
        return n, tot_f, tot_t

    def traffic_per_launch(self):
        """HBM bytes per launch of the dominant kernel, averaged over the launches timed above, from the PMC table
        in the newest profiles/rNN_nt256*_hbm_traffic.json (rocprofv3 FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
        corrections applied; measured on the plain epilogue) plus the algorithmic bytes of the extra epilogue
        operands (residual read / pre-activation write).  None if a launched shape is not in the table."""
        path = _newest_profile("r*_nt256*_hbm_traffic.json")
        if path is None:
            return None
        self.traffic_path = os.path.relpath(path, ROOT)
        doc = json.load(open(path))
        table = doc["shapes"]
        # the table was measured on one version of the kernel's source: say so when the source has moved on since
        import hashlib
        src = os.path.join(ROOT, "meant_amd", "csrc", "gemm_bf16.hip")
        self.traffic_stale = bool(doc.get("kernel_source_sha16")) and os.path.exists(src) and \
            hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] != doc["kernel_source_sha16"]
        tot, n = 0.0, 0
        for _, _, f, key in self.recs:
            if f <= 0:
                continue
            M, N, K, extra = key
            ent = table.get(f"{M},{N},{K}")
            if ent is None:
                return None
            tot += ent["hbm_bytes"] + extra
            n += 1
        return tot / n if n else None


def _fold_state():
    """which encoder layers folded encode2[0] into encode2[1]'s GEMM during the run (meant_amd.ops.fold_wanted: by the size of
    the stack unless forced): False / True / "mixed" (one stack did, the other did not)"""
    from meant_amd import ops
    sep, folded = ops.fold_calls
    return "mixed" if sep and folded else bool(folded)


def build_model(E: int, device):
    import meant_amd
    torch.manual_seed(1234)
    if MODEL == "meant":
        m = meant_amd.meant(D, D, 4, IMG, IMG, P, L, NCLS, torch.nn.Embedding(V, D), num_heads=H, num_encoders=E, channels=C)
    elif MODEL == "meant_vqa":          # meant_vqa(768, 768, 4, 224, 224, 16, lag=1, num_classes=3129, ...)
        m = meant_amd.meant_vqa(D, D, 4, IMG, IMG, P, 1, 3129, torch.nn.Embedding(V, D), num_heads=H, num_encoders=E, channels=C)
    else:                               # meant_vision(768, 4, 224, 224, 16, lag=1, num_classes=2, ...)
        m = meant_amd.meant_vision(D, 4, IMG, IMG, P, 1, NCLS, num_heads=H, num_encoders=E, channels=C)
    m.compute_dtype = torch.bfloat16
    return m.to(device)


def make_batch(B: int, rank: int, device):
    """(model inputs, target): synthetic, seeded per rank (rank r holds rows [B r, B r + B) of the global batch)"""
    lag, s, n, ncls, _ = _shape()
    rs = np.random.RandomState(99 + rank)
    gen = torch.Generator(device=device).manual_seed(99 + rank)
    if MODEL == "meant_vision":
        images = torch.randn(B, 1, C, IMG, IMG, device=device, generator=gen)
        target = torch.from_numpy(rs.randint(0, ncls, (B,)).astype("int64")).to(device)
        return (images,), target
    tweets = torch.from_numpy(rs.randint(0, V, (B, lag, S)).astype("int64")).to(device)
    images = torch.randn(B, lag, C, IMG, IMG, device=device, generator=gen)
    mask = torch.ones(B, lag, S)
    pad = rs.randint(0, 384, (B, lag))
    for b in range(B):
        for l in range(lag):
            if pad[b, l]:
                mask[b, l, S - pad[b, l]:] = 0
    target = torch.from_numpy(rs.randint(0, ncls, (B,)).astype("int64")).to(device)
    if MODEL == "meant_vqa":            # no lag axis (meant/meant_vqa.py:205)
        tweets, images, mask = tweets[:, 0], images[:, 0], mask[:, 0]
    return (tweets, images, mask.to(device)), target


def cpu_baseline(E: int):
    """the oracle on the host cores: same config, fp32 eager, eval mode, fwd + CE + bwd.  SURVEY 8(d): batch 1 and batch 8
    on the box's threads (16 for a 1-GPU box), and batch 1 on a single thread.  Bounded: ~30 s in all."""
    from oracle import meant_oracle as O
    torch.manual_seed(0)
    cores = min(os.cpu_count() or 1, 16)            # the 1-GPU box gives a 16-thread CPU share
    lag, s, n, ncls, _ = _shape()
    if MODEL == "meant":
        m = O.meant(D, D, 4, IMG, IMG, P, L, NCLS, torch.nn.Embedding(V, D), num_heads=H, num_encoders=E, channels=C)
    elif MODEL == "meant_vqa":
        m = O.meant_vqa(D, D, 4, IMG, IMG, P, 1, 3129, torch.nn.Embedding(V, D), num_heads=H, num_encoders=E, channels=C)
    else:
        m = O.meant_vision(D, 4, IMG, IMG, P, 1, NCLS, num_heads=H, num_encoders=E, channels=C)
    m = m.eval()
    O.fill_weights_(m, 1234)
    rs = np.random.RandomState(99)

    def run(B, threads, warm, reps):
        torch.set_num_threads(threads)
        ids = torch.from_numpy(rs.randint(0, V, (B, lag, S)).astype("int64"))
        img = torch.from_numpy(rs.standard_normal((B, lag, C, IMG, IMG)).astype("float32"))
        mask = torch.ones(B, lag, S)
        mask[:, :, 400:] = 0
        tgt = torch.from_numpy(rs.randint(0, ncls, (B,)).astype("int64"))
        inputs = {"meant": (ids, img, mask), "meant_vqa": (ids[:, 0], img[:, 0], mask[:, 0]), "meant_vision": (img,)}[MODEL]
        times = []
        for it in range(warm + reps):
            t0 = time.time()
            m.zero_grad(set_to_none=True)
            O.cross_entropy_on_probs(m(*inputs), tgt).backward()
            if it >= warm:
                times.append(time.time() - t0)
        return B / float(np.median(times)), len(times)

    small = MODEL != "meant"                        # a twelfth of the work per sample: more repetitions fit the same bound
    v1, n1 = run(1, cores, 2, 15 if small else 5)
    v8, n8 = run(8, cores, 1, 6 if small else 2)
    vs, ns = run(1, 1, 1, 6 if small else 2)
    return {"value": round(v1, 4), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": f"CPU oracle (fp32 eager restatement pinned to the reference's golden vectors), same {MODEL} config "
                      f"(lag={lag}, d=768, S=512, 224x224, E={E}), batch 1, fwd+CE+bwd, median of {n1} iterations after 2 warm-ups",
            "batch8": {"value": round(v8, 4), "cores": int(cores), "sample": f"same, batch 8, median of {n8} after 1 warm-up"},
            "single_thread": {"value": round(vs, 4), "cores": 1, "sample": f"same, batch 1 on one thread, median of {ns} after 1 warm-up"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", choices=["meant", "meant_vqa", "meant_vision"], default="meant",
                    help="meant = BASELINE.json configs[2]/[3] (the headline); meant_vqa = configs[4]; meant_vision = configs[1]")
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="samples per GPU (default: 128; meant_vision: 256)")
    ap.add_argument("--encoders", type=int, default=1)
    ap.add_argument("--heads", type=int, default=12, help="attention heads (12 = BASELINE.json; 8 = the reference classes' default, head dim 96)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="disable dropout (parity-mode numerics)")
    ap.add_argument("--two-streams", type=int, default=-1, help="override meant_amd.modules.TWO_STREAMS (0/1)")
    ap.add_argument("--with-optimizer", action="store_true", help="also time the step with clip + fused AdamW (extra field)")
    ap.add_argument("--forward-only", action="store_true", help="also time the eval-mode forward alone (serving-style secondary figure)")
    ap.add_argument("--checkpoint", action="store_true", help="model.activation_checkpointing = True: every encoder layer is recomputed in backward")
    ap.add_argument("--checkpoint-layers", type=int, default=0,
                    help="recompute only the first N layers of each stack (--encoders 12 at 128 samples per GPU fits the 288 GB with N = 4)")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="run the per-GPU batch as this many micro-batches per step (gradient accumulation in the reducer's buckets, one "
                         "collective per step): --encoders 12 keeps 128 samples per GPU per step with 64 in flight and no recomputation")
    ap.add_argument("--extra-streams", type=int, default=0,
                    help="lab: create (and touch once) this many more HIP streams before the model runs -- DESIGN.md section 7, the stream-count cliff")
    ap.add_argument("--fp32-batch", type=int, default=0,
                    help="also time fwd+CE+bwd of the fp32 tier (north_star's 1e-3 tolerance tier) at this batch (secondary field, never `value`)")
    ap.add_argument("--from-host", choices=["f64", "f32", "u8"], default=None,
                    help="also time the step fed by meant_amd.data.DeviceBatchLoader from host arrays of this pixel type "
                         "(PCIe-inclusive secondary figure, never `value`)")
    args = ap.parse_args()
    global H, MODEL
    H, MODEL = args.heads, args.model
    if args.batch_per_gpu <= 0:
        args.batch_per_gpu = 256 if MODEL == "meant_vision" else 128

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MEANT_REDUCE_ALWAYS=1: a one-rank RCCL group whose (trivial) collectives are really issued -- rehearsal on a one-GPU box
    if args.gpus > 1 or world > 1 or os.environ.get("MEANT_REDUCE_ALWAYS") == "1":
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # rehearsal knobs for a one-GPU box: MEANT_DIST_BACKEND=gloo and MEANT_ALL_RANKS_ON_GPU0=1 run every rank on
        # cuda:0 over gloo to exercise the N>1 code path; the real multi-GPU run uses RCCL ("nccl"), one GPU per rank
        backend = os.environ.get("MEANT_DIST_BACKEND", "nccl")
        if os.environ.get("MEANT_ALL_RANKS_ON_GPU0") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import meant_amd
    from meant_amd.parallel import GradReducer
    from meant_amd.train import FusedAdamW, cross_entropy_on_probs

    if args.two_streams >= 0:
        import meant_amd.modules as _mm
        _mm.TWO_STREAMS = bool(args.two_streams)
    extra_streams = [torch.cuda.Stream(device=dev) for _ in range(args.extra_streams)]
    for es in extra_streams:                        # a stream gets its hardware queue at first use
        with torch.cuda.stream(es):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    timer = GemmTimer()
    timer.install()
    E, B = args.encoders, args.batch_per_gpu
    model = build_model(E, dev)
    model.train(not args.eval_mode)
    model.activation_checkpointing = True if args.checkpoint else int(args.checkpoint_layers)
    if world > 1:                                   # identical replicas: broadcast rank 0's weights once
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    # the dropout seeds are drawn from torch's CPU generator (meant_amd.modules._seed): every rank gets its own stream of
    # masks, as independent replicas of nn.Dropout would have
    torch.manual_seed(1234 + rank)
    reducer = GradReducer(model.parameters(), bucket_mb=64.0, direct_grads=True)
    inputs, target = make_batch(B, rank, dev)
    step_events = []

    micro = max(1, args.micro_batches)
    assert B % micro == 0, "--micro-batches must divide the per-GPU batch"
    mb = B // micro
    chunks = [(tuple(t[i * mb:(i + 1) * mb] for t in inputs), target[i * mb:(i + 1) * mb]) for i in range(micro)]

    def step():
        reducer.prepare()
        if micro == 1:
            out = model(*inputs)
            loss = cross_entropy_on_probs(out, target)          # CE on the probabilities, as in_loop_train.py:232
            loss.backward()
        else:
            # gradient accumulation: every micro-batch adds its share of the batch-mean loss's gradient into the reducer's
            # buckets; only the last backward counts parameters and starts the bucket all-reduces (reducer.no_sync)
            for i, (inp, tgt) in enumerate(chunks):
                if i < micro - 1:
                    with reducer.no_sync():
                        loss = cross_entropy_on_probs(model(*inp), tgt) * (1.0 / micro)
                        loss.backward()
                else:
                    loss = cross_entropy_on_probs(model(*inp), tgt) * (1.0 / micro)
                    loss.backward()
        reducer.wait()
        if step_events is not None and timer.enabled and not iso:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append(ev)
        return loss

    iso = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    import meant_amd.modules as _mm
    timer.enabled = True
    timer.others_enabled = not _mm.TWO_STREAMS           # one stream: the timed steps themselves serve for every kernel
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-step durations from one HIP event per step on the main stream (no host sync inside the timed region): the median
    # beside the mean
    marks = [ev0] + step_events
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    timer.enabled = timer.others_enabled = False
    # The timed region runs the two encoder stacks on two HIP streams, so a launch of the dominant kernel shares the
    # CUs with whatever the other stream is running and its event-to-event time is not the kernel's own.  For the
    # roofline the same step is therefore run a few more times on ONE stream with the same event timers.
    import meant_amd.modules as _mm
    overlapped_recs, iso_steps = timer.recs, 0
    if _mm.TWO_STREAMS:                             # every rank: the step holds the gradient collective
        timer.recs, timer.other, iso_steps = [], [], min(args.steps, 4)
        iso = True
        _mm.TWO_STREAMS = False
        step()
        torch.cuda.synchronize()
        timer.enabled = timer.others_enabled = True
        for _ in range(iso_steps):
            step()
        torch.cuda.synchronize()
        timer.enabled = timer.others_enabled = False
        _mm.TWO_STREAMS = True
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(loss).item(), "loss is not finite"

    # secondary figure (not the headline metric): the same step plus global-norm clip + fused AdamW
    opt_ms = None
    if args.with_optimizer:
        opt = FusedAdamW(reducer, lr=5e-5, max_grad_norm=1.0)
        for _ in range(2):
            step(); opt.step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(); opt.step()
        barrier()
        opt_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # secondary figure: the eval-mode forward alone (what a deployment that only scores runs)
    fwd_ms = None
    if args.forward_only:
        was_training = model.training
        model.eval()
        with torch.no_grad():
            for _ in range(2):
                model(*inputs)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out = model(*inputs)
            barrier()
            fwd_ms = (time.perf_counter() - t1) / args.steps * 1e3
        assert torch.isfinite(out).all().item()
        model.train(was_training)

    # secondary figure: the fp32 tier (exact-f32 MFMA GEMMs, fp32 attention): fwd + CE + bwd at a small batch
    fp32_ms = None
    if args.fp32_batch > 0:
        Bf = min(args.fp32_batch, B)
        model.compute_dtype = torch.float32
        f_in, f_tgt = tuple(t[:Bf] for t in inputs), target[:Bf]

        def fstep():
            reducer.prepare()
            l_ = cross_entropy_on_probs(model(*f_in), f_tgt)
            l_.backward()
            reducer.wait()
            return l_
        for _ in range(2):
            fstep()
        barrier()
        t1 = time.perf_counter()
        nf = max(2, min(args.steps, 5))
        for _ in range(nf):
            l_ = fstep()
        barrier()
        fp32_ms = (time.perf_counter() - t1) / nf * 1e3
        assert torch.isfinite(l_).item()
        model.compute_dtype = torch.bfloat16

    # secondary figure: the same step fed from HOST arrays in the data set's storage type through the double-buffered
    # loader (gather into pinned memory, H2D on its own stream, conversion + normalisation + patchify on the device)
    host_ms = None
    if args.from_host is not None:
        assert MODEL == "meant", "--from-host feeds the lagged (graphs, tweets, masks) data set of in_loop_train.py: --model meant only"
        from meant_amd.data import DeviceBatchLoader
        nbatch = 3 + args.steps
        npdt = {"f64": np.float64, "f32": np.float32, "u8": np.uint8}[args.from_host]
        rs = np.random.RandomState(7 + rank)
        pool = 2 * B                                      # two batches' worth of distinct samples, visited repeatedly
        if npdt == np.uint8:
            h_graphs = rs.randint(0, 256, (pool, L, C, IMG, IMG)).astype(np.uint8)
        else:
            h_graphs = rs.standard_normal((pool, L, C, IMG, IMG)).astype(npdt)
        h_tweets = rs.randint(0, V, (pool, L, S)).astype(np.int64)
        h_masks = np.ones((pool, L, S), dtype=np.float32)
        h_labels = rs.randint(0, NCLS, (pool,)).astype(np.int64)
        if npdt == np.uint8:
            model.patchEmbed[0].set_normalization(127.5, 73.9)
        loader = DeviceBatchLoader(h_graphs, h_tweets, None, h_masks, h_labels, batch_size=B, device=dev,
                                   pin_source_bytes=(1 << 20) if os.environ.get("MEANT_PIN_SOURCE") else 0)

        def host_steps(n):
            done = 0
            while done < n:
                for g_, tw_, _, am_, y_ in loader:
                    reducer.prepare()
                    loss_ = cross_entropy_on_probs(model(tw_, g_, am_), y_)
                    loss_.backward()
                    reducer.wait()
                    done += 1
                    if done >= n:
                        break
        host_steps(2)
        barrier()
        t2 = time.perf_counter()
        host_steps(args.steps)
        barrier()
        host_ms = (time.perf_counter() - t2) / args.steps * 1e3
        model.patchEmbed[0].set_normalization(0.0, 1.0)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        sps = world * B * args.steps / elapsed
        n, gf, gt = timer.summary()
        achieved = gf / gt / 1e12 if gt > 0 else 0.0
        traffic = timer.traffic_per_launch() if (B == 128 and MODEL == "meant") else None   # the PMC table holds the headline's shapes
        roofline = {"kernel": _dominant_kernel(), "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                    "traffic": None if traffic is None else round(traffic),
                    "traffic_source": f"{getattr(timer, 'traffic_path', None)} (rocprofv3 PMC passes; lookup by launched shape)",
                    "traffic_table_older_than_kernel_source": getattr(timer, "traffic_stale", None),
                    "launches_timed": n, "avg_launch_ms": round(gt / max(n, 1) * 1e3, 4),
                    "avg_launch_gflop": round(gf / max(n, 1) / 1e9, 2),
                    "whole_step_mfma_frac": round(sps / world * flops_per_sample_executed(E) / (PEAK_BF16_TFLOPS * 1e12), 4),
                    # the same step priced at the reference graph's FLOPs (what a literal evaluation of the module list would
                    # execute; the pooled last Linear of each stack skips part of it, see DESIGN.md section 6 "Pooled tail")
                    "whole_step_mfma_frac_reference_graph": round(sps / world * flops_per_sample(E) / (PEAK_BF16_TFLOPS * 1e12), 4)}
        if iso_steps:
            n2, gf2, gt2 = timer.summary(overlapped_recs)
            roofline["timed_in"] = f"{iso_steps} extra single-stream steps after the timed region"
            roofline["achieved_while_sharing_cus_with_second_stream"] = round(gf2 / gt2 / 1e12, 1) if gt2 > 0 else None
        roofline["others"] = timer.others_summary()
        roofline["others_pmc"] = timer.pmc_ratios()
        lag_, s_, n_, ncls_, _ = _shape()
        metric = {"meant": "samples/sec fwd+bwd, MEANT lag=12 d=768", "meant_vqa": "samples/sec fwd+bwd, meant_vqa d=768 seq=512",
                  "meant_vision": "samples/sec fwd+bwd, meant_vision lag=1 d=768"}[MODEL]
        workload = {"meant": f"full MEANT (tweet+image) fwd+CE+bwd, lag=12, d=768, {H} heads, seq=512, 224x224 p=16, E={E}, vocab 64001 "
                             f"(BASELINE.json configs[2]/[3])",
                    "meant_vqa": f"meant_vqa (image+text, no lag axis) fwd+CE+bwd, d=768, {H} heads, seq=512, 224x224 p=16, E={E}, vocab 64001, "
                                 f"3129 classes (BASELINE.json configs[4])",
                    "meant_vision": f"meant_vision (images only) fwd+CE+bwd, lag=1, d=768, {H} heads, 224x224 p=16, E={E} "
                                    f"(BASELINE.json configs[1])"}[MODEL]
        res = {"metric": metric, "value": round(sps, 2), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "ms_per_step_median": round(float(np.median(step_ms)), 3) if step_ms else None,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": workload,
                          "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", "micro_batches": micro,
                          "train_mode_dropout": not args.eval_mode, "activation_checkpointing": True if args.checkpoint else (int(args.checkpoint_layers) or False),
                          "norm_linear_fold": _fold_state(), "grad_allreduce": reducer.active,
                          "gflop_per_sample": round(flops_per_sample(E) / 1e9, 1),
                          "gflop_per_sample_executed": round(flops_per_sample_executed(E) / 1e9, 1)},
               "roofline": roofline}
        if fwd_ms is not None:
            res["forward_only"] = {"ms_per_step": round(fwd_ms, 3), "samples_per_s": round(world * B / fwd_ms * 1e3, 2),
                                   "what": "eval-mode forward under no_grad, same batch"}
        if opt_ms is not None:
            res["with_optimizer"] = {"ms_per_step": round(opt_ms, 3), "samples_per_s": round(world * B / opt_ms * 1e3, 2),
                                     "what": "fwd+CE+bwd + global-norm clip(1.0) + fused AdamW on the flat fp32 buckets"}
        if fp32_ms is not None:
            Bf = min(args.fp32_batch, B)
            sps32 = world * Bf / fp32_ms * 1e3
            res["fp32_tier"] = {"batch_per_gpu": Bf, "ms_per_step": round(fp32_ms, 3), "samples_per_s": round(sps32, 2),
                                "whole_step_f32_matrix_frac": round(sps32 / world * flops_per_sample_executed(E) / 157.3e12, 4),
                                "what": "same model and step with compute_dtype float32: every product on v_mfma_f32_32x32x2_f32 (exact f32, "
                                        "157.3 TFLOP/s peak), fp32 attention; the tier north_star's 1e-3 tolerance is stated for"}
        if host_ms is not None:
            bytes_per_sample = L * C * IMG * IMG * {"f64": 8, "f32": 4, "u8": 1}[args.from_host] + L * S * 12 + 8
            res["from_host"] = {"pixels": args.from_host, "ms_per_step": round(host_ms, 3),
                                "samples_per_s": round(world * B / host_ms * 1e3, 2),
                                "h2d_GB_per_s_per_gpu": round(B * bytes_per_sample / host_ms / 1e6, 2),
                                "what": "same step, batches gathered from host numpy arrays into pinned staging, H2D on a side "
                                        "stream (double-buffered), pixel conversion + patchify on the device"}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(E)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
