"""dev probe: the step's two attention shapes (text: G=1536 S=512 causal, trailing padding uniform in [0, 384) as in bench.py;
vision: G=1536 S=196, no mask) forward + backward with rotary tables, N iterations.  Run under
`rocprofv3 --kernel-trace --stats` for per-kernel times; prints event-timed totals itself.
usage: python3 tools/probe_attn2.py [iters] [heads]"""
import math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import meant_amd
from meant_amd._lib import lib, check
dev = "cuda"; BF16 = 1
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H = int(sys.argv[2]) if len(sys.argv) > 2 else 12
Dh = 768 // H
D = H * Dh
st = torch.cuda.current_stream().cuda_stream
def timeit(f, n):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
rs = np.random.RandomState(0)
for (G, S, causal) in [(1536, 512, 1), (1536, 196, 0)]:
    qkv = torch.randn(G * S, 3 * D, device=dev).bfloat16()
    o = torch.empty(G * S, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(G, H, S, 2, device=dev)
    mask = None
    if causal:
        m = np.ones((G, S), dtype=np.float32)
        for g, p in enumerate(rs.randint(0, 384, G)):
            if p: m[g, S - p:] = 0
        mask = torch.from_numpy(m).to(dev)
        rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True)
    else:
        rot = meant_amd.RotaryEmbedding(dim=math.floor(Dh / 2), freqs_for="pixel")
    qa, qb, ka, kb = rot.tables(S, torch.device(dev))
    R = qa.shape[1]
    scale = 1 / math.sqrt(D)
    wsb = lib.meant_attn_ws(G, S, H, Dh, BF16); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    mp = mask.data_ptr() if mask is not None else None
    tf = timeit(lambda: check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st)), iters)
    do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    tb = timeit(lambda: check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mp, dqkv.data_ptr(), G, S, H, Dh, scale, causal,
                                                 qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), R, BF16, ws.data_ptr(), wsb, st)), iters)
    fl = 4.0 * G * H * S * S * Dh
    # single-pass backward (attn_bwd1) against the two-pass form: time and result
    if Dh == 64:
        bwd = lambda: check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mp, dqkv.data_ptr(), G, S, H, Dh, scale, causal,
                                               qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), R, BF16, ws.data_ptr(), wsb, st))
        lib.meant_set_option(b"attn_bwd1", 0)
        t2 = timeit(bwd, iters); ref = dqkv.clone().float()
        lib.meant_set_option(b"attn_bwd1", 1)
        dqkv.fill_(float("nan")); t1 = timeit(bwd, iters); torch.cuda.synchronize()
        got = dqkv.float()
        for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
            a_, r_ = got[:, sl], ref[:, sl]
            print(f"   {name}: max|two-pass| {r_.abs().max().item():.4g}  max|diff| {(a_ - r_).abs().max().item():.4g}  rel-norm {((a_ - r_).norm() / r_.norm()).item():.3g}  nan {int(torch.isnan(a_).sum())}", flush=True)
        if os.environ.get("PROBE_REPEAT"):                     # run-to-run determinism of the single-pass kernel (no atomics: must be bit-identical)
            first = dqkv.clone()
            for rep in range(int(os.environ["PROBE_REPEAT"])):
                dqkv.fill_(float("nan")); bwd(); torch.cuda.synchronize()
                ne = (dqkv != first) | torch.isnan(dqkv)
                if ne.any():
                    idx = ne.nonzero()
                    if os.environ.get("PROBE_DUMP") and not os.path.exists(os.environ["PROBE_DUMP"]):
                        r0 = int(idx[0, 0]); c0 = (int(idx[0, 1]) // Dh) * Dh
                        r0 = (r0 // 32) * 32
                        np.savez(os.environ["PROBE_DUMP"], right=first[r0:r0 + 32, c0:c0 + Dh].float().cpu().numpy(), wrong=dqkv[r0:r0 + 32, c0:c0 + Dh].float().cpu().numpy(),
                                 r0=r0, c0=c0, S=S, qa=qa.cpu().numpy(), qb=qb.cpu().numpy(), scale=scale,
                                 qkv=qkv[(r0 // S) * S:(r0 // S + 1) * S].float().cpu().numpy(), do=do[(r0 // S) * S:(r0 // S + 1) * S].float().cpu().numpy(),
                                 o=o[(r0 // S) * S:(r0 // S + 1) * S].float().cpu().numpy(), lse=lse[r0 // S].cpu().numpy(),
                                 mask=(mask[r0 // S].cpu().numpy() if mask is not None else np.ones(S, dtype=np.float32)), H=H, Dh=Dh)
                    rows = idx[:, 0]; cols = idx[:, 1]
                    print(f"      rep {rep}: {int(ne.sum())} elements differ; rows (g, s) {sorted(set((int(r) // S, int(r) % S) for r in rows[:2000].tolist()))[:12]} cols {sorted(set((int(c) // D, (int(c) % D) // Dh) for c in cols[:2000].tolist()))[:12]}", flush=True)
                else:
                    print(f"      rep {rep}: identical", flush=True)
        if os.environ.get("PROBE_DETAIL"):
            e = (got[:, :D] - ref[:, :D]).view(G, S, H, Dh)
            r = ref[:, :D].view(G, S, H, Dh)
            for c in range((S + 127) // 128):
                sl = slice(c * 128, min(S, c * 128 + 128))
                print(f"      dq rows {c * 128:3d}..: rel-norm {(e[:, sl].norm() / r[:, sl].norm()).item():.3g}  max {e[:, sl].abs().max().item():.4g}", flush=True)
            bad = (e.abs() > 8 * e.abs().mean()).nonzero()
            print("      outliers:", bad.shape[0], bad[:12].tolist(), flush=True)
        print(f"   two-pass {t2:.3f} ms   single-pass {t1:.3f} ms", flush=True)
        del ref, got
    print(f"G={G} S={S} H={H} causal={causal}: fwd {tf:.3f} ms  bwd {tb:.3f} ms   (full-square: fwd {fl/tf/1e9:.0f} TFLOP/s, bwd {2.5*fl/tb/1e9:.0f} TFLOP/s)", flush=True)
    del qkv, o, do, dqkv
