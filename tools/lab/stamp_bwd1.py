"""lab: per-workgroup s_memtime sums of attn_bwd1_kernel (lib built by `bash tools/lab/build_lab.sh STAMP1 -DATTN_LAB_STAMP`):
prologue / phase 1 / barrier wait / phase 2 / barrier wait / dQ epilogue / dK dV epilogue, text and vision shapes of the step"""
import ctypes, math, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ["MEANT_LIB_PATH"] = os.path.join(ROOT, "tools", "lab", "lib_STAMP1.so")
sys.path.insert(0, ROOT)
import meant_amd
from meant_amd._lib import lib, check
dev = "cuda"; BF16 = 1; H = 12; Dh = 64; D = 768
st = torch.cuda.current_stream().cuda_stream
rs = np.random.RandomState(0)
for (G, S, causal) in [(1536, 512, 1), (1536, 196, 0)]:
    qkv = torch.randn(G * S, 3 * D, device=dev).bfloat16()
    o = torch.empty(G * S, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(G, H, S, 2, device=dev)
    mp = None
    if causal:
        m = np.ones((G, S), dtype=np.float32)
        for g, p in enumerate(rs.randint(0, 384, G)):
            if p: m[g, S - p:] = 0
        mask = torch.from_numpy(m).to(dev); mp = mask.data_ptr()
        rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True)
    else:
        rot = meant_amd.RotaryEmbedding(dim=32, freqs_for="pixel")
    qa, qb, ka, kb = rot.tables(S, torch.device(dev))
    scale = 1 / math.sqrt(D)
    wsb = lib.meant_attn_ws(G, S, H, Dh, BF16); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st))
    do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    for _ in range(3):
        check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mp, dqkv.data_ptr(), G, S, H, Dh, scale, causal,
                                 qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), qa.shape[1], BF16, ws.data_ptr(), wsb, st))
    torch.cuda.synchronize()
    n = min(G * H, 20000)
    buf = np.zeros(n * 16, dtype=np.uint64)
    f = ctypes.CDLL(os.environ["MEANT_LIB_PATH"]).meant_lab_stamps1
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    a = buf.reshape(n, 16).astype(np.float64)
    names = ["prologue", "phase1", "wait Y", "phase2", "wait X'", "dQ epi", "dKdV epi"]
    print(f"S={S} causal={causal}: s_memtime ticks per workgroup (wave 0), mean over {n} workgroups; chunks/wg {a[:, 7].mean():.2f}")
    tot = a[:, :7].sum(axis=1).mean()
    for i, nm in enumerate(names):
        print(f"   {nm:10s} {a[:, i].mean():9.0f}  ({a[:, i].mean() / tot:5.1%})   per chunk {a[:, i].sum() / max(a[:, 7].sum(), 1):8.0f}")
    print(f"   total      {tot:9.0f}   prologue split: set-up + requests {a[:, 8].mean():.0f}, wait for memory {a[:, 9].mean():.0f}, statistics {a[:, 10].mean():.0f}, barrier {a[:, 0].mean():.0f}")
    tot += a[:, 8:16].sum(axis=1).mean()
    nch = max(a[:, 7].sum(), 1)
    print("   per chunk: wait for tile b %.0f, its statistics %.0f, barrier M %.0f, barrier Y %.0f, wait for next tile a %.0f, its statistics + request %.0f, barrier X' %.0f   (all buckets together %.0f per workgroup)"
          % (a[:, 12].sum() / nch, a[:, 13].sum() / nch, a[:, 2].sum() / nch, a[:, 11].sum() / nch, a[:, 14].sum() / nch, a[:, 15].sum() / nch, a[:, 4].sum() / nch, tot))
    for nc in sorted(set(a[:, 7].astype(int))):
        sel = a[:, 7] == nc
        print(f"   chunks={nc}: n={int(sel.sum()):6d} " + " ".join(f"{a[sel, i].mean():8.0f}" for i in range(7)))
    del qkv, o, do, dqkv
