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
    # persistent workgroups: every workgroup holds the sums over all the items it drew; report per chunk (128 queries x one key half)
    a = a[a[:, 7] > 0]
    nch = a[:, 7].sum()
    names = {0: "stage start: barrier", 8: "stage start: set-up + requests", 9: "stage start: wait for memory", 10: "stage start: statistics",
             1: "phase 1 (two tiles)", 12: "wait for tile b", 13: "statistics of tile b", 2: "barrier M", 11: "barrier Y", 3: "phase 2",
             14: "wait for the next tile a", 15: "its statistics + requests", 4: "barrier X'", 5: "dQ epilogue", 6: "dK dV epilogue (per stage)"}
    tot = a[:, [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15]].sum()
    print(f"S={S} causal={causal}: {len(a)} workgroups, {int(nch)} chunks; s_memtime ticks of wave 0 per chunk, share of the total")
    for i, nm in names.items():
        print(f"   {nm:34s} {a[:, i].sum() / nch:8.0f}  {a[:, i].sum() / tot:6.1%}")
    print(f"   {'total':34s} {tot / nch:8.0f}")
    del qkv, o, do, dqkv
