"""lab: per-workgroup cycle stamps of attn_bwd_dkv_kernel<1> (lib built with -DATTN_LAB_STAMP): prologue / tile loop / epilogue"""
import ctypes, math, os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ["MEANT_LIB_PATH"] = os.path.join(ROOT, "tools", "lab", "lib_STAMP.so")
sys.path.insert(0, ROOT)
import meant_amd
from meant_amd._lib import lib, check
dev = "cuda"; BF16 = 1; H = 12; Dh = 64; D = 768
st = torch.cuda.current_stream().cuda_stream
G, S, causal = 1536, 512, 1
rs = np.random.RandomState(0)
qkv = torch.randn(G * S, 3 * D, device=dev).bfloat16()
o = torch.empty(G * S, D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(G, H, S, 2, device=dev)
m = np.ones((G, S), dtype=np.float32)
for g, p in enumerate(rs.randint(0, 384, G)):
    if p: m[g, S - p:] = 0
mask = torch.from_numpy(m).to(dev)
rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True)
qa, qb, ka, kb = rot.tables(S, torch.device(dev))
scale = 1 / math.sqrt(D)
wsb = lib.meant_attn_ws(G, S, H, Dh, BF16); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mask.data_ptr(), G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st))
do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
for _ in range(3):
    check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mask.data_ptr(), dqkv.data_ptr(), G, S, H, Dh, scale, causal,
                             qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), 48, BF16, ws.data_ptr(), wsb, st))
torch.cuda.synchronize()
n = 4 * 12 * 1536
buf = np.zeros(n * 4, dtype=np.uint64)
f = ctypes.CDLL(os.environ["MEANT_LIB_PATH"]).meant_lab_stamps
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert f(buf.ctypes.data, buf.nbytes) == 0
a = buf.reshape(n, 4).astype(np.float64)
pro, loop, epi, tiles = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
print("s_memtime ticks (100 MHz constant clock on some parts, shader clock on others: compare shares)")
for nt in sorted(set(tiles.astype(int))):
    sel = tiles == nt
    print(f"tiles={nt}: n={sel.sum():6d}  prologue {pro[sel].mean():8.0f}  loop {loop[sel].mean():8.0f} ({loop[sel].mean()/max(nt,1):7.0f}/tile)  epilogue {epi[sel].mean():8.0f}")
print("sum over WGs: prologue %.3g loop %.3g epilogue %.3g" % (pro.sum(), loop.sum(), epi.sum()))
