"""dev probe: flash attention forward / backward at a few (G, S, causal) points"""
import sys, os, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = "cuda"; BF16 = 1
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
st = torch.cuda.current_stream().cuda_stream
H, Dh = 12, 64; D = H * Dh
for (G, S, causal) in [(384, 512, 1), (384, 512, 0), (96, 2048, 0), (96, 2048, 1), (1536, 512, 1), (1536, 196, 0)]:
    qkv = torch.randn(G * S, 3 * D, device=dev).bfloat16()
    o = torch.empty(G * S, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(G, H, S, 2, device=dev)
    mask = torch.ones(G, S, device=dev) if causal else None
    scale = 1 / math.sqrt(D)
    wsb = lib.meant_attn_ws(G, S, H, Dh, BF16); ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    mp = mask.data_ptr() if mask is not None else None
    tf = timeit(lambda: check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st)))
    do = torch.randn_like(o); dqkv = torch.empty_like(qkv)
    tb = timeit(lambda: check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mp, dqkv.data_ptr(), G, S, H, Dh, scale, causal, None, None, None, None, 0, BF16, ws.data_ptr(), wsb, st)))
    # executed 64-key tiles per 32-query wave (causal skips tiles above the diagonal)
    nq = (S + 31) // 32
    wt = sum(min((32 * i + 31) // 64 + 1, (S + 63) // 64) if causal else (S + 63) // 64 for i in range(nq)) * G * H
    print(f"G={G} S={S} causal={causal}: fwd {tf*1e3:.3f} ms ({tf*2.0e9*1024/wt:.0f} clk@2GHz per wave-tile per SIMD)  bwd {tb*1e3:.3f} ms ({tb*2.0e9*1024/wt:.0f})", flush=True)
    del qkv, o, do, dqkv
