"""round 4: what each epilogue mode of the ping-pong GEMM costs over the plain one, at the step's shapes: plain (bias), + residual,
GELU + pre-activation, rotary (fused q|k|v projection, text xPos tables / vision pixel tables), rows in TFLOP/s and ms"""
import math, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import meant_amd
from meant_amd._lib import lib, check
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
EPI_GELU, EPI_RES = 1, 2
for (tag, M, S) in [("text", 786432, 512), ("vision", 301056, 196)]:
    K = 768
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for N in (768, 2304):
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        res = {}
        res["plain"] = timed(lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st)))
        if N == 768:
            r = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
            pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            res["+ residual"] = timed(lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), r.data_ptr(), N, y.data_ptr(), N, None, M, N, K, EPI_RES, 1, st)))
            res["GELU + preact"] = timed(lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, pre.data_ptr(), M, N, K, EPI_GELU, 1, st)))
            del r, pre
        else:
            rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True) if tag == "text" else meant_amd.RotaryEmbedding(dim=32, freqs_for="pixel")
            qa, qb, ka, kb = rot.tables(S, dev)
            res["rotary (q|k|v)"] = timed(lambda: check(lib.meant_qkv_proj_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, S, 12, 64, qa.shape[1],
                                                                                 qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), 1, st)))
        print(f"{tag:6s} M={M} N={N} K={K}: " + "   ".join(f"{k} {fl/t/1e12:7.1f} TF ({t*1e3:.3f} ms)" for k, t in res.items()), flush=True)
        del w, y
    del x
