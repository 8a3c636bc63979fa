"""dev probe: fixed cost (prologue + atomic epilogue) of the TN dW kernel: time it at two token counts with the same grid"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
for (N, K) in [(768, 768), (2304, 768), (768, 3072)]:
    tiles = (N // 256) * (K // 256)
    splits = 256 // tiles
    res = []
    for steps in (32, 128, 512):
        M = splits * steps * 64
        dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        def run():
            check(lib.meant_linear_bwd_dw(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, 1, None, 0, st), "dw")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append((steps, e0.elapsed_time(e1) / 20 * 1e3))
        del dy, x
    (s0, t0), (s1, t1), (s2, t2) = res
    per = (t2 - t1) / (s2 - s1)
    print(f"TN N={N} K={K} ({tiles} tiles x {splits} splits): " + "  ".join(f"{s} steps {t:.0f} us" for s, t in res) +
          f"   -> {per:.3f} us/step, fixed {t1 - per * s1:.0f} us", flush=True)
