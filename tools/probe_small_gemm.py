"""dev probe: the small products of the step's serial stretch (temporal encoder, pooled tails): bf16 NT at 1536-row shapes, the fp32
products of the pooled Linear, one line per build (MEANT_LIB_PATH)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = [os.path.basename(os.environ.get("MEANT_LIB_PATH", "tree"))]
for dt, shapes in ((1, [(1536, 1536, 1536), (1536, 4608, 1536), (1536, 768, 768), (128, 1536, 1536), (1664, 1536, 1536)]), (0, [(1536, 768, 768), (1536, 1536, 1536)])):
    for (M, N, K) in shapes:
        T = torch.bfloat16 if dt else torch.float32
        x = torch.randn(M, K, device=dev, dtype=T); w = torch.randn(N, K, device=dev, dtype=T); y = torch.empty(M, N, device=dev, dtype=T)
        def run():
            check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, 0, dt, st), "lin")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 50 * 1e3
        ref = x.float() @ w.float().t()
        err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
        out.append(f"{'bf16' if dt else 'f32'} {M}x{N}x{K}: {t:.1f} us (err {err:.1e})")
print("  ".join(out), flush=True)
