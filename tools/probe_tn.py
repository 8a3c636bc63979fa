"""dev probe: time the TN (weight-gradient) bf16 GEMM entry point at the step's shapes and check it"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
M = 786432
st = torch.cuda.current_stream().cuda_stream
for (N, K) in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    # meant_linear_bwd_dw(dy, lddy, x, ldx, dw, dbias, M, N, K, dtype, stream)
    def run():
        check(lib.meant_linear_bwd_dw(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr() if os.environ.get("NOBIAS") is None else None, M, N, K, 1, None, 0, st), "dw")
    run(); torch.cuda.synchronize()
    ref = dy[:, :64].float().t() @ x[:, :64].float()
    err = (dw[:64, :64] - ref).abs().max().item() / ref.abs().max().item()
    refb = dy.float().sum(0)
    errb = (db - refb).abs().max().item() / refb.abs().max().item()
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"TN M={M} N={N} K={K}: {2.0*M*N*K/t/1e12:7.1f} TF ({t*1e3:.3f} ms)  relerr dW {err:.2e} db {errb:.2e}", flush=True)
    del dy, x
