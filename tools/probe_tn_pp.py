"""round 4: the 256 x 256 dW kernel in its lock-step (tn_pp = 0) and ping-pong (tn_pp = 1) forms, alternating in one process, at the
step's shapes; checks dW (full matrix) and dbias of the ping-pong form against fp32 products of a row sample"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
shapes = [(786432, 768, 768), (786432, 2304, 768), (786432, 768, 2304), (301056, 768, 768), (301056, 2304, 768), (786432, 768, 1024)]
rounds = 3
for (M, N, K) in shapes:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    def run():
        check(lib.meant_linear_bwd_dw(dy.data_ptr(), N, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, 1, None, 0, st), "dw")
    def timed(n=10):
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    PPS = [int(v) for v in os.environ.get("PROBE_TN", "0,1").split(",")]
    res = {v: [] for v in PPS}
    for r in range(rounds):
        for pp in PPS:
            check(lib.meant_set_option(b"tn_pp", pp), "opt")
            res[pp].append(timed())
    outs = {}
    for pp in PPS:
        check(lib.meant_set_option(b"tn_pp", pp), "opt")
        dw.zero_(); db.zero_()
        run(); torch.cuda.synchronize()
        outs[pp] = (dw.clone(), db.clone())
    # reference in chunks (fp32 accumulate of bf16 products is what the kernels do up to summation order)
    ref = torch.zeros(N, K, device=dev)
    for m0 in range(0, M, 65536):
        ref += dy[m0:m0 + 65536].float().t() @ x[m0:m0 + 65536].float()
    refb = dy.float().sum(0)
    e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"TN M={M} N={N} K={K}:" + "".join(f"  tn_pp={v}: " + " ".join(f"{fl/t/1e12:7.1f}" for t in res[v]) for v in PPS) + " TF   relerr dW " +
          " ".join(f"{e(outs[v][0], ref):.2e}" for v in PPS) + "  db " + " ".join(f"{e(outs[v][1], refb):.2e}" for v in PPS), flush=True)
    del dy, x, ref
