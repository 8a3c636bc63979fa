"""round 4: the streaming NT GEMM in its lock-step (nt_pp = 0) and ping-pong (nt_pp = 1) forms, alternating in one process, with
hipBLASLt (torch.matmul) beside them; checks the ping-pong result against an fp32 product on the first and last 2048 rows"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meant_amd._lib import lib, check
dev = torch.device("cuda")
shapes = [(786432, 768, 768), (786432, 2304, 768), (786432, 768, 2304), (786432, 768, 3072), (301056, 768, 768), (301056, 2304, 768), (8192, 8192, 8192)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2:]]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rounds = int(os.environ.get("PROBE_ROUNDS", "3"))
blas = os.environ.get("PROBE_BLAS", "1") == "1"
def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.float32)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), None, 0, y.data_ptr(), N, None, M, N, K, 0, 1, st), "lin")
    variants = [int(v) for v in os.environ.get("PROBE_PP", "0,1,4").split(",")]
    res = {v: [] for v in variants}
    for r in range(rounds):
        for pp in variants:
            check(lib.meant_set_option(b"nt_pp", pp), "opt")
            res[pp].append(timed(run))
    y.zero_()
    check(lib.meant_set_option(b"nt_pp", variants[-1]), "opt")
    run(); torch.cuda.synchronize()
    err = 0.0
    for sl in (slice(0, 2048), slice(M - 2048, M), slice(M // 2 - 1024, M // 2 + 1024)):
        ref = x[sl].float() @ w.float().t() + b
        err = max(err, ((y[sl].float() - ref).abs().max() / ref.abs().max()).item())
    y1 = y.clone()
    check(lib.meant_set_option(b"nt_pp", 0), "opt")
    run(); torch.cuda.synchronize()
    same = torch.equal(y, y1)
    fl = 2.0 * M * N * K
    line = f"M={M} N={N} K={K}:" + "".join(f"  pp={v}: " + " ".join(f"{fl/t/1e12:7.1f}" for t in res[v]) for v in variants)
    if blas:
        tb = timed(lambda: torch.matmul(x, w.t(), out=y))
        line += f"  hipBLASLt {fl/tb/1e12:7.1f}"
    print(line + f" TF   relerr {err:.2e}  bit-equal to lock-step: {same}", flush=True)
    del x, w, y, y1
