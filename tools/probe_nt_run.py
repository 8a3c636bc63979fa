"""lab: the streaming NT GEMM at the step's big shapes under MEANT_NT_DYNAMIC = 1 (per-XCD counters, the default), 0 (fixed walk) and
4 (fixed walk in runs of one A row panel: the nine column tiles of a panel by ONE workgroup, back to back) -- time per launch.
usage: MEANT_NT_DYNAMIC=4 python3 tools/probe_nt_run.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd._lib import lib, check, BF16, EPI_NONE
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(786432, 2304, 768), (786432, 768, 768), (786432, 768, 2304), (301056, 2304, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: check(lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, None, 0, y.data_ptr(), N, None, M, N, K, EPI_NONE, BF16, st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = (x[:512].float() @ w.float().t())
    err = (y[:512].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"mode {os.environ.get('MEANT_NT_DYNAMIC', '1')}  M={M} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s  rel err {err:.1e}", flush=True)
    del x, w, y
