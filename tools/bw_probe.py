import torch, time
x = torch.empty(1<<30, device="cuda", dtype=torch.bfloat16)  # 2 GiB
y = torch.empty_like(x)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e-3
tw = t(lambda: x.zero_()); print(f"fill 2GiB: {tw*1e3:.3f} ms  write {2.147/tw/1e3:.2f} TB/s")
tc = t(lambda: y.copy_(x)); print(f"copy 2GiB: {tc*1e3:.3f} ms  r+w {2*2.147/tc/1e3:.2f} TB/s")
tr = t(lambda: x.sum()); print(f"sum 2GiB: {tr*1e3:.3f} ms  read {2.147/tr/1e3:.2f} TB/s")
