"""dev probe: meant_vision (C2) step time by head count, with a torch-profiler kernel table"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import meant_amd as M
from meant_amd.train import cross_entropy_on_probs
dev = torch.device("cuda")
B = 256
img = torch.randn(B, 1, 4, 224, 224, device=dev, dtype=torch.bfloat16)
tgt = torch.randint(0, 2, (B,), device=dev)
for H in (12, 8, 12):
    m = M.meant_vision(768, 4, 224, 224, 16, 1, 2, num_heads=H, num_encoders=1, channels=4).to(dev).train()
    m.compute_dtype = torch.bfloat16
    def step():
        for p in m.parameters(): p.grad = None
        l = cross_entropy_on_probs(m(img), tgt); l.backward(); return l
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); print(f"H={H}: {(time.perf_counter()-t0)/20*1e3:.2f} ms", flush=True)
    if H == 12:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3): step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
