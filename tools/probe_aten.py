"""dev probe: which aten ops (i.e. torch's own small kernels) a headline step still runs, by count and GPU time"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from meant_amd.train import cross_entropy_on_probs
from meant_amd.parallel import GradReducer
dev = torch.device("cuda")
model = bench.build_model(1, dev); model.train()
red = GradReducer(model.parameters(), bucket_mb=64.0)
tw, im, mask, tgt = bench.make_batch(128, 0, dev)
def step():
    red.prepare(); l = cross_entropy_on_probs(model(tw, im, mask), tgt); l.backward(); red.wait()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:18]:
    print(f"{e.key:40s} calls {e.count:4d}  gpu {e.device_time_total/1e3:7.3f} ms  cpu {e.cpu_time_total/1e3:7.3f} ms")
print("all aten with GPU time:", sum(e.count for e in rows), "calls,", round(sum(e.self_device_time_total for e in rows)/1e3, 3), "ms self GPU")
