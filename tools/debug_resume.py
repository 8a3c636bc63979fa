"""dev: is a 5-step TrainStep run bit-reproducible (deterministic option), and does a resume from the state dicts after step 3 reproduce it?"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import meant_amd
from meant_amd import _lib
from meant_amd.train import TrainStep, CosineWarmRestarts
dev = torch.device("cuda:0")
_lib.set_option("deterministic", int(os.environ.get("DET", "1")))
mode = os.environ.get("MODE", "train")
def make():
    torch.manual_seed(0)
    m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2).to(dev)
    m.train(mode == "train")
    m.compute_dtype = torch.bfloat16
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    return m, ts, CosineWarmRestarts(ts.opt, T_0=7, eta_min=1e-5)
g = torch.Generator().manual_seed(5)
ids = torch.randint(0, 100, (8, 3, 16), generator=g).to(dev)
img = torch.randn(8, 3, 4, 32, 32, generator=g).to(dev)
mask = torch.ones(8, 3, 16, device=dev)
tgt = torch.tensor([0, 1, 0, 1, 1, 0, 1, 0], device=dev)
def steps(ts, sched, first, n):
    for i in range(first, first + n):
        torch.manual_seed(100 + i)
        ts(ids, img, mask, target=tgt)
        sched.step()
def full():
    m, ts, sched = make()
    for _ in range(5): sched.step()
    steps(ts, sched, 0, 3)
    saved = {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": ts.opt.state_dict(), "sched": sched.state_dict()}
    steps(ts, sched, 3, 2)
    return {k: v.clone() for k, v in m.state_dict().items()}, saved
a, saved = full()
b, _ = full()
def diff(x, y, tag):
    worst = max(((x[k].float() - y[k].float()).abs().max().item(), k) for k in x)
    nd = sum(1 for k in x if not torch.equal(x[k], y[k]))
    print(f"{tag}: {nd} of {len(x)} tensors differ, worst abs diff {worst[0]:.3e} at {worst[1]}")
diff(a, b, "two fresh runs")
m2, ts2, sched2 = make()
with torch.no_grad():
    for p in m2.parameters(): p.add_(0.123)
m2.load_state_dict(saved["model"])
ts2.opt.load_state_dict(saved["opt"])
sched2.load_state_dict(saved["sched"])
steps(ts2, sched2, 3, 2)
diff(a, {k: v.clone() for k, v in m2.state_dict().items()}, "resumed vs original")
# without the perturbation
m3, ts3, sched3 = make()
m3.load_state_dict(saved["model"]); ts3.opt.load_state_dict(saved["opt"]); sched3.load_state_dict(saved["sched"])
steps(ts3, sched3, 3, 2)
diff(a, {k: v.clone() for k, v in m3.state_dict().items()}, "resumed (no perturbation) vs original")
