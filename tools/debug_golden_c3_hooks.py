#!/usr/bin/env python3
"""dev probe: module outputs of the full-dims golden model (bf16 tier) saved per build; with two files: first differences"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import numpy as np, torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        x, y = a[k].float(), b[k].float()
        print(f"{k:60s} rel {((x - y).norm() / max(y.norm().item(), 1e-30)).item():.3e} |b| {y.norm().item():.4e} shape {tuple(y.shape)}")
    sys.exit(0)
from tests.util import t
from tests.test_gpu_models import _mk
dev = torch.device("cuda:0")
g = np.load(os.path.join(R, "tests", "golden", "meant_full_c3.npz"), allow_pickle=False)
r = np.random.RandomState(99)
ids = t(r.randint(0, 2000, (2, 12, 512)).astype("int64")); img = t(r.standard_normal((2, 12, 4, 224, 224)).astype("float32"))
mask = torch.ones(2, 12, 512); mask[1, :, 400:] = 0
_, hip = _mk("meant", (768, 768, 4, 224, 224, 16, 12, 2), dict(num_heads=12, num_encoders=1), (2000, 768), dev)
hip.compute_dtype = torch.float32 if os.environ.get("DT") == "f32" else torch.bfloat16
out = {}
def flat(o, pre):
    if torch.is_tensor(o): yield pre, o
    elif isinstance(o, (tuple, list)):
        for i, z in enumerate(o): yield from flat(z, f"{pre}[{i}]")
def hook(name):
    def f(m, inp, o):
        for k, z in flat(o, name + " out"):
            if z is not None and z.is_floating_point(): out[k] = z.detach().float().cpu()
    return f
def bhook(name):
    def f(m, gi, go):
        for k, z in flat(go, name + " grad_out"):
            if z is not None and z.is_floating_point(): out[k] = z.detach().float().cpu()
        for k, z in flat(gi, name + " grad_in"):
            if z is not None and z.is_floating_point(): out[k] = z.detach().float().cpu()
    return f
for n, m in hip.named_modules():
    if n and n.count(".") <= 3:
        m.register_forward_hook(hook(n))
        if os.environ.get("BHOOK"): m.register_full_backward_hook(bhook(n))
o = hip(ids.to(dev), img.to(dev), mask.to(dev)); out["final"] = o.detach().cpu()
go = t(g["out"]).to(dev).requires_grad_()
torch.nn.functional.cross_entropy(go, t(g["target"]).to(dev)).backward()
o.backward(go.grad)
for n, p in hip.named_parameters():
    if p.grad is not None and ("bias" in n or "scale" in n): out["grad " + n] = p.grad.detach().float().cpu()
torch.save(out, sys.argv[1])
