#!/usr/bin/env python3
"""dev probe: relative gradient-norm errors of the full-dims golden (tests/test_gpu_models.py::test_meant_full_c3_golden, bf16
tier), ten worst parameters, for the build named by MEANT_LIB_PATH"""
import os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
import numpy as np, torch
from tests.util import t, norm_floor
from tests.test_gpu_models import _mk
dev = torch.device("cuda:0")
g = np.load(os.path.join(R, "tests", "golden", "meant_full_c3.npz"), allow_pickle=False)
r = np.random.RandomState(99)
ids = t(r.randint(0, 2000, (2, 12, 512)).astype("int64")); img = t(r.standard_normal((2, 12, 4, 224, 224)).astype("float32"))
mask = torch.ones(2, 12, 512); mask[1, :, 400:] = 0
_, hip = _mk("meant", (768, 768, 4, 224, 224, 16, 12, 2), dict(num_heads=12, num_encoders=1), (2000, 768), dev)
hip.compute_dtype = torch.bfloat16
out = hip(ids.to(dev), img.to(dev), mask.to(dev))
if os.environ.get("UPSTREAM") == "golden":
    go = t(g["out"]).to(dev).requires_grad_()
    torch.nn.functional.cross_entropy(go, t(g["target"]).to(dev)).backward()
    print("upstream grad from golden out", go.grad.flatten().tolist(), "out", out.detach().flatten().tolist(), "golden", go.detach().flatten().tolist())
    out.backward(go.grad)
else:
    loss = torch.nn.functional.cross_entropy(out, t(g["target"]).to(dev)); loss.backward()
params = dict(hip.named_parameters()); floor = norm_floor(g["grad_norms"], torch.bfloat16)
res = []
for name, ref in zip(g["grad_names"], g["grad_norms"]):
    a = params[str(name)].grad.double().norm().item()
    res.append((abs(a - ref) / max(ref, floor), str(name), a, float(ref)))
res.sort(reverse=True)
print(os.path.basename(os.environ.get("MEANT_LIB_PATH", "tree")), "out err", (out.detach().cpu() - t(g["out"])).abs().max().item())
for x in res[:8]: print("  %.4f %s %.5f %.5f" % x)
