#!/usr/bin/env python3
"""Launch the HBM-bound kernels of the bench step (RMSNorm forward / backward in their packed, pooled and chained forms, the
three attention kernels) at the step's two shapes -- text: 128 x 12 sequences of 512 tokens, vision: 128 x 12 of 196 -- three
times each, so that rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md) can put measured HBM
bytes next to the algorithmic ones.  tools/parse_pmc_kernels.py turns the two CSVs into profiles/*.json.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_step_kernels.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meant_amd import ops

dev = "cuda"
d, H = 768, 12
REP = 3
for name, G, S, causal in (("text", 1536, 512, True), ("vision", 1536, 196, False)):
    T = G * S
    x = torch.randn(G, S, d, device=dev).bfloat16()
    g = torch.ones(d, device=dev, requires_grad=True)
    W = (torch.randn(d, d, device=dev) / d ** 0.5).requires_grad_()
    b = torch.zeros(d, device=dev, requires_grad=True)
    for _ in range(REP):                                   # packed forward + backward with the residual gradient folded in
        xr = x.clone().requires_grad_()
        y, res = ops.rmsnorm_fork(xr, g)
        torch.autograd.backward([y, res], [torch.ones_like(y), torch.ones_like(res)])
    for _ in range(REP):                                   # packed forward + backward with dropout (language encode[3])
        xr = x.clone().requires_grad_()
        y = ops.rmsnorm(xr, g, 1e-8, 0.5, 1234)
        y.backward(torch.ones_like(y))
    for fused in (False, True):                            # the pooled tail: separate kernels, then the folded form
        ops.FUSE_NORM_LINEAR = fused
        for _ in range(REP):
            xr = x.clone().requires_grad_()
            if fused:
                hm, xm = ops.norm_linear_gelu_norm_pooled(xr, g, 1e-8, W, b, g, 1e-8, 0.5, 99)
            else:
                n, xm = ops.rmsnorm_fork_pooled(xr, g, 1e-8)
                hm = ops.linear_gelu_rmsnorm_pooled(n, W, b, g, 1e-8, 0.5, 99)
            torch.autograd.backward([hm, xm], [torch.ones_like(hm), torch.ones_like(xm)])
        for _ in range(REP):                               # the same chain inside the stack (token-level)
            xr = x.clone().requires_grad_()
            if fused:
                h, res = ops.norm_linear_gelu_norm(xr, g, 1e-8, W, b, g, 1e-8, 0.5, 99)
            else:
                n, res = ops.rmsnorm_fork(xr, g, 1e-8)
                h = ops.linear_gelu_rmsnorm(n, W, b, g, 1e-8, 0.5, 99)
            torch.autograd.backward([h, res], [torch.ones_like(h), torch.ones_like(res)])
    del x
    qkv = (torch.randn(T, 3 * d, device=dev) * 0.5).bfloat16()
    mask = torch.ones(G, S, device=dev)
    if causal:
        pad = torch.randint(0, 384, (G,))
        for i in range(G):
            if pad[i]:
                mask[i, S - pad[i]:] = 0
    for _ in range(REP):
        q = qkv.clone().requires_grad_()
        o = ops.attention_core(q, G, S, H, 1.0 / d ** 0.5, None, causal, mask if causal else None)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
    del qkv
print("done")
