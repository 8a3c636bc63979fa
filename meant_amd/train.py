"""The train step around the hot path (reference: in_loop_train.py:202-239, 547-567).

Reference recipe: fp16 autocast + GradScaler, CrossEntropyLoss on the Sigmoid-ed output, clip_grad_norm_(1.0),
AdamW(lr 5e-5), CosineAnnealingWarmRestarts(T_0=7) stepped per epoch.  Here: bf16 compute needs no loss scaling,
and the tail of the step is three kinds of HIP launches over the flat fp32 buckets of the gradient reducer
(meant_amd/parallel.py) -- one sum-of-squares pass, one fused clip+AdamW pass per bucket -- instead of PyTorch's
per-tensor foreach chains.  Nothing in the step synchronises with the host.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from . import ops
from ._lib import lib, check
from .parallel import GradReducer


# ---------------------------------------------------------------------------------------------
class _CEOnProbs(torch.autograd.Function):
    """nn.CrossEntropyLoss()(probabilities, target), as in_loop_train.py:232 applies it to the Sigmoid output."""

    @staticmethod
    def forward(ctx, probs, target):
        ops._need_gpu(probs, target)
        p = ops._c(probs.float())
        t = ops._c(target.long())
        B, C = p.shape
        loss = torch.zeros(1, device=p.device, dtype=torch.float32)
        dp = torch.empty_like(p)
        check(lib.meant_ce_probs(ops._p(p), ops._p(t), ops._p(loss), ops._p(dp), B, C, ops._stream()), "ce_probs")
        ctx.save_for_backward(dp)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        (dp,) = ctx.saved_tensors
        return dp * dloss, None


def cross_entropy_on_probs(probs: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return _CEOnProbs.apply(probs, target)


# ---------------------------------------------------------------------------------------------
def flatten_parameters(reducer: GradReducer):
    """Make every parameter a view into a flat fp32 buffer laid out like its gradient bucket, so that one
    fused kernel can update a whole bucket.  Returns the list of flat parameter buffers (one per bucket)."""
    flats = []
    for b in reducer.buckets:
        flat = torch.zeros_like(b.flat)
        with torch.no_grad():
            for p, off in zip(b.params, b.offsets):
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view_as(p)
        flats.append(flat)
    return flats


class FusedAdamW:
    """torch.optim.AdamW + torch.nn.utils.clip_grad_norm_ semantics on the reducer's flat buckets.

        red = GradReducer(model.parameters()); opt = FusedAdamW(red, lr=5e-5, max_grad_norm=1.0)
        red.prepare(); loss = f(model(x)); loss.backward(); red.wait(); opt.step()

    (`prepare()` comes BEFORE the forward pass: it also clears the per-step gradient-sink claims the forward's ops make.)
    `state_dict()` / `load_state_dict()` carry the step count, the learning rate and both moment buffers: a fresh optimizer over a
    model with the same parameter registration order continues exactly (checkpoint_train.py:217-219,333-336 of the reference
    saves and restores `optimizer.state_dict()` the same way)."""

    def __init__(self, reducer: GradReducer, lr: float = 5e-5, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: Optional[float] = None):
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.flat_params = flatten_parameters(reducer)
        self.exp_avg = [torch.zeros_like(f) for f in self.flat_params]
        self.exp_avg_sq = [torch.zeros_like(f) for f in self.flat_params]
        self.step_count = 0
        dev = self.flat_params[0].device if self.flat_params else torch.device("cuda")
        self._sumsq = torch.zeros(1, device=dev, dtype=torch.float32)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        self.step_count += 1
        st = ops._stream()
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if clip:
            self._sumsq.zero_()
            for b in self.reducer.buckets:
                check(lib.meant_sumsq_f32(ops._p(b.flat), b.flat.numel(), ops._p(self._sumsq), st), "sumsq_f32")
        for b, p, m, v in zip(self.reducer.buckets, self.flat_params, self.exp_avg, self.exp_avg_sq):
            check(lib.meant_adamw_f32(ops._p(p), ops._p(b.flat), ops._p(m), ops._p(v), p.numel(), self.lr, self.betas[0],
                                      self.betas[1], self.eps, self.weight_decay, self.step_count,
                                      ops._p(self._sumsq) if clip else None, float(self.max_grad_norm or 0.0), float(grad_scale), st),
                  "adamw_f32")
        ops.weights.invalidate()        # parameters changed through raw pointers: drop the cached bf16 / transposed copies

    def grad_norm(self) -> torch.Tensor:
        """global L2 norm of the (already reduced) gradients, as a device scalar"""
        self._sumsq.zero_()
        for b in self.reducer.buckets:
            check(lib.meant_sumsq_f32(ops._p(b.flat), b.flat.numel(), ops._p(self._sumsq), ops._stream()), "sumsq_f32")
        return self._sumsq.sqrt()

    def state_dict(self):
        return {"step": self.step_count, "lr": self.lr, "exp_avg": [t.clone() for t in self.exp_avg],
                "exp_avg_sq": [t.clone() for t in self.exp_avg_sq]}

    def load_state_dict(self, sd):
        self.step_count, self.lr = sd["step"], sd["lr"]
        for d, s in zip(self.exp_avg, sd["exp_avg"]):
            d.copy_(s)
        for d, s in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            d.copy_(s)


class CosineWarmRestarts:
    """torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(T_0, T_mult=1, eta_min) stepped once per epoch
    (in_loop_train.py:567, :280): lr(e) = eta_min + (base - eta_min) * (1 + cos(pi * (e mod T_0) / T_0)) / 2."""

    def __init__(self, optimizer: FusedAdamW, T_0: int = 7, eta_min: float = 0.0):
        self.opt, self.T_0, self.eta_min = optimizer, T_0, eta_min
        self.base_lr = optimizer.lr
        self.epoch = 0

    def lr_at(self, epoch: int) -> float:
        t = epoch % self.T_0
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * t / self.T_0)) / 2

    def step(self):
        self.epoch += 1
        self.opt.lr = self.lr_at(self.epoch)

    def state_dict(self):
        return {"epoch": self.epoch, "base_lr": self.base_lr, "T_0": self.T_0, "eta_min": self.eta_min}

    def load_state_dict(self, sd):
        self.epoch, self.base_lr, self.T_0, self.eta_min = sd["epoch"], sd["base_lr"], sd["T_0"], sd["eta_min"]
        self.opt.lr = self.lr_at(self.epoch)


class TrainStep:
    """forward -> CE on probabilities -> backward (+ overlapped gradient all-reduce) -> clip + AdamW."""

    def __init__(self, model: torch.nn.Module, lr: float = 5e-5, weight_decay: float = 1e-2, max_grad_norm: float = 1.0,
                 bucket_mb: float = 64.0, direct_grads: bool = True, micro_batches: int = 1):
        """micro_batches: run a step's batch as this many equal slices, one forward/backward each, gradients accumulated in the
        reducer's buckets (reducer.no_sync) -- the same gradients with 1/micro_batches of the activations in flight (the
        reference's CLI default of 12 encoder layers at 128 samples per GPU).
        direct_grads (GradReducer): parameter gradients are accumulated straight into the flat buckets by the backward
        kernels; pass False if the model's parameters are also differentiated outside this step (torch.autograd.grad,
        several backward passes per optimizer step without reducer.no_sync())"""
        self.model = model
        self.reducer = GradReducer(model.parameters(), bucket_mb=bucket_mb, direct_grads=direct_grads)
        self.opt = FusedAdamW(self.reducer, lr=lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        self.micro_batches = max(1, int(micro_batches))

    def __call__(self, *inputs, target):
        self.reducer.prepare()
        k = self.micro_batches
        if k == 1:
            out = self.model(*inputs)
            loss = cross_entropy_on_probs(out, target)
            loss.backward()
        else:
            B = target.shape[0]
            assert B % k == 0, "micro_batches must divide the batch"
            mb, outs, loss = B // k, [], 0.0
            for i in range(k):
                sl = slice(i * mb, (i + 1) * mb)
                part = tuple(t[sl] if torch.is_tensor(t) else t for t in inputs)
                if i < k - 1:
                    with self.reducer.no_sync():
                        o = self.model(*part)
                        li = cross_entropy_on_probs(o, target[sl]) * (1.0 / k)
                        li.backward()
                else:
                    o = self.model(*part)
                    li = cross_entropy_on_probs(o, target[sl]) * (1.0 / k)
                    li.backward()
                outs.append(o.detach())
                loss = loss + li.detach()
            out = torch.cat(outs)
        self.reducer.wait()
        self.opt.step()
        return loss.detach(), out.detach()
