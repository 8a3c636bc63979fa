"""Data parallelism for the MEANT path: one process per GPU, batch sharded, weights replicated,
one bucketed gradient all-reduce per step over RCCL (torch.distributed backend "nccl" on ROCm)
riding the xGMI mesh, overlapped with the rest of backward.

What it replaces: the reference's only multi-GPU mechanism is single-process nn.DataParallel
(pretrain_mlm.py:329-330, pretrain_mim.py:347-348), which re-broadcasts every weight and gathers
every output through GPU 0 each step.  Samples are independent through the whole forward/backward
(SURVEY.md 8e), so the only exchange the path needs is the gradient sum.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large buckets.
Parameters are packed, in reverse registration order (roughly the order their gradients become
final), into flat fp32 buckets of `bucket_mb`; every p.grad is a VIEW into its bucket, so there is
no flatten/unflatten copy.  A bucket's all-reduce is launched from the autograd hook of its last
parameter to finish, asynchronously, while backward continues; `wait()` joins before the optimizer.
The big embedding table (49 M of the 73.6 M parameters at V=64001) is its own bucket and, being the
first layer, is reduced last.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


ALIGN = 64      # every parameter starts on a 256-byte boundary inside its bucket (the kernels want 16-byte alignment)


def _padded(n: int) -> int:
    return (n + ALIGN - 1) // ALIGN * ALIGN


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "handle", "streams", "slice_handles")

    def __init__(self, flat, params, offsets):
        self.flat, self.params, self.offsets, self.pending, self.handle = flat, params, offsets, 0, None
        self.streams = []
        self.slice_handles = []


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, process_group=None,
                 average: bool = True, direct_grads: bool = False, row_slices: int = 4):
        """direct_grads: let the Linear / embedding backward write dW / db straight into the bucket views
        (meant_amd.ops.grad_sinks) instead of returning fresh tensors for autograd to add to them.  A parameter that two
        different kinds of op read in one step (a word embedding tied to the vocabulary decoder) is detected in forward
        and keeps autograd's accumulation; two Linears sharing one weight report twice and raise.  With direct_grads a
        backward pass ACCUMULATES into the buckets and counts every parameter once per `prepare()`: for gradient
        accumulation over micro-batches run all but the last backward under `no_sync()`.  `torch.autograd.grad` on a
        parameter with a sink returns None (its gradient went into the bucket).  A parameter belongs to one reducer at a
        time: constructing a second reducer over it closes the first."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        # a one-rank group normally skips the collective; MEANT_REDUCE_ALWAYS=1 issues it anyway (used on a one-GPU box to
        # run the real RCCL call sequence: hooks, stream ordering, async handles)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("MEANT_REDUCE_ALWAYS") == "1")
        # RCCL can average inside the collective (ncclAvg); gloo cannot, there the 1/world scaling is a separate pass
        self.fused_avg = average and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        plist: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in plist:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        uniq.reverse()
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        cur_n = 0

        def flush():
            nonlocal cur, cur_n
            if not cur:
                return
            flat = torch.zeros(cur_n, device=cur[0].device, dtype=torch.float32)
            off, offsets = 0, []
            for p in cur:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                offsets.append(off)
                off += _padded(n)
            self.buckets.append(_Bucket(flat, cur, offsets))
            cur, cur_n = [], 0

        for p in uniq:
            assert p.dtype == torch.float32, "master weights / gradients are fp32"
            if cur and (cur_n + _padded(p.numel()) > cap or p.device != cur[0].device):
                flush()
            cur.append(p)
            cur_n += _padded(p.numel())
        flush()
        self._owner = {}
        self._launch_streams = {}
        self._hooks = []
        self._sync = True
        self._closed = False
        import weakref
        wself = weakref.ref(self)            # the parameters must not keep the reducer (and its buckets) alive through their hooks

        def hook(p, _w=wself):
            r = _w()
            if r is not None:
                r._hook(p)
        for b in self.buckets:
            for p in b.params:
                self._owner[id(p)] = b
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))
        self.direct_grads = direct_grads
        # a parameter that fills a bucket on its own and is larger than the bucket size (the 64001 x 768 embedding table) may be
        # delivered by its backward in this many row slices, each reduced as soon as it is final (direct_grads only)
        self.row_slices = max(1, int(row_slices))
        self.row_slice_min_bytes = 32 * 1024 * 1024
        self._reported = set()
        self._sink_ids = []
        self.prepare()
        if direct_grads:
            self._register_sinks()

    def _register_sinks(self):
        import weakref
        from . import ops
        for b in self.buckets:
            for p, off in zip(b.params, b.offsets):
                if p.is_cuda:
                    old = ops.grad_sinks.get(id(p))
                    if old is not None and old.param() is p:
                        prev = old.reducer()
                        if prev is not None and prev is not self:
                            prev.close()                 # its hooks and sinks would otherwise keep firing on our parameters
                    view = b.flat[off:off + p.numel()].view_as(p)
                    ops.grad_sinks[id(p)] = ops.GradSink(p, view, self)
                    self._sink_ids.append(id(p))
                    weakref.finalize(p, ops.grad_sinks.pop, id(p), None)
        # the table holds only weak references to this reducer; when it is collected its entries go with it
        weakref.finalize(self, GradReducer._drop_sinks, list(self._sink_ids))

    @staticmethod
    def _drop_sinks(ids, owner=None):
        from . import ops
        for i in ids:
            ent = ops.grad_sinks.get(i)
            if ent is not None and (ent.reducer() is None or ent.reducer() is owner):
                ops.grad_sinks.pop(i, None)

    def close(self):
        """detach from the parameters: remove the autograd hooks and the gradient sinks (the .grad views stay valid)"""
        if self._closed:
            return
        self._closed = True
        for h in self._hooks:
            h.remove()
        self._hooks = []
        GradReducer._drop_sinks(self._sink_ids, self)
        self._sink_ids = []

    def no_sync(self):
        """context manager for gradient accumulation (as DistributedDataParallel.no_sync): backward passes inside it add
        into the buckets but neither count parameters nor start collectives; the first backward after it reduces the
        accumulated sums.  Do not call prepare() between the micro-batches (it zeroes the buckets)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = prev
                self._rearm()
        return ctx()

    def _rearm(self):
        from . import ops
        self._reported.clear()
        for b in self.buckets:
            b.pending = len(b.params)
            b.handle = None
            b.streams = []
            b.slice_handles = []
        for i in self._sink_ids:
            ent = ops.grad_sinks.get(i)
            if ent is not None:
                ent.sites.clear()

    # -- per step ---------------------------------------------------------------------------
    def prepare(self):
        """zero the buckets (== zero_grad) and re-arm the hooks; call before each forward"""
        self._rearm()
        for b in self.buckets:
            b.flat.zero_()
            for p, off in zip(b.params, b.offsets):  # keep .grad pointing into the bucket
                n = p.numel()
                if p.grad is None or p.grad.data_ptr() != b.flat.data_ptr() + off * 4:
                    p.grad = b.flat[off:off + n].view_as(p)

    def _sink_report(self, p):
        """called by a backward that accumulated this parameter's gradient straight into its bucket view"""
        if not self._sync:
            return
        if id(p) in self._reported:
            raise RuntimeError(f"GradReducer(direct_grads=True): a parameter of shape {tuple(p.shape)} received its gradient twice "
                               "in one backward pass (shared weights?); construct the reducer with direct_grads=False")
        self._reported.add(id(p))
        self._count(p)

    def _row_slices(self, p) -> int:
        b = self._owner.get(id(p))
        if b is None or not self.active or not self._sync or len(b.params) != 1 or p.dim() != 2 or p.numel() * 4 < self.row_slice_min_bytes:
            return 1
        return self.row_slices

    def _sink_report_rows(self, p, row_lo: int, row_hi: int, last: bool):
        """rows [row_lo, row_hi) of a parameter that owns its bucket are final: reduce that slice now"""
        if not self._sync:
            return
        b = self._owner[id(p)]
        if id(p) in self._reported:
            raise RuntimeError("GradReducer(direct_grads=True): a parameter delivered in row slices also reported as a whole")
        ncol = p.shape[1]
        sl = b.flat[b.offsets[0] + row_lo * ncol:b.offsets[0] + row_hi * ncol]
        if sl.numel():
            if p.is_cuda:
                ls = self._launch_stream(p.device)
                ls.wait_stream(torch.cuda.current_stream(p.device))
                with torch.cuda.stream(ls):
                    b.slice_handles.append(dist.all_reduce(sl, op=self._op(), group=self.group, async_op=True))
            else:
                b.slice_handles.append(dist.all_reduce(sl, op=self._op(), group=self.group, async_op=True))
        if last:
            self._reported.add(id(p))
            b.pending -= 1                                # the bucket is complete; its collectives are the slices'

    def _hook(self, p):
        """autograd's post-accumulate hook; a parameter whose gradient went through its sink has been counted already"""
        if not self._sync or id(p) in self._reported:
            return
        self._count(p)

    def _count(self, p):
        b = self._owner[id(p)]
        b.pending -= 1
        if p.is_cuda:
            # the model may run its two encoder stacks on different HIP streams (modules.TWO_STREAMS): remember every
            # stream that produced a gradient of this bucket, so that the collective is ordered after all of them
            cur = torch.cuda.current_stream(p.device)
            if all(cur != s for s in b.streams):
                b.streams.append(cur)
        if b.pending == 0 and self.active:
            if p.is_cuda:
                # Launch from a side stream that waits for every producer: ProcessGroupNCCL orders its own stream after the
                # stream that is current at the call, so calling from a compute stream would first have to JOIN the model's
                # streams there (wait_stream) and stall whichever is ahead, once per bucket, in the middle of backward.
                ls = self._launch_stream(p.device)
                for s_ in b.streams:
                    ls.wait_stream(s_)
                with torch.cuda.stream(ls):
                    b.handle = dist.all_reduce(b.flat, op=self._op(), group=self.group, async_op=True)
            else:
                b.handle = dist.all_reduce(b.flat, op=self._op(), group=self.group, async_op=True)

    def _launch_stream(self, device):
        key = (device.type, device.index)
        if key not in self._launch_streams:
            self._launch_streams[key] = torch.cuda.Stream(device=device)
        return self._launch_streams[key]

    def _op(self):
        return dist.ReduceOp.AVG if self.fused_avg else dist.ReduceOp.SUM

    def wait(self):
        """join the outstanding all-reduces (parameters that received no gradient this step still get
        their bucket reduced here) and apply the 1/world average"""
        for b in self.buckets:
            if self.active:
                if b.slice_handles:                       # delivered and reduced in row slices (see _sink_report_rows)
                    for h in b.slice_handles:
                        h.wait()
                else:
                    if b.handle is None:
                        b.handle = dist.all_reduce(b.flat, op=self._op(), group=self.group, async_op=True)
                    b.handle.wait()
                if self.average and not self.fused_avg:
                    b.flat.mul_(1.0 / self.world)

    @property
    def num_buckets(self) -> int:
        return len(self.buckets)

    def grad_bytes(self) -> int:
        return sum(b.flat.numel() for b in self.buckets) * 4


def shard_batch(total: int, rank: int, world: int):
    """rows [lo, hi) of the global batch owned by `rank` (rank r takes rows [128r, 128r+128))"""
    per = total // world
    assert per * world == total, "global batch must divide evenly over the ranks"
    return rank * per, (rank + 1) * per
