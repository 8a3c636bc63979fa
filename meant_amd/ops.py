"""torch.autograd.Function wrappers over the C ABI (include/meant_hip.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape.  Every
numerical operation of the hot path is a call into libmeant_hip.so.  The activation dtype
(torch.float32 or torch.bfloat16) of the input selects the precision tier; parameters stay
fp32 (master weights), gradients of parameters are produced in fp32.
"""
from __future__ import annotations

import math
import os
import threading
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import lib, check, F32, BF16, EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_SIGMOID


# ---------------------------------------------------------------------------------------------
def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"meant_amd: activations must be float32 or bfloat16, got {t.dtype}")


def _need_gpu(*ts):
    """every tensor must live on the CURRENT HIP device: the library acts on the calling thread's current device (its
    streams, its per-device kernel state), so a tensor of another GPU would be a cross-device access.  Callers that
    drive several GPUs from one process (nn.DataParallel replicas) already run each replica under its own
    torch.cuda.device(...)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("meant_amd: this op runs only on an MI355X (HIP) device; got a tensor on "
                               f"{t.device}.  There is no CPU fallback.")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"meant_amd: tensor on {t.device} but the current device is cuda:{cur}; wrap the call in "
                               "`with torch.cuda.device(tensor.device):`")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# weight cache: compute-dtype copy and transposed copy of an fp32 parameter, refreshed when the
# parameter is updated in place (optimizer step bumps ._version) or replaced.
class _WeightCache:
    """Entries are keyed on the PARAMETER OBJECTS behind the tensors handed in (a view such as conv.weight.view(N, K) is
    keyed on its base parameter + geometry), carry weak references to them and are dropped by a finalizer when a parameter
    dies.  Tensors without a stable identity (results of F.pad / cat / arithmetic) are converted without being cached, so
    nothing can accumulate from per-step temporaries.  `pad_rows` appends zero rows to the (concatenated) weight before
    the conversion: the padded bf16 / transposed copies of a vocabulary matrix are built once per optimizer step instead
    of once per call.  Thread-safe (autograd worker threads, nn.DataParallel replicas)."""

    def __init__(self):
        self._store = {}
        self._epoch = 0
        self._lock = threading.RLock()

    def invalidate(self):
        """call after parameters were modified through raw device pointers (the fused optimizer), which does not
        bump their autograd version counters"""
        self._epoch += 1

    @staticmethod
    def _root(p):
        base = p._base if p._base is not None else p
        return base if (base.is_leaf and not isinstance(base, torch.nn.parameter.UninitializedParameter)) else None

    def _drop(self, key):
        with self._lock:
            self._store.pop(key, None)

    def _build(self, params, dtype, transposed, pad_rows):
        with torch.no_grad():
            w = params[0].detach() if len(params) == 1 else torch.cat([p.detach() for p in params], dim=0)
            if pad_rows:
                w = torch.nn.functional.pad(w, (0, 0, 0, pad_rows))
            w = _c(w.float())
            N, K = w.shape
            dd = F32 if dtype == torch.float32 else BF16
            if transposed:
                out = torch.empty((K, N), device=w.device, dtype=dtype)
                check(lib.meant_transpose2d(_p(w), F32, _p(out), dd, N, K, _stream()), "transpose2d")
            elif dtype == torch.float32:
                out = w
            else:
                out = torch.empty((N, K), device=w.device, dtype=dtype)
                check(lib.meant_cast(_p(w), F32, _p(out), dd, N * K, _stream()), "cast")
        return out

    def get(self, params, dtype: torch.dtype, transposed: bool, pad_rows: int = 0):
        """params: tuple of [N_i, K] fp32 tensors, concatenated along N (+ pad_rows zero rows)."""
        roots = tuple(self._root(p) for p in params)
        if any(r is None for r in roots):
            return self._build(params, dtype, transposed, pad_rows)
        key = (tuple((id(r), tuple(p.shape), p.storage_offset(), p.stride()) for r, p in zip(roots, params)), dtype, transposed,
               pad_rows, params[0].device)
        ver = (self._epoch,) + tuple((r._version, r.data_ptr()) for r in roots)
        with self._lock:
            hit = self._store.get(key)
            # id() values are recycled once a parameter is freed: an entry is valid only for the very same objects
            if hit is not None and hit[0] == ver and all(wr() is r for wr, r in zip(hit[2], roots)):
                return hit[1]
        out = self._build(params, dtype, transposed, pad_rows)
        with self._lock:
            fresh = key not in self._store
            self._store[key] = (ver, out, tuple(weakref.ref(r) for r in roots))
            if fresh:
                for r in roots:
                    weakref.finalize(r, self._drop, key)
        return out

    def __len__(self):
        return len(self._store)

    def clear(self):
        with self._lock:
            self._store.clear()


weights = _WeightCache()


# ---------------------------------------------------------------------------------------------
def _rmsnorm_fwd_raw(x, scale, eps, drop_p, seed):
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    rinv = torch.empty(rows, device=x.device, dtype=torch.float32)
    sc = _c(scale.detach().float())
    check(lib.meant_rmsnorm_fwd(_p(x), _p(sc), _p(y), _p(rinv), rows, d, eps, drop_p, seed, _dt(x), _stream()), "rmsnorm_fwd")
    return y, sc, rinv


def _rmsnorm_bwd_raw(dy, x, sc, rinv, eps, drop_p, seed, dres=None, gelu_pre=None):
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dscale = torch.empty(d, device=x.device, dtype=torch.float32)
    wsb = lib.meant_rmsnorm_bwd_ws(rows, d)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    check(lib.meant_rmsnorm_bwd(_p(dy), _p(x), _p(sc), _p(rinv), _p(dx), _p(dscale), rows, d, eps, drop_p, seed, _p(dres), _p(gelu_pre),
                                _dt(x), _p(ws), wsb, _stream()), "rmsnorm_bwd")
    return dx, dscale


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, eps, drop_p, seed):
        _need_gpu(x, scale)
        x = _c(x)
        y, sc, rinv = _rmsnorm_fwd_raw(x, scale, eps, drop_p, seed)
        ctx.save_for_backward(x, sc, rinv)
        ctx.args = (eps, drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sc, rinv = ctx.saved_tensors
        dx, dscale = _rmsnorm_bwd_raw(_c(dy), x, sc, rinv, *ctx.args)
        return dx, dscale, None, None, None


def rmsnorm(x, scale, eps=1e-8, drop_p=0.0, seed=0):
    return _RMSNorm.apply(x, scale, float(eps), float(drop_p), int(seed))


class _RMSNormPartial(torch.autograd.Function):
    """the partial / bias forms of the reference class (utils/rms_norm.py:44-57): statistics over the first d_part elements of a
    row, optional learned offset.  Not on the MEANT path (every model uses p = -1, bias = False); generic kernels."""

    @staticmethod
    def forward(ctx, x, scale, offset, d_part, eps):
        _need_gpu(x, scale, offset)
        x = _c(x)
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        rinv = torch.empty(rows, device=x.device, dtype=torch.float32)
        sc = _c(scale.detach().float())
        off = _c(offset.detach().float()) if offset is not None else None
        check(lib.meant_rmsnorm_partial_fwd(_p(x), _p(sc), _p(off), _p(y), _p(rinv), rows, d, int(d_part), eps, _dt(x), _stream()),
              "rmsnorm_partial_fwd")
        ctx.save_for_backward(x, sc, rinv)
        ctx.args = (int(d_part), eps, offset is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sc, rinv = ctx.saved_tensors
        d_part, eps, has_off = ctx.args
        dy = _c(dy)
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        dscale = torch.empty(d, device=x.device, dtype=torch.float32)
        doff = torch.empty(d, device=x.device, dtype=torch.float32) if has_off else None
        wsb = lib.meant_rmsnorm_bwd_ws(rows, d)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        check(lib.meant_rmsnorm_partial_bwd(_p(dy), _p(x), _p(sc), _p(rinv), _p(dx), _p(dscale), _p(doff), rows, d, d_part, eps, _dt(x),
                                            _p(ws), wsb, _stream()), "rmsnorm_partial_bwd")
        return dx, dscale, doff, None, None


def rmsnorm_partial(x, scale, offset, d_part, eps=1e-8):
    return _RMSNormPartial.apply(x, scale, offset, int(d_part), float(eps))


class _RMSNormFork(torch.autograd.Function):
    """(RMSNorm(x), x): the second output is x itself, to be used as the residual operand further down
    (meant/meant.py:71,74).  Backward receives both gradients and adds the residual one inside the RMSNorm
    backward kernel instead of in a separate elementwise pass."""

    @staticmethod
    def forward(ctx, x, scale, eps):
        _need_gpu(x, scale)
        x = _c(x)
        y, sc, rinv = _rmsnorm_fwd_raw(x, scale, eps, 0.0, 0)
        ctx.save_for_backward(x, sc, rinv)
        ctx.eps = eps
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, sc, rinv = ctx.saved_tensors
        if dy is None:
            return dres, None, None
        dres = _c(dres) if dres is not None else None
        dx, dscale = _rmsnorm_bwd_raw(_c(dy), x, sc, rinv, ctx.eps, 0.0, 0, dres=dres)
        return dx, dscale, None


def rmsnorm_fork(x, scale, eps=1e-8):
    return _RMSNormFork.apply(x, scale, float(eps))


class _GeluRMSNorm(torch.autograd.Function):
    """RMSNorm(gelu(pre)) given both the activation a = gelu(pre) (emitted by the GEMM epilogue) and pre: the
    backward returns the gradient w.r.t. pre, the GELU derivative being applied inside the RMSNorm backward
    kernel (meant/meant.py:64).  `a` is treated as an intermediate: its gradient slot is not used."""

    @staticmethod
    def forward(ctx, a, pre, scale, eps, drop_p, seed):
        _need_gpu(a, pre, scale)
        a = _c(a)
        y, sc, rinv = _rmsnorm_fwd_raw(a, scale, eps, drop_p, seed)
        ctx.save_for_backward(a, _c(pre), sc, rinv)
        ctx.args = (eps, drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, pre, sc, rinv = ctx.saved_tensors
        dpre, dscale = _rmsnorm_bwd_raw(_c(dy), a, sc, rinv, *ctx.args, gelu_pre=pre)
        return None, dpre, dscale, None, None, None


class _LinearPre(torch.autograd.Function):
    """Linear with GELU epilogue that hands out (gelu(h), h) with h = x W^T + b; gradients flow through h only
    (the consumer, _GeluRMSNorm, chains the GELU derivative itself)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, weight)
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        w_c = weights.get((weight,), x.dtype, False)
        bias_f = _c(bias.detach().float()) if bias is not None else None
        y, pre = _linear_fwd_raw(x2, w_c, bias_f, None, EPI_GELU, True)
        ctx.weight, ctx.bias, ctx.has_bias, ctx.in_shape = weight, bias, bias is not None, shp
        _claim(weight, "linear"); _claim(bias, "linear")
        ctx.save_for_backward(x2)
        N = weight.shape[0]
        yv, pv = y.view(*shp[:-1], N), pre.view(*shp[:-1], N)
        ctx.mark_non_differentiable(yv)
        ctx.set_materialize_grads(False)     # no zero tensor for the non-differentiable output in backward
        return yv, pv

    @staticmethod
    def backward(ctx, _dy_unused, dpre):
        if dpre is None:
            return None, None, None
        (x2,) = ctx.saved_tensors
        N = ctx.weight.shape[0]
        d2 = _c(dpre).view(-1, N)
        dx, dw, db = _linear_bwd_raw(d2, x2, (ctx.weight,), ctx.needs_input_grad[0], ctx.has_bias, bias_param=ctx.bias)
        return (dx.view(ctx.in_shape) if dx is not None else None), dw, db


def linear_gelu_rmsnorm(x, weight, bias, scale, eps=1e-8, drop_p=0.0, seed=0):
    """RMSNorm(gelu(x W^T + b)) as two fused calls forward and two backward (meant/meant.py:64 / :107)."""
    a, pre = _LinearPre.apply(x, weight, bias)
    return _GeluRMSNorm.apply(a, pre, scale, float(eps), float(drop_p), int(seed))


# ---- pooled forms: the norms next to the sequence mean-pool (see _PooledLinearCat below) -------------------------------
def pooled_norm_ok(x, group_rows: int) -> bool:
    d = x.shape[-1]
    return bool(lib.meant_rmsnorm_pooled_ok(x.numel() // d, d, int(group_rows)))


def _rmsnorm_bwd_pooled_raw(dy, dy_pooled, x, sc, rinv, S, eps, drop_p, seed, dres, dres_pooled, gelu_pre):
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dscale = torch.empty(d, device=x.device, dtype=torch.float32)
    wsb = lib.meant_rmsnorm_bwd_ws(rows, d)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    check(lib.meant_rmsnorm_bwd_pooled(_p(dy), int(dy_pooled), _p(x), _p(sc), _p(rinv), _p(dx), _p(dscale), rows, d, S, eps, drop_p, seed,
                                       _p(dres), int(dres_pooled), _p(gelu_pre), _dt(x), _p(ws), wsb, _stream()), "rmsnorm_bwd_pooled")
    return dx, dscale


class _RMSNormForkPooled(torch.autograd.Function):
    """(RMSNorm(x), mean over each sequence of x): the second output replaces the residual operand x of _RMSNormFork when the
    only thing the residual feeds is the mean-pool.  x [G, S, d]; the mean is float [G, d] and comes out of the norm's own
    pass over x; its gradient is added row-wise inside the backward kernel."""

    @staticmethod
    def forward(ctx, x, scale, eps):
        _need_gpu(x, scale)
        x = _c(x)
        G, S, d = x.shape
        y = torch.empty_like(x)
        rinv = torch.empty(G * S, device=x.device, dtype=torch.float32)
        xm = torch.empty((G, d), device=x.device, dtype=torch.float32)
        sc = _c(scale.detach().float())
        check(lib.meant_rmsnorm_fwd_pooled(_p(x), _p(sc), _p(y), _p(rinv), _p(xm), G * S, d, S, 1, 0, eps, 0.0, 0, _dt(x), _stream()),
              "rmsnorm_fwd_pooled")
        ctx.save_for_backward(x, sc, rinv)
        ctx.eps, ctx.S = eps, S
        return y, xm

    @staticmethod
    def backward(ctx, dy, dxm):
        x, sc, rinv = ctx.saved_tensors
        if dxm is None:
            dx, dscale = _rmsnorm_bwd_raw(_c(dy), x, sc, rinv, ctx.eps, 0.0, 0)
        else:
            dx, dscale = _rmsnorm_bwd_pooled_raw(_c(dy), False, x, sc, rinv, ctx.S, ctx.eps, 0.0, 0, _c(dxm.float()), True, None)
        return dx, dscale, None


class _GeluRMSNormPooled(torch.autograd.Function):
    """mean over each sequence of dropout(RMSNorm(gelu(pre))) given the pre-activation `pre` [G, S, d] alone: float [G, d].
    Neither gelu(pre) nor the normalised tokens are ever written -- both are formed on load, forward and backward --, and
    the backward reads the [G, d] gradient of the mean."""

    @staticmethod
    def forward(ctx, pre, scale, eps, drop_p, seed):
        _need_gpu(pre, scale)
        pre = _c(pre)
        G, S, d = pre.shape
        rinv = torch.empty(G * S, device=pre.device, dtype=torch.float32)
        hm = torch.empty((G, d), device=pre.device, dtype=torch.float32)
        sc = _c(scale.detach().float())
        check(lib.meant_rmsnorm_fwd_pooled(_p(pre), _p(sc), None, _p(rinv), _p(hm), G * S, d, S, 0, 1, eps, drop_p, seed, _dt(pre), _stream()),
              "rmsnorm_fwd_pooled")
        ctx.save_for_backward(pre, sc, rinv)
        ctx.args, ctx.S = (eps, drop_p, seed), S
        return hm

    @staticmethod
    def backward(ctx, dhm):
        pre, sc, rinv = ctx.saved_tensors
        eps, drop_p, seed = ctx.args
        d = pre.shape[-1]
        rows = pre.numel() // d
        dpre = torch.empty_like(pre)
        dscale = torch.empty(d, device=pre.device, dtype=torch.float32)
        wsb = lib.meant_rmsnorm_bwd_ws(rows, d)
        ws = torch.empty(wsb, device=pre.device, dtype=torch.uint8)
        dh = _c(dhm.float())
        check(lib.meant_rmsnorm_bwd_pooled(_p(dh), 1, None, _p(sc), _p(rinv), _p(dpre), _p(dscale), rows, d, ctx.S, eps, drop_p, seed,
                                           None, 0, _p(pre), _dt(pre), _p(ws), wsb, _stream()), "rmsnorm_bwd_pooled")
        return dpre, dscale, None, None, None


def rmsnorm_fork_pooled(x, scale, eps=1e-8):
    return _RMSNormForkPooled.apply(x, scale, float(eps))


def linear_gelu_rmsnorm_pooled(x, weight, bias, scale, eps=1e-8, drop_p=0.0, seed=0):
    """mean_s(dropout(RMSNorm(gelu(x W^T + b)))) -> float [G, d]; the GEMM stores the pre-activation only"""
    pre = linear(x, weight, bias)
    return _GeluRMSNormPooled.apply(pre, scale, float(eps), float(drop_p), int(seed))


# ---- RMSNorm folded into the Linear that consumes it (bf16 tier) ---------------------------------------------------------
# Linear(RMSNorm(x)) = r (x) (x W'^T) + b with W' = W diag(g), r = 1 / (rms(x) + eps) (utils/rms_norm.py:40-57 followed by the
# nn.Linear of meant/meant.py:62 / :104): the normalisation is a per-row factor in the GEMM epilogue, so the normalised tensor
# is never written, read or saved -- forward: one statistics pass over x instead of a read + write; backward: no pass at all,
# the norm's own term d x = ... - kcoef x rides the input-gradient GEMM's epilogue, and the backward of the NEXT norm down the
# layer (which produces the gradient of this Linear's output anyway) emits everything that term needs in the same pass
# (meant_rmsnorm_bwd_chain).
# AUTO by default: on when one saved [tokens, d] tensor per layer of the stack in flight adds up to FOLD_AUTO_BYTES or more
# (layers x tokens x d x 2 B: the memory the fold gives back), off below.  Measured on the bench step (DESIGN.md section 6,
# "RMSNorm folded into the consumer Linear") the fold saves 0.25 ms of forward passes and the whole RMSNorm backward of
# encode2[0] (0.66 ms text, 0.24 ms vision per layer), but the chained norm backward that has to emit the scaled gradient, the
# row coefficients and the bias gradient on top of its own work costs 0.63 ms more than the plain kernel (a third accumulator
# set per lane: 256 registers, scheduling fences) and the extended GEMM epilogue 0.06 ms: the step comes out even to 2 % behind
# (2963-3003 against 3013-3024 samples/s; 3050 against 3059 late in round 3; the 12-layer MLM pretrainer at 64 sequences 794 k
# against 810 k tokens/s).  What the fold always saves is MEMORY: the normalised tensor of every folded norm.  At 12 encoder
# layers (the reference CLI's default depth, in_loop_train.py:418) and 128 samples per GPU that is the difference between the
# step fitting in one pass (253-259 samples/s) and needing two micro-batches (247) or recomputation (218): hence the rule --
# text stack there 12 layers x 786,432 tokens (128 x lag 12 x 512) x 768 x 2 B = 14.5 GB, vision stack 5.5 GB, both fold; the
# same depth at 32 per GPU (3.6 / 1.4 GB), the 12-layer MLM pretrainer (0.6 GB) and the one-layer headline step (1.2 / 0.5 GB)
# keep the separate norm.
# MEANT_FUSE_NORM_LINEAR=1 / =0 (or ops.FUSE_NORM_LINEAR = True / False) force it on / off; the parity tests run both paths.
_fold_env = os.environ.get("MEANT_FUSE_NORM_LINEAR")
FUSE_NORM_LINEAR = None if _fold_env is None else (_fold_env == "1")      # None: by the size of the stack in flight
FOLD_AUTO_BYTES = 4 << 30                                                 # sized for 288 GB of HBM per GPU
_stack_bytes = 0


def set_stack_hint(layers: int, tokens: int, d: int) -> None:
    """what one bf16 [tokens, d] tensor per layer of a stack adds up to.  Only a DEFAULT for encoder layers that are called on
    their own: the model classes decide per stack (stack_fold) and hand the decision to every layer as an argument, so that a
    layer recomputed under torch.utils.checkpoint takes the path its forward took whatever ran in between."""
    global _stack_bytes
    _stack_bytes = int(layers) * int(tokens) * int(d) * 2


def fold_wanted() -> bool:
    return bool(FUSE_NORM_LINEAR) if FUSE_NORM_LINEAR is not None else _stack_bytes >= FOLD_AUTO_BYTES


def stack_fold(layers: int, tokens: int, d: int) -> bool:
    """the fold decision of ONE stack of encoder layers (also leaves the hint for layers called outside a stack)"""
    set_stack_hint(layers, tokens, d)
    return fold_wanted()


fold_calls = [0, 0]                                                       # [separate, folded] decisions taken so far (bench.py reports them)


def norm_linear_ok(x, weight, wanted=None) -> bool:
    """shapes the folded path covers: bf16 tier, K a multiple of 64, a packed norm width, 16-byte aligned rows.
    wanted: the stack's decision (modules._run_stack); None = by the process-wide hint"""
    d = x.shape[-1]
    rows = x.numel() // d
    want = fold_wanted() if wanted is None else bool(wanted)
    ok = bool(want and x.dtype == torch.bfloat16 and x.is_cuda and d % 64 == 0 and weight.shape[1] == d
              and weight.shape[0] % 8 == 0 and rows > 0 and lib.meant_rmsnorm_pooled_ok(rows, d, rows))
    fold_calls[ok] += 1
    return ok


def _scaled_weight(weight, gain):
    """W' = W diag(g) in fp32 (kept for the backward) and its bf16 copy"""
    w_f, g_f = _c(weight.detach().float()), _c(gain.detach().float())
    N, K = w_f.shape
    wp = torch.empty_like(w_f)
    check(lib.meant_colscale(_p(w_f), _p(g_f), _p(wp), N, K, _stream()), "colscale")
    return w_f, g_f, wp, cast(wp, torch.bfloat16)


def _norm_linear_backward(ctx, dpre_s, kcoef, x2, wp, w_f, g_f, dres2, dres_pooled, group_rows):
    """the two backward GEMMs of the folded Linear and the weight-side chain rule: returns (dx2, dW or None, dg)"""
    M, K = x2.shape
    N = wp.shape[0]
    dt = _dt(x2)
    wpT = torch.empty((K, N), device=x2.device, dtype=x2.dtype)
    check(lib.meant_transpose2d(_p(wp), F32, _p(wpT), dt, N, K, _stream()), "transpose2d")
    dx = torch.empty_like(x2)
    check(lib.meant_linear_bwd_dx_norm(_p(dpre_s), N, _p(wpT), _p(x2), K, _p(kcoef), _p(dres2), K if dres2 is not None else 0,
                                       _p(dres_pooled), int(group_rows), _p(dx), K, M, N, K, dt, _stream()), "linear_bwd_dx_norm")
    dwp = torch.zeros((N, K), device=x2.device, dtype=torch.float32)
    _bwd_dw(dpre_s, x2, dwp, None)
    dg = torch.zeros(K, device=x2.device, dtype=torch.float32)
    sink = _sink_of(ctx.weight, "linear") if grad_sinks else None
    if sink is not None and sink.view.shape == (N, K) and sink.view.is_contiguous():
        check(lib.meant_colscale_bwd(_p(dwp), _p(w_f), _p(g_f), _p(sink.view), _p(dg), N, K, _stream()), "colscale_bwd")
        sink.report(ctx.weight)
        return dx, None, dg
    dw = torch.zeros((N, K), device=x2.device, dtype=torch.float32)
    check(lib.meant_colscale_bwd(_p(dwp), _p(w_f), _p(g_f), _p(dw), _p(dg), N, K, _stream()), "colscale_bwd")
    return dx, dw, dg


def _bias_grad_target(ctx, N, device):
    """where the chained norm backward adds the Linear's bias gradient: the bias's gradient sink, or a fresh zero vector"""
    if ctx.bias is None:
        return torch.zeros(N, device=device, dtype=torch.float32), False
    sink = _sink_of(ctx.bias, "linear") if grad_sinks else None
    if sink is not None and sink.view.shape == (N,) and sink.view.is_contiguous():
        return sink.view, True
    return torch.zeros(N, device=device, dtype=torch.float32), False


class _NormLinearGeluNorm(torch.autograd.Function):
    """(dropout(RMSNorm_3(gelu(Linear(RMSNorm_0(x))))), x): the encode2 chain of an encoder layer up to its last Linear
    (meant/meant.py:61-64, :103-107) with RMSNorm_0 folded into the Linear; the second output is x itself, the residual
    operand further down (as _RMSNormFork).  Saved: x, its row statistics, the pre-activation, the activation."""

    @staticmethod
    def forward(ctx, x, gain0, eps0, weight, bias, gain3, eps3, drop_p, seed):
        _need_gpu(x, weight, gain0, gain3)
        shp = x.shape
        K = shp[-1]
        x2 = _c(x).view(-1, K)
        M, N = x2.shape[0], weight.shape[0]
        w_f, g_f, wp, wp_c = _scaled_weight(weight, gain0)
        bias_f = _c(bias.detach().float()) if bias is not None else torch.zeros(N, device=x.device, dtype=torch.float32)
        r0 = torch.empty(M, device=x.device, dtype=torch.float32)
        check(lib.meant_rmsnorm_stats(_p(x2), _p(r0), M, K, eps0, _dt(x2), _stream()), "rmsnorm_stats")
        a = torch.empty((M, N), device=x.device, dtype=x.dtype)
        pre = torch.empty_like(a)
        check(lib.meant_linear_fwd_rowscale(_p(x2), K, _p(wp_c), _p(bias_f), _p(r0), None, 0, _p(a), N, _p(pre), M, N, K, EPI_GELU,
                                            _dt(x2), _stream()), "linear_fwd_rowscale")
        y, sc3, rinv3 = _rmsnorm_fwd_raw(a, gain3, eps3, drop_p, seed)
        ctx.weight, ctx.bias = weight, bias
        _claim(weight, "linear"); _claim(bias, "linear")
        ctx.save_for_backward(x2, r0, a, pre, sc3, rinv3, wp, w_f, g_f, bias_f)
        ctx.args = (eps0, eps3, drop_p, seed, shp)
        return y.view(*shp[:-1], N), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x2, r0, a, pre, sc3, rinv3, wp, w_f, g_f, bias_f = ctx.saved_tensors
        eps0, eps3, drop_p, seed, shp = ctx.args
        M, K = x2.shape
        N = wp.shape[0]
        dres2 = _c(dres).view(M, K) if dres is not None else None
        if dy is None:
            return dres, None, None, None, None, None, None, None, None
        dy2 = _c(dy).view(M, N)
        dpre_s = torch.empty_like(pre)
        kcoef = torch.empty(M, device=x2.device, dtype=torch.float32)
        dscale3 = torch.empty(N, device=x2.device, dtype=torch.float32)
        db, db_sunk = _bias_grad_target(ctx, N, x2.device)
        wsb = lib.meant_rmsnorm_bwd_ws(M, N)
        ws = torch.empty(wsb, device=x2.device, dtype=torch.uint8)
        check(lib.meant_rmsnorm_bwd_chain(_p(dy2), 0, _p(a), _p(sc3), _p(rinv3), _p(dpre_s), _p(dscale3), M, N, 1, eps3, drop_p, seed, _p(pre),
                                          _p(r0), _p(bias_f), eps0, K, _p(kcoef), _p(db), _dt(x2), _p(ws), wsb, _stream()), "rmsnorm_bwd_chain")
        if db_sunk:
            _sink_of(ctx.bias, "linear").report(ctx.bias)
        dx, dw, dg0 = _norm_linear_backward(ctx, dpre_s, kcoef, x2, wp, w_f, g_f, dres2, None, 1)
        return dx.view(shp), dg0, None, dw, (None if (db_sunk or ctx.bias is None) else db), dscale3, None, None, None


def norm_linear_gelu_norm(x, gain0, eps0, weight, bias, gain3, eps3, drop_p=0.0, seed=0):
    return _NormLinearGeluNorm.apply(x, gain0, float(eps0), weight, bias, gain3, float(eps3), float(drop_p), int(seed))


class _NormLinearGeluNormPooled(torch.autograd.Function):
    """the same chain in the LAST encoder layer of a stack, where its only consumer is the sequence mean-pool
    (_GeluRMSNormPooled / _RMSNormForkPooled): (mean_s dropout(RMSNorm_3(gelu(Linear(RMSNorm_0(x))))), mean_s x), both float
    [G, d].  x [G, S, d].  Neither the normalised input, nor the activation, nor the normalised activation is ever written."""

    @staticmethod
    def forward(ctx, x, gain0, eps0, weight, bias, gain3, eps3, drop_p, seed):
        _need_gpu(x, weight, gain0, gain3)
        x = _c(x)
        G, S, K = x.shape
        M, N = G * S, weight.shape[0]
        x2 = x.view(M, K)
        w_f, g_f, wp, wp_c = _scaled_weight(weight, gain0)
        bias_f = _c(bias.detach().float()) if bias is not None else torch.zeros(N, device=x.device, dtype=torch.float32)
        r0 = torch.empty(M, device=x.device, dtype=torch.float32)
        xm = torch.empty((G, K), device=x.device, dtype=torch.float32)
        check(lib.meant_rmsnorm_fwd_pooled(_p(x2), _p(g_f), None, _p(r0), _p(xm), M, K, S, 1, 0, eps0, 0.0, 0, _dt(x2), _stream()),
              "rmsnorm_fwd_pooled")                                   # y == NULL: statistics and the means of x
        pre = torch.empty((M, N), device=x.device, dtype=x.dtype)
        check(lib.meant_linear_fwd_rowscale(_p(x2), K, _p(wp_c), _p(bias_f), _p(r0), None, 0, _p(pre), N, None, M, N, K, EPI_NONE,
                                            _dt(x2), _stream()), "linear_fwd_rowscale")
        rinv3 = torch.empty(M, device=x.device, dtype=torch.float32)
        hm = torch.empty((G, N), device=x.device, dtype=torch.float32)
        sc3 = _c(gain3.detach().float())
        check(lib.meant_rmsnorm_fwd_pooled(_p(pre), _p(sc3), None, _p(rinv3), _p(hm), M, N, S, 0, 1, eps3, drop_p, seed, _dt(pre), _stream()),
              "rmsnorm_fwd_pooled")
        ctx.weight, ctx.bias = weight, bias
        _claim(weight, "linear"); _claim(bias, "linear")
        ctx.save_for_backward(x2, r0, pre, sc3, rinv3, wp, w_f, g_f, bias_f)
        ctx.args = (eps0, eps3, drop_p, seed, (G, S, K))
        return hm, xm

    @staticmethod
    def backward(ctx, dhm, dxm):
        x2, r0, pre, sc3, rinv3, wp, w_f, g_f, bias_f = ctx.saved_tensors
        eps0, eps3, drop_p, seed, (G, S, K) = ctx.args
        M, N = G * S, wp.shape[0]
        if dhm is None:
            raise RuntimeError("meant_amd: the pooled encoder tail received no gradient for its pooled activations")
        dh = _c(dhm.float())
        dpre_s = torch.empty_like(pre)
        kcoef = torch.empty(M, device=x2.device, dtype=torch.float32)
        dscale3 = torch.empty(N, device=x2.device, dtype=torch.float32)
        db, db_sunk = _bias_grad_target(ctx, N, x2.device)
        wsb = lib.meant_rmsnorm_bwd_ws(M, N)
        ws = torch.empty(wsb, device=x2.device, dtype=torch.uint8)
        check(lib.meant_rmsnorm_bwd_chain(_p(dh), 1, None, _p(sc3), _p(rinv3), _p(dpre_s), _p(dscale3), M, N, S, eps3, drop_p, seed, _p(pre),
                                          _p(r0), _p(bias_f), eps0, K, _p(kcoef), _p(db), _dt(x2), _p(ws), wsb, _stream()), "rmsnorm_bwd_chain")
        if db_sunk:
            _sink_of(ctx.bias, "linear").report(ctx.bias)
        dxm_f = _c(dxm.float()) if dxm is not None else None
        dx, dw, dg0 = _norm_linear_backward(ctx, dpre_s, kcoef, x2, wp, w_f, g_f, None, dxm_f, S)
        return dx.view(G, S, K), dg0, None, dw, (None if (db_sunk or ctx.bias is None) else db), dscale3, None, None, None


def norm_linear_gelu_norm_pooled(x, gain0, eps0, weight, bias, gain3, eps3, drop_p=0.0, seed=0):
    return _NormLinearGeluNormPooled.apply(x, gain0, float(eps0), weight, bias, gain3, float(eps3), float(drop_p), int(seed))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x, gamma, beta)
        x = _c(x)
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        g, b = _c(gamma.detach().float()), _c(beta.detach().float())
        check(lib.meant_layernorm_fwd(_p(x), _p(g), _p(b), _p(y), _p(stats), rows, d, eps, _dt(x), _stream()), "layernorm_fwd")
        ctx.save_for_backward(x, g, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, stats = ctx.saved_tensors
        dy = _c(dy)
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        dg = torch.empty(d, device=x.device, dtype=torch.float32)
        db = torch.empty(d, device=x.device, dtype=torch.float32)
        wsb = lib.meant_rmsnorm_bwd_ws(rows, d)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
        check(lib.meant_layernorm_bwd(_p(dy), _p(x), _p(g), _p(stats), _p(dx), _p(dg), _p(db), rows, d, _dt(x), _p(ws), wsb,
                                      _stream()), "layernorm_bwd")
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


# ---------------------------------------------------------------------------------------------
def _linear_fwd_raw(x2, w_c, bias_f, residual2, epilogue, want_preact):
    M, K = x2.shape
    N = w_c.shape[0]
    y = torch.empty((M, N), device=x2.device, dtype=x2.dtype)
    pre = torch.empty_like(y) if want_preact else None
    check(lib.meant_linear_fwd(_p(x2), x2.stride(0), _p(w_c), _p(bias_f), _p(residual2),
                               residual2.stride(0) if residual2 is not None else 0, _p(y), N, _p(pre), M, N, K, epilogue,
                               _dt(x2), _stream()), "linear_fwd")
    return y, pre


def _bwd_dw(dy2, x2, dw, db):
    """dw [N, K] += dy2^T x2, db [N] += colsum(dy2) (float accumulators)"""
    M, N = dy2.shape
    K = x2.shape[1]
    dt = _dt(dy2)
    wsb = lib.meant_linear_bwd_dw_ws(M, N, K, dt)          # non-zero only with the "deterministic" option
    ws = torch.empty(wsb, device=dy2.device, dtype=torch.uint8) if wsb else None
    check(lib.meant_linear_bwd_dw(_p(dy2), dy2.stride(0), _p(x2), x2.stride(0), _p(dw), _p(db), M, N, K, dt, _p(ws), wsb, _stream()),
          "linear_bwd_dw")


# Gradient sinks: a gradient reducer (meant_amd.parallel.GradReducer(direct_grads=True)) registers, per parameter, the
# view of its flat bucket that IS the parameter's .grad.  A Linear whose weight (and bias) have a sink accumulates dW / db
# straight into those views -- the C ABI's contract is "+=" -- and reports to the reducer, instead of handing autograd a
# freshly zeroed tensor that AccumulateGrad then adds to the same view with one more launch each (~100 small fills and adds
# per step).  A sink may only be used when ONE kind of consumer produces the parameter's gradient in a backward pass: a
# parameter that two different ops read (the tied word embedding / vocabulary decoder of the MLM pretrainer,
# pretrain_mlm.py:318-319) gets one contribution early and one late, and a report from the first would start the bucket's
# all-reduce before the second has landed.  Every op that reads a parameter therefore CLAIMS it in forward
# (`_claim(param, site)`); in backward a sink is handed out only if no other site claimed the parameter since the reducer's
# last `prepare()`, otherwise the op returns its gradient to autograd, whose AccumulateGrad node runs after all
# contributions and fires the reducer's ordinary hook.  (Recomputation under activation checkpointing claims the same site
# again, which is fine.)  Two calls of the SAME op kind on one parameter still report twice and raise in the reducer.
grad_sinks = {}


class GradSink:
    """one parameter's gradient sink: weak references to the parameter and to the reducer that owns the bucket (so that
    neither is kept alive by this table), the bucket view that is the parameter's .grad, and the sites that claimed the
    parameter in the current step"""
    __slots__ = ("param", "view", "reducer", "sites")

    def __init__(self, param, view, reducer):
        self.param, self.view, self.reducer, self.sites = weakref.ref(param), view, weakref.ref(reducer), set()

    def report(self, p):
        r = self.reducer()
        if r is not None:
            r._sink_report(p)

    def row_slices(self, p) -> int:
        """in how many row slices the reducer wants this parameter's gradient delivered (1: all at once)"""
        r = self.reducer()
        return r._row_slices(p) if r is not None else 1

    def report_rows(self, p, row_lo: int, row_hi: int, last: bool):
        """rows [row_lo, row_hi) of the parameter's gradient are final in the bucket view"""
        r = self.reducer()
        if r is not None:
            r._sink_report_rows(p, row_lo, row_hi, last)


def _lib_option(name: str) -> int:
    from . import _lib
    return _lib.get_option(name)


def _claim(param, site: str):
    """forward side of the sink protocol: remember that an op of kind `site` reads `param` in this step.  (Called from inside
    autograd.Function.forward, where grad mode is off: what matters is whether the parameter wants a gradient.)"""
    if grad_sinks and param is not None and param.requires_grad:
        ent = grad_sinks.get(id(param))
        if ent is not None and ent.param() is param:
            ent.sites.add(site)


def _sink_of(param, site: str):
    """the parameter's sink, or None if it has none, its reducer is gone, or another kind of op also reads the parameter
    in this step (then the gradient must go through autograd's accumulation)"""
    ent = grad_sinks.get(id(param))
    if ent is None or ent.param() is not param or ent.reducer() is None:
        return None
    if ent.sites and ent.sites != {site}:
        return None
    return ent


def _linear_bwd_raw(dy2, x2, params, need_dx, has_bias, pad_rows=0, bias_param=None, site="linear"):
    """returns dx2 (or None), dW [sum N_i (+ pad_rows), K] fp32, db [same] fp32 or None; dW / db are None when they went
    straight into the parameters' gradient sinks"""
    M, N = dy2.shape
    K = x2.shape[1]
    dx = None
    if need_dx:
        wT = weights.get(params, dy2.dtype, True, pad_rows)             # [K, N]
        dx = torch.empty((M, K), device=dy2.device, dtype=dy2.dtype)
        check(lib.meant_linear_bwd_dx(_p(dy2), dy2.stride(0), _p(wT), _p(dx), K, M, N, K, _dt(dy2), _stream()), "linear_bwd_dx")
    if len(params) == 1 and pad_rows == 0 and grad_sinks and (bias_param is not None or not has_bias):
        ws_, bs_ = _sink_of(params[0], site), (_sink_of(bias_param, site) if has_bias else None)
        if ws_ is not None and (not has_bias or bs_ is not None) and ws_.view.shape == (N, K) and ws_.view.is_contiguous():
            _bwd_dw(dy2, x2, ws_.view, bs_.view if has_bias else None)
            ws_.report(params[0])
            if has_bias:
                bs_.report(bias_param)
            return dx, None, None
    dw = torch.zeros((N, K), device=dy2.device, dtype=torch.float32)
    db = torch.zeros(N, device=dy2.device, dtype=torch.float32) if has_bias else None
    _bwd_dw(dy2, x2, dw, db)
    return dx, dw, db


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual).  epilogue in {NONE, GELU, SIGMOID} | RESIDUAL."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, epilogue):
        _need_gpu(x, weight)
        shp = x.shape
        x2 = _c(x).view(-1, shp[-1])
        res2 = _c(residual).view(-1, weight.shape[0]) if residual is not None else None
        w_c = weights.get((weight,), x.dtype, False)
        bias_f = _c(bias.detach().float()) if bias is not None else None
        epi = epilogue | (EPI_RESIDUAL if residual is not None else 0)
        y, pre = _linear_fwd_raw(x2, w_c, bias_f, res2, epi, bool(epilogue & EPI_GELU))
        ctx.epilogue = epilogue
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.weight, ctx.bias = weight, bias
        _claim(weight, "linear"); _claim(bias, "linear")
        ctx.save_for_backward(x2, pre if (epilogue & EPI_GELU) else (y if (epilogue & EPI_SIGMOID) else None))
        ctx.in_shape = shp
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        N = ctx.weight.shape[0]
        dy2 = _c(dy).view(-1, N)
        dres = dy if ctx.has_res else None
        if ctx.epilogue & EPI_GELU:
            d2 = torch.empty_like(dy2)
            check(lib.meant_gelu_bwd(_p(dy2), _p(aux), _p(d2), dy2.numel(), _dt(dy2), _stream()), "gelu_bwd")
            dy2 = d2
        elif ctx.epilogue & EPI_SIGMOID:
            d2 = torch.empty_like(dy2)
            check(lib.meant_sigmoid_bwd(_p(dy2), _p(aux), _p(d2), dy2.numel(), _dt(dy2), _stream()), "sigmoid_bwd")
            dy2 = d2
        dx, dw, db = _linear_bwd_raw(dy2, x2, (ctx.weight,), ctx.needs_input_grad[0], ctx.has_bias, bias_param=ctx.bias)
        return (dx.view(ctx.in_shape) if dx is not None else None), dw, db, dres, None


def linear(x, weight, bias=None, residual=None, epilogue=EPI_NONE):
    return _Linear.apply(x, weight, bias, residual, epilogue)


# ---------------------------------------------------------------------------------------------
def _gemm_f32(A, B, C, M, N, K, sA, sB, sC, alpha=1.0, accumulate=False):
    """C[m,n] (+)= alpha * sum_k A(m,k) B(k,n) on fp32 tensors with explicit (row, col) element strides"""
    import ctypes
    arr = ctypes.c_int64 * 4
    check(lib.meant_gemm_f32_strided(_p(A), _p(B), _p(C), M, N, K, 1, 1, arr(0, 0, *sA), arr(0, 0, *sB), arr(0, 0, *sC),
                                     float(alpha), int(accumulate), _stream()), "gemm_f32_strided")


class _ComposeLinear(torch.autograd.Function):
    """Two consecutive Linears with nothing between them -- the encoder's Linear(d,d) feeding the q/k/v Linears
    (meant/meant.py:60-61 -> meant/attention.py:36-37, meant/meant.py:102-103 -> meant/xPosAttention.py:37-38) --
    are one affine map:  (x W1^T + b1) Wqkv^T + bqkv = x (Wqkv W1)^T + (Wqkv b1 + bqkv).
    This function forms the composed weight and bias (fp32, a 3D x d x d product on the weights only) and, in
    backward, maps their gradients back to all four parameter tensors, so the per-token projection runs once
    instead of twice in forward and in both backward GEMMs.  Same function, same parameters, same gradients."""

    @staticmethod
    def forward(ctx, w1, b1, wqkv, bqkv):
        _need_gpu(w1, wqkv)
        w1c, wqc = _c(w1.detach().float()), _c(wqkv.detach().float())
        b1c, bqc = _c(b1.detach().float()), _c(bqkv.detach().float())
        N3, dj = wqc.shape
        dk = w1c.shape[1]
        wc = torch.empty((N3, dk), device=w1.device, dtype=torch.float32)
        _gemm_f32(wqc, w1c, wc, N3, dk, dj, (dj, 1), (dk, 1), (dk, 1))
        bc = bqc.clone()
        _gemm_f32(wqc, b1c, bc, N3, 1, dj, (dj, 1), (1, 0), (1, 1), accumulate=True)
        ctx.save_for_backward(w1c, b1c, wqc)
        return wc, bc

    @staticmethod
    def backward(ctx, dwc, dbc):
        w1c, b1c, wqc = ctx.saved_tensors
        N3, dj = wqc.shape
        dk = w1c.shape[1]
        dwc, dbc = _c(dwc.float()), _c(dbc.float())
        dwq = torch.empty_like(wqc)                                         # dWqkv = dWc W1^T + dbc (x) b1
        _gemm_f32(dwc, w1c, dwq, N3, dj, dk, (dk, 1), (1, dk), (dj, 1))
        _gemm_f32(dbc, b1c, dwq, N3, dj, 1, (1, 0), (0, 1), (dj, 1), accumulate=True)
        dw1 = torch.empty_like(w1c)                                         # dW1 = Wqkv^T dWc
        _gemm_f32(wqc, dwc, dw1, dj, dk, N3, (1, dj), (dk, 1), (dk, 1))
        db1 = torch.empty_like(b1c)                                         # db1 = Wqkv^T dbc
        _gemm_f32(wqc, dbc, db1, dj, 1, N3, (1, dj), (1, 0), (1, 1))
        return dw1, db1, dwq, dbc


def compose_linear(w1, b1, wqkv, bqkv):
    return _ComposeLinear.apply(w1, b1, wqkv, bqkv)


class _QKVAttention(torch.autograd.Function):
    """Fused q|k|v projection + rotary + attention core of meant/attention.py:35-57 and
    meant/xPosAttention.py:35-63 (up to, not including, multi_mad).

    wqkv [3D, d] / bqkv [3D] hold the projections in the order (query, KEY, VALUE) -- i.e. the reference's
    parameters (q, v, k): the k/v naming swap is resolved by the caller.  They may be the composed
    weight of _ComposeLinear.  tables = (qa, qb, ka, kb) float [S, R] or None."""

    @staticmethod
    def forward(ctx, x, wqkv, bqkv, tables, key_mask, causal, num_heads, scale=None):
        _need_gpu(x, wqkv)
        G, S, d = x.shape
        D = wqkv.shape[0] // 3
        Dh = D // num_heads
        x2 = _c(x).view(G * S, d)
        w_f = _c(wqkv.detach().float())
        w_c = cast(w_f, x.dtype)                                              # [3D, d] in the compute dtype
        bias_f = _c(bqkv.detach().float())
        dt = _dt(x)
        qa, qb, ka, kb = tables if tables is not None else (None, None, None, None)
        R = qa.shape[1] if qa is not None else 0
        qkv = torch.empty((G * S, 3 * D), device=x.device, dtype=x.dtype)
        check(lib.meant_qkv_proj_fwd(_p(x2), x2.stride(0), _p(w_c), _p(bias_f), _p(qkv), G * S, d, S, num_heads, Dh, R,
                                     _p(qa), _p(qb), _p(ka), _p(kb), dt, _stream()), "qkv_proj_fwd")
        o = torch.empty((G * S, D), device=x.device, dtype=x.dtype)
        lse = torch.empty((G, num_heads, S, 2), device=x.device, dtype=torch.float32)
        km = _c(key_mask.float()) if key_mask is not None else None
        scale = 1.0 / math.sqrt(Dh * num_heads) if scale is None else float(scale)
        wsb = lib.meant_attn_fwd_ws(G, S, num_heads, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=x.device, dtype=torch.uint8)
        check(lib.meant_attn_fwd(_p(qkv), _p(o), _p(lse), _p(km), G, S, num_heads, Dh, scale, int(causal), dt, _p(ws), wsb,
                                 _stream()), "attn_fwd")
        ctx.save_for_backward(x2, qkv, o, lse, km, w_f)
        ctx.tables = tables
        ctx.meta = (G, S, d, D, Dh, num_heads, scale, int(causal))
        return o.view(G, S, D)

    @staticmethod
    def backward(ctx, do):
        x2, qkv, o, lse, km, w_f = ctx.saved_tensors
        G, S, d, D, Dh, H, scale, causal = ctx.meta
        do2 = _c(do).view(G * S, D)
        dt = _dt(do2)
        dqkv = torch.empty_like(qkv)
        wsb = lib.meant_attn_ws(G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=do2.device, dtype=torch.uint8)
        qa, qb, ka, kb = ctx.tables if ctx.tables is not None else (None, None, None, None)
        R = qa.shape[1] if qa is not None else 0
        check(lib.meant_attn_bwd(_p(qkv), _p(o), _p(do2), _p(lse), _p(km), _p(dqkv), G, S, H, Dh, scale, causal,
                                 _p(qa), _p(qb), _p(ka), _p(kb), R, dt, _p(ws), wsb, _stream()), "attn_bwd")
        del ws
        M, N, K = G * S, 3 * D, d
        dx = None
        if ctx.needs_input_grad[0]:
            wT = torch.empty((K, N), device=do2.device, dtype=do2.dtype)
            check(lib.meant_transpose2d(_p(w_f), F32, _p(wT), dt, N, K, _stream()), "transpose2d")
            dx = torch.empty((M, K), device=do2.device, dtype=do2.dtype)
            check(lib.meant_linear_bwd_dx(_p(dqkv), N, _p(wT), _p(dx), K, M, N, K, dt, _stream()), "linear_bwd_dx")
        dw = torch.zeros((N, K), device=do2.device, dtype=torch.float32)
        db = torch.zeros(N, device=do2.device, dtype=torch.float32)
        _bwd_dw(dqkv, x2, dw, db)
        return (dx.view(G, S, d) if dx is not None else None), dw, db, None, None, None, None, None


NATIVE_HEAD_DIMS = (64, 96, 128)   # what the bf16 MFMA attention kernels take (csrc/attn_bf16.hip); 96 = the reference's default 8 heads


def qkv_attention(x, wq, bq, wk, bk, wv, bv, tables, key_mask, causal, num_heads, pre=None):
    """pre = (W1, b1) of a Linear applied to x immediately before the projections (composed into them).

    bf16 tier, head dims other than 64 / 96 / 128 (96 = the reference classes' default of 8 heads at d = 768, served
    natively): every head is widened to 128 columns by zero rows in the projection weights, so q, k, v come out of the GEMM already padded, the
    scores are unchanged (zeros add nothing to q.k, the scale stays 1/sqrt(dim)), and the zero columns of v give zero
    columns of the output, which are dropped again.  The padding is built from the parameters with differentiable ops,
    so their gradients need no special handling."""
    wqkv = torch.cat([wq, wk, wv], dim=0)
    bqkv = torch.cat([bq, bk, bv], dim=0)
    if pre is not None:
        wqkv, bqkv = compose_linear(pre[0], pre[1], wqkv, bqkv)
    D = wqkv.shape[0] // 3
    Dh = D // num_heads
    if x.dtype == torch.bfloat16 and Dh not in NATIVE_HEAD_DIMS and Dh < 128 and Dh % 8 == 0:
        Dp, d = 128, wqkv.shape[1]
        wp = torch.nn.functional.pad(wqkv.view(3 * num_heads, Dh, d), (0, 0, 0, Dp - Dh)).reshape(3 * num_heads * Dp, d)
        bp = torch.nn.functional.pad(bqkv.view(3 * num_heads, Dh), (0, Dp - Dh)).reshape(3 * num_heads * Dp)
        o = _QKVAttention.apply(x, wp, bp, tables, key_mask, causal, num_heads, 1.0 / math.sqrt(D))
        G, S = o.shape[0], o.shape[1]
        return o.view(G, S, num_heads, Dp)[..., :Dh].reshape(G, S, D)
    return _QKVAttention.apply(x, wqkv, bqkv, tables, key_mask, causal, num_heads)


# ---------------------------------------------------------------------------------------------
class _TemporalAttention(torch.autograd.Function):
    """meant/temporal.py:34-56 up to multi_mad: q from the last lag step, k/v from all L."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, num_heads):
        _need_gpu(x, wq)
        B, L, d = x.shape
        D = wq.shape[0]
        Dh = D // num_heads
        xc = _c(x)
        x2 = xc.view(B * L, d)
        xlast = xc[:, L - 1, :]                                                # [B, d] rows strided by L*d
        wq_c = weights.get((wq,), x.dtype, False)
        wkv_c = weights.get((wk, wv), x.dtype, False)
        q, _ = _linear_fwd_raw(xlast, wq_c, _c(bq.detach().float()), None, EPI_NONE, False)            # [B, D]
        kv, _ = _linear_fwd_raw(x2, wkv_c, torch.cat([bk.detach(), bv.detach()]).float().contiguous(), None, EPI_NONE, False)
        o = torch.empty((B, D), device=x.device, dtype=x.dtype)
        p = torch.empty((B, num_heads, L), device=x.device, dtype=torch.float32)
        scale = 1.0 / math.sqrt(Dh * num_heads)
        check(lib.meant_temporal_attn_fwd(_p(q), _p(kv), _p(o), _p(p), B, L, num_heads, Dh, scale, _dt(x), _stream()),
              "temporal_attn_fwd")
        ctx.save_for_backward(x2, q, kv, p)
        ctx.params = (wq, wk, wv)
        ctx.meta = (B, L, d, D, Dh, num_heads, scale)
        return o.view(B, 1, D)

    @staticmethod
    def backward(ctx, do):
        x2, q, kv, p = ctx.saved_tensors
        B, L, d, D, Dh, H, scale = ctx.meta
        wq, wk, wv = ctx.params
        do2 = _c(do).view(B, D)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        check(lib.meant_temporal_attn_bwd(_p(q), _p(kv), _p(p), _p(do2), _p(dq), _p(dkv), B, L, H, Dh, scale, _dt(do2), _stream()),
              "temporal_attn_bwd")
        xlast = x2.view(B, L, d)[:, L - 1, :]
        dxl, dwq, dbq = _linear_bwd_raw(dq, xlast, (wq,), True, True)          # [B, d]
        dx2, dwkv, dbkv = _linear_bwd_raw(dkv, x2, (wk, wv), True, True)        # [B*L, d]
        dx = dx2.view(B, L, d)
        last = dx[:, L - 1, :]
        tmp = torch.empty((B, d), device=dx.device, dtype=dx.dtype)
        lc = last.contiguous()
        check(lib.meant_add(_p(lc), _p(dxl), _p(tmp), B * d, _dt(dx), _stream()), "add")
        dx[:, L - 1, :] = tmp
        return dx, dwq, dbq, dwkv[:D], dbkv[:D], dwkv[D:], dbkv[D:], None


def temporal_attention(x, wq, bq, wk, bk, wv, bv, num_heads):
    return _TemporalAttention.apply(x, wq, bq, wk, bk, wv, bv, num_heads)


# ---------------------------------------------------------------------------------------------
class _MeanPoolCat(torch.autograd.Function):
    """torch.cat([mean_s(a), mean_n(b)], -1) of meant/meant.py:231 (b optional).  `out_dtype` is the tier of everything
    downstream (temporal encoder, head): the inputs' dtype, or fp32 when TAIL_FP32 is set."""

    @staticmethod
    def forward(ctx, a, b, out_dtype):
        _need_gpu(a, b)
        a = _c(a)
        G, S, da = a.shape
        db_ = 0
        if b is not None:
            b = _c(b)
            assert b.shape[0] == G and b.dtype == a.dtype
            db_ = b.shape[2]
        out = torch.empty((G, da + db_), device=a.device, dtype=out_dtype)
        dto = F32 if out_dtype == torch.float32 else BF16
        check(lib.meant_meanpool_fwd(_p(a), _p(out), da + db_, 0, G, S, da, _dt(a), dto, _stream()), "meanpool_fwd")
        if b is not None:
            check(lib.meant_meanpool_fwd(_p(b), _p(out), da + db_, da, G, b.shape[1], db_, _dt(a), dto, _stream()), "meanpool_fwd")
        ctx.meta = (a.shape, None if b is None else b.shape, a.dtype, out_dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        sa, sb, dtype, out_dtype = ctx.meta
        dout = _c(dout.to(out_dtype))
        ld = dout.shape[1]
        dti = F32 if dtype == torch.float32 else BF16
        dto = F32 if out_dtype == torch.float32 else BF16
        da = torch.empty(sa, device=dout.device, dtype=dtype)
        check(lib.meant_meanpool_bwd(_p(dout), ld, 0, _p(da), sa[0], sa[1], sa[2], dti, dto, _stream()), "meanpool_bwd")
        db = None
        if sb is not None:
            db = torch.empty(sb, device=dout.device, dtype=dtype)
            check(lib.meant_meanpool_bwd(_p(dout), ld, sa[2], _p(db), sb[0], sb[1], sb[2], dti, dto, _stream()), "meanpool_bwd")
        return da, db, None


# The temporal encoder and the head (0.06 % of the FLOPs, rows = B * lag) follow the tier of the encoders by default.  They used
# to stay fp32 in the bf16 tier: on the f32 MFMA their 1536^2 Linears cost 2.5 % of the step (nothing else can run beside them),
# for 1.4e-3 instead of 1.9e-3 max deviation of the output from the golden at full dims (gate 1e-2; the reference runs them in
# fp16 under autocast).  MEANT_TAIL_FP32=1 (or ops.TAIL_FP32 = True) restores the fp32 tail.
TAIL_FP32 = os.environ.get("MEANT_TAIL_FP32") == "1"


def meanpool_cat(a, b=None):
    return _MeanPoolCat.apply(a, b, torch.float32 if TAIL_FP32 else a.dtype)


def meanpool_f32(a):
    """mean over the sequence axis of [G, S, d] -> float [G, d]"""
    return _MeanPoolCat.apply(a, None, torch.float32)


class _PoolLinearCat(torch.autograd.Function):
    """torch.cat([mean_s(h_a W_a^T + b_a + x_a), mean_n(h_b W_b^T + b_b + x_b)], -1): the LAST Linear (+ residual) of the last
    encoder layer of each stack followed by the sequence mean-pool (meant/meant.py:74 / :120 -> :231), evaluated as
    mean_s(h) W^T + b + mean_s(x) -- a per-token affine map commutes with the mean over tokens.  Same function and
    gradients; the [tokens, d] x [d, d] GEMM, its input-gradient GEMM and its weight-gradient GEMM (3 of the ~16 big GEMMs
    per stack at one encoder layer) become [groups, d] products, and their [tokens, d] gradients are row broadcasts.
    Parts: (h [G, S, d], x [G, S, d], W [d, d], b [d]); the second part is optional.  The pooled arithmetic is fp32 in
    both tiers (fp32 means, exact-f32 MFMA products on the master weights: [groups, d] is tiny), rounded once to
    `out_dtype`, the dtype of everything after the pooling (see TAIL_FP32)."""

    @staticmethod
    def forward(ctx, out_dtype, *flat):
        parts = [flat[i:i + 4] for i in range(0, len(flat), 4)]
        _need_gpu(*[t for p in parts for t in p])
        G = parts[0][0].shape[0]
        widths = [p[2].shape[0] for p in parts]
        Dt = sum(widths)
        dev = parts[0][0].device
        out = torch.empty((G, Dt), device=dev, dtype=torch.float32)
        xm = torch.empty((G, Dt), device=dev, dtype=torch.float32)    # mean_s(x_p), laid out like `out` (same row stride)
        saved, off = [], 0
        for (h, x, W, b), N in zip(parts, widths):
            h, x = _c(h), _c(x)
            assert h.shape[0] == G and x.shape == h.shape[:2] + (N,) and h.dtype == x.dtype
            S, K = h.shape[1], h.shape[2]
            hm = torch.empty((G, K), device=dev, dtype=torch.float32)
            check(lib.meant_meanpool_fwd(_p(h), _p(hm), K, 0, G, S, K, _dt(h), F32, _stream()), "meanpool_fwd")
            check(lib.meant_meanpool_fwd(_p(x), _p(xm), Dt, off, G, S, N, _dt(x), F32, _stream()), "meanpool_fwd")
            w_f = _c(W.detach().float())
            bias_f = _c(b.detach().float()) if b is not None else None
            check(lib.meant_linear_fwd(_p(hm), K, _p(w_f), _p(bias_f), xm.data_ptr() + off * 4, Dt, out.data_ptr() + off * 4, Dt, None,
                                       G, N, K, EPI_RESIDUAL, F32, _stream()), "linear_fwd")
            saved.append((hm, W, b is not None, h.shape, h.dtype, off))
            off += N
        ctx.parts = saved
        return cast(out, out_dtype)

    @staticmethod
    def backward(ctx, dout):
        dout = cast(_c(dout), torch.float32)
        G, Dt = dout.shape
        grads = [None]
        for hm, W, has_b, hshape, hdt, off in ctx.parts:
            N, K = W.shape
            S = hshape[1]
            dy_ptr = dout.data_ptr() + off * 4                              # [G, N] slice of dout, row stride Dt
            dti = F32 if hdt == torch.float32 else BF16
            wT = weights.get((W,), torch.float32, True)                    # [K, N]
            dhm = torch.empty((G, K), device=dout.device, dtype=torch.float32)
            check(lib.meant_linear_bwd_dx(dy_ptr, Dt, _p(wT), _p(dhm), K, G, N, K, F32, _stream()), "linear_bwd_dx")
            dw = torch.zeros((N, K), device=dout.device, dtype=torch.float32)
            db = torch.zeros(N, device=dout.device, dtype=torch.float32) if has_b else None
            check(lib.meant_linear_bwd_dw(dy_ptr, Dt, _p(hm), K, _p(dw), _p(db), G, N, K, F32, None, 0, _stream()), "linear_bwd_dw")
            dh = torch.empty(hshape, device=dout.device, dtype=hdt)         # every token of a group gets d(mean h) / S
            check(lib.meant_meanpool_bwd(_p(dhm), K, 0, _p(dh), G, S, K, dti, F32, _stream()), "meanpool_bwd")
            dx = torch.empty(hshape[:2] + (N,), device=dout.device, dtype=hdt)
            check(lib.meant_meanpool_bwd(_p(dout), Dt, off, _p(dx), G, S, N, dti, F32, _stream()), "meanpool_bwd")
            grads += [dh, dx, dw, db]
        return tuple(grads)


class _PooledLinearCat(torch.autograd.Function):
    """as _PoolLinearCat, on means that the norm kernels already produced: parts (hm [G, d] float, xm [G, d] float, W, b)"""

    @staticmethod
    def forward(ctx, out_dtype, *flat):
        parts = [flat[i:i + 4] for i in range(0, len(flat), 4)]
        _need_gpu(*[t for p in parts for t in p])
        G = parts[0][0].shape[0]
        widths = [p[2].shape[0] for p in parts]
        Dt = sum(widths)
        dev = parts[0][0].device
        out = torch.empty((G, Dt), device=dev, dtype=torch.float32)
        # the generic fp32 GEMM wants residual stride == output stride: the means of x side by side, like `out`
        xs = _c(parts[0][1]) if len(parts) == 1 else torch.cat([p[1] for p in parts], dim=1)
        saved, off = [], 0
        for (hm, xm, W, b), N in zip(parts, widths):
            hm = _c(hm)
            K = hm.shape[1]
            w_f = _c(W.detach().float())
            bias_f = _c(b.detach().float()) if b is not None else None
            check(lib.meant_linear_fwd(_p(hm), K, _p(w_f), _p(bias_f), xs.data_ptr() + off * 4, Dt, out.data_ptr() + off * 4, Dt, None,
                                       G, N, K, EPI_RESIDUAL, F32, _stream()), "linear_fwd")
            saved.append((hm, W, b is not None, off))
            off += N
        ctx.parts = saved
        return cast(out, out_dtype)

    @staticmethod
    def backward(ctx, dout):
        dout = cast(_c(dout), torch.float32)
        G, Dt = dout.shape
        grads = [None]
        for hm, W, has_b, off in ctx.parts:
            N, K = W.shape
            dy_ptr = dout.data_ptr() + off * 4
            wT = weights.get((W,), torch.float32, True)
            dhm = torch.empty((G, K), device=dout.device, dtype=torch.float32)
            check(lib.meant_linear_bwd_dx(dy_ptr, Dt, _p(wT), _p(dhm), K, G, N, K, F32, _stream()), "linear_bwd_dx")
            dw = torch.zeros((N, K), device=dout.device, dtype=torch.float32)
            db = torch.zeros(N, device=dout.device, dtype=torch.float32) if has_b else None
            check(lib.meant_linear_bwd_dw(dy_ptr, Dt, _p(hm), K, _p(dw), _p(db), G, N, K, F32, None, 0, _stream()), "linear_bwd_dw")
            dxm = dout[:, off:off + N] if Dt == N else dout[:, off:off + N].contiguous()
            grads += [dhm, dxm, dw, db]
        return tuple(grads)


def pool_linear_cat(parts):
    """parts: [(h, x, weight, bias), ...] (one or two): cat_p(mean_tokens(h_p W_p^T + b_p + x_p)) -> [G, sum d_p]"""
    flat = [t for p in parts for t in p]
    out_dtype = torch.float32 if TAIL_FP32 else parts[0][0].dtype
    return _PoolLinearCat.apply(out_dtype, *flat)


def pooled_linear_cat(parts, compute_dtype):
    """parts: [(mean_s(h) [G, d] float, mean_s(x) [G, d] float, weight, bias), ...] -> cat_p(hm_p W_p^T + b_p + xm_p) in the dtype
    of everything after the pooling"""
    out_dtype = torch.float32 if TAIL_FP32 else compute_dtype
    return _PooledLinearCat.apply(out_dtype, *[t for p in parts for t in p])


class _AddRowVec(torch.autograd.Function):
    """x[b, l, :] + v[0, l, :]  (temp_embedding, meant/meant.py:141-142)."""

    @staticmethod
    def forward(ctx, x, v):
        _need_gpu(x, v)
        x = _c(x)
        B, L, d = x.shape
        vf = _c(v.detach().float()).view(L, d)
        y = torch.empty_like(x)
        check(lib.meant_add_rowvec(_p(x), _p(vf), _p(y), B * L, d, L, _dt(x), _stream()), "add_rowvec")
        ctx.meta = (B, L, d, v.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, L, d, vshape = ctx.meta
        dy = _c(dy)
        dv = torch.empty((L, d), device=dy.device, dtype=torch.float32)
        check(lib.meant_add_rowvec_bwd(_p(dy), _p(dv), B * L, d, L, _dt(dy), _stream()), "add_rowvec_bwd")
        return dy, dv.view(vshape)


def add_rowvec(x, v):
    return _AddRowVec.apply(x, v)


_RAW_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2, torch.uint8: 3}


def patchify(images, p: int, dtype: torch.dtype, mean: float = 0.0, std: float = 1.0):
    """einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (meant/meant.py:194); no gradient to pixels.

    `images` may be in any storage type the reference's data path produces (float64 .npy graphs,
    in_loop_train.py:48,589; uint8 renderings; float32; bf16): the conversion, the optional global
    normalisation (x - mean) / std of in_loop_train.py:591-593 and the patch layout are one pass on the device."""
    _need_gpu(images)
    if images.dtype not in _RAW_DTYPES:
        images = images.float()
    images = _c(images)
    G, Cc, Hh, Ww = images.shape
    n = (Hh // p) * (Ww // p)
    out = torch.empty((G, n, p * p * Cc), device=images.device, dtype=dtype)
    odt = F32 if dtype == torch.float32 else BF16
    if images.dtype in (torch.float32, torch.bfloat16) and mean == 0.0 and std == 1.0:
        check(lib.meant_patchify(_p(images), _dt(images), _p(out), G, Cc, Hh, Ww, p, odt, _stream()), "patchify")
    else:
        check(lib.meant_patchify_raw(_p(images), _RAW_DTYPES[images.dtype], float(mean), 1.0 / float(std), _p(out), G, Cc, Hh, Ww, p,
                                     odt, _stream()), "patchify_raw")
    return out


EMB_BF16_TABLE = os.environ.get("MEANT_EMB_BF16_TABLE", "1") == "1"     # 0: the bf16 tier's lookup reads the fp32 table (A/B measurements)


EMB_PRESORT = os.environ.get("MEANT_EMB_PRESORT", "1") == "1"          # 0: sort the ids in backward (A/B measurements)
_aux_streams = {}
_aux_override = {}                                  # (device type, index) -> a stream the model already owns (set_index_stream)


def set_index_stream(device, stream) -> None:
    """let small index work (the id sort of the embedding backward) ride a stream the model runs anyway -- `meant` hands over its
    vision stream -- instead of one more stream of its own: same step time, and one stream fewer towards the cliff described below"""
    _aux_override[(device.type, device.index)] = stream


def _aux_stream(device):
    """one extra HIP stream per device for small index work that should not sit on a compute stream's critical path.
    (ONE: a fifth stream next to main / vision / language / this one -- tried for the weight compositions, which depend on
    parameters only -- cost the step 11 %, 2800 against 3140 samples/s, whether or not anything ran on it)"""
    key = (device.type, device.index)
    if _aux_override.get(key) is not None:
        return _aux_override[key]
    st = _aux_streams.get(key)
    if st is None:
        st = _aux_streams[key] = torch.cuda.Stream(device=device)
    return st


def _emb_sorted_bwd_ok(n: int, d: int) -> bool:
    """shapes whose embedding gradient is summed in sorted-id order (meant_embedding_bwd_sorted)"""
    return d <= 1024 and d % 8 == 0 and (n >= 4096 or bool(_lib_option("deterministic")))


class _Embedding(torch.autograd.Function):
    """nn.Embedding lookup (meant/meant.py:211) emitting the compute dtype directly."""

    @staticmethod
    def forward(ctx, ids, table, dtype):
        _need_gpu(ids, table)
        ids_c = _c(ids.long())
        V, d = table.shape
        n = ids_c.numel()
        out = torch.empty((*ids.shape, d), device=table.device, dtype=dtype)
        if dtype == torch.bfloat16 and EMB_BF16_TABLE and d % 8 == 0 and V < (1 << 31):
            # bf16 tier: rows copied from a bf16 image of the table (the weight cache's: rebuilt when the parameter changes, shared
            # with a tied vocabulary decoder) instead of read as fp32 and rounded on the way -- same values, half the gather's reads
            # (0.51 -> 0.37 ms at 786 k tokens of 768).  Ids are clamped into the table as meant_embedding_fwd clamps them.
            tb = weights.get((table,), torch.bfloat16, False)
            idx = ids_c.view(-1).clamp(0, V - 1).to(torch.int32)
            check(lib.meant_gather_rows(_p(tb), _p(idx), None, _p(out), n, d, BF16, _stream()), "gather_rows")
        else:
            tf = _c(table.detach().float())
            check(lib.meant_embedding_fwd(_p(tf), _p(ids_c), _p(out), n, d, V, F32 if dtype == torch.float32 else BF16, _stream()),
                  "embedding_fwd")
        # The backward sums rows in the order of the sorted ids.  The sort (a radix block sort + ~20 merge launches, 0.2 ms of
        # kernels that do not fill the chip) used to run in backward, where the embedding's gradient is the very last thing
        # of the step; the ids are known now, so it runs here on a side stream, beside the forward's big kernels.
        presort = None
        if EMB_PRESORT and ctx.needs_input_grad[1] and _emb_sorted_bwd_ok(n, d):
            main = torch.cuda.current_stream(ids_c.device)
            side = _aux_stream(ids_c.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sorted_ids, order = torch.sort(ids_c.view(-1))
                ev = torch.cuda.Event()
                ev.record(side)
            presort = (sorted_ids, order, ev)
        ctx.save_for_backward(ids_c)
        ctx.presort = presort
        ctx.meta = (V, d)
        ctx.table = table                                # the Parameter itself: its gradient sink, if any, is looked up in backward
        _claim(table, "embedding")
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids_c,) = ctx.saved_tensors
        V, d = ctx.meta
        dout = _c(dout)
        # with a gradient sink (GradReducer(direct_grads=True)) the rows are summed straight into the reducer's bucket view:
        # no [V, d] zeros + autograd add per step (0.8 GB of traffic at V = 64001)
        # (not when another op reads the table as well -- the tied vocabulary decoder: see grad_sinks)
        sink = _sink_of(ctx.table, "embedding") if grad_sinks else None
        if sink is not None and (sink.view.shape != (V, d) or not sink.view.is_contiguous()):
            sink = None
        dtab = sink.view if sink is not None else torch.zeros((V, d), device=dout.device, dtype=torch.float32)
        n = ids_c.numel()
        if _emb_sorted_bwd_ok(n, d):                     # the other kernel is float atomics per token
            # index preparation (a sort of the token ids) is host-side plumbing; the reduction itself is the HIP kernel
            if ctx.presort is not None:
                sorted_ids, order, ev = ctx.presort
                cur = torch.cuda.current_stream(dout.device)
                cur.wait_event(ev)
                sorted_ids.record_stream(cur)            # allocated on the side stream, read on this one
                order.record_stream(cur)
            else:
                sorted_ids, order = torch.sort(ids_c.view(-1))
            nsl = sink.row_slices(ctx.table) if sink is not None else 1
            if nsl > 1:
                # data parallel: the table's gradient is the last thing backward produces and two thirds of the bytes to reduce.
                # It is summed in row slices; each slice's all-reduce starts while the next slice is being summed
                for c in range(nsl):
                    lo, hi = V * c // nsl, V * (c + 1) // nsl
                    check(lib.meant_embedding_bwd_sorted_range(_p(dout), _p(sorted_ids), _p(order), _p(dtab), n, d, V, lo, hi, _dt(dout),
                                                               _stream()), "embedding_bwd_sorted_range")
                    sink.report_rows(ctx.table, lo, hi, c == nsl - 1)
                return None, None, None
            check(lib.meant_embedding_bwd_sorted(_p(dout), _p(sorted_ids), _p(order), _p(dtab), n, d, V, _dt(dout), _stream()),
                  "embedding_bwd_sorted")
        else:
            check(lib.meant_embedding_bwd(_p(dout), _p(ids_c), _p(dtab), n, d, V, _dt(dout), _stream()), "embedding_bwd")
        if sink is not None:
            sink.report(ctx.table)
            return None, None, None
        return None, dtab, None


def embedding(ids, table, dtype):
    return _Embedding.apply(ids, table, dtype)


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """dtype conversion of an activation through the library's cast kernel (no autograd)."""
    _need_gpu(x)
    if x.dtype == dtype:
        return x
    x = _c(x)
    out = torch.empty_like(x, dtype=dtype)
    check(lib.meant_cast(_p(x), _dt(x), _p(out), F32 if dtype == torch.float32 else BF16, x.numel(), _stream()), "cast")
    return out


# ---------------------------------------------------------------------------------------------
# MLM pretrainer (pretrain_mlm.py:74-88, :160, :178)
VOCAB_PAD = 256        # the vocabulary GEMM runs on N rounded up to the 256-wide tile of the streaming kernel


class _SoftmaxCE(torch.autograd.Function):
    """nn.CrossEntropyLoss()(logits[:, :V], target) with ignore_index (default -100), mean over the rows that count,
    on a row-padded logits buffer [T, ld >= V]: one pass over the logits forward, one backward, no [T, V] fp32
    log-softmax tensor.  The gradient is returned for the whole padded buffer (zeros in the padding)."""

    @staticmethod
    def forward(ctx, logits_pad, target, V, ignore_index):
        _need_gpu(logits_pad, target)
        logits_pad = _c(logits_pad)
        T, ld = logits_pad.shape
        assert ld % 8 == 0 and V <= ld
        tgt = _c(target.long())
        row_loss = torch.empty(T, device=logits_pad.device, dtype=torch.float32)
        lse = torch.empty(T, device=logits_pad.device, dtype=torch.float32)
        check(lib.meant_softmax_ce_fwd(_p(logits_pad), ld, _p(tgt), T, V, int(ignore_index), _p(row_loss), _p(lse), _dt(logits_pad),
                                       _stream()), "softmax_ce_fwd")
        nvalid = ((tgt != ignore_index) & (tgt >= 0) & (tgt < V)).sum().clamp_(min=1).float()
        ctx.save_for_backward(logits_pad, tgt, lse, nvalid)
        ctx.V, ctx.ignore_index = int(V), int(ignore_index)
        return row_loss.sum() / nvalid

    @staticmethod
    def backward(ctx, dloss):
        logits_pad, tgt, lse, nvalid = ctx.saved_tensors
        T, ld = logits_pad.shape
        g = (dloss.float() / nvalid).reshape(1).contiguous()
        dl = torch.empty_like(logits_pad)
        check(lib.meant_softmax_ce_bwd(_p(logits_pad), ld, _p(tgt), _p(lse), T, ctx.V, ctx.ignore_index, _p(g), _p(dl), _dt(logits_pad),
                                       _stream()), "softmax_ce_bwd")
        return dl, None, None, None


class _VocabLinear(torch.autograd.Function):
    """x W^T + b on W [V, d] with the output rows padded to Vp = ceil(V / 256) * 256 columns.  The padded bf16 copy of W
    (forward) and its transpose (input gradient) come from the weight cache keyed on the real parameter, so they are
    rebuilt once per optimizer step, not per call; the gradient is produced for the Vp rows and the V real ones handed
    back as a view (no pad / un-pad passes over the 196 MB matrix)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, weight)
        V, d = weight.shape
        Vp = (V + VOCAB_PAD - 1) // VOCAB_PAD * VOCAB_PAD
        shp = x.shape
        x2 = _c(x).view(-1, d)
        w_c = weights.get((weight,), x.dtype, False, Vp - V)
        bias_f = None
        if bias is not None:
            bias_f = _c(torch.nn.functional.pad(bias.detach().float(), (0, Vp - V)))
        y, _ = _linear_fwd_raw(x2, w_c, bias_f, None, EPI_NONE, False)
        ctx.weight, ctx.bias, ctx.has_bias, ctx.in_shape, ctx.V = weight, bias, bias is not None, shp, V
        _claim(weight, "vocab"); _claim(bias, "vocab")
        ctx.save_for_backward(x2)
        return y.view(*shp[:-1], Vp)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        V = ctx.V
        Vp = dy.shape[-1]
        dy2 = _c(dy).view(-1, Vp)
        # V a multiple of 256: nothing is padded and the gradients may go straight into the parameters' sinks (dw, db None)
        dx, dw, db = _linear_bwd_raw(dy2, x2, (ctx.weight,), ctx.needs_input_grad[0], ctx.has_bias, pad_rows=Vp - V,
                                     bias_param=ctx.bias, site="vocab")
        return ((dx.view(ctx.in_shape) if dx is not None else None), (dw[:V] if dw is not None else None),
                (db[:V] if db is not None else None))


def _vocab_logits_padded(x, weight, bias):
    return _VocabLinear.apply(x, weight, bias)                           # [..., Vp]


def vocab_linear(x, weight, bias=None):
    """logits = x W^T + b for a big (tied) vocabulary matrix W [V, d] (pretrain_mlm.py:88 -> RobertaLMHead.decoder).
    Returns a [..., V] VIEW of a buffer whose rows are padded to a multiple of 256 columns, so that the GEMM takes
    the streaming 256 x 256 kernel (padding rows of W are zeros; gradients come back for exactly the V real rows)."""
    return _vocab_logits_padded(x, weight, bias)[..., :weight.shape[0]]


def vocab_linear_cross_entropy(x, weight, bias, target, ignore_index: int = -100):
    """CrossEntropyLoss(vocab_linear(x).view(-1, V), target.view(-1)) without ever slicing or re-homing the logits:
    the padded logits buffer goes straight into the fused loss and its gradient straight back into the GEMMs."""
    lp = _vocab_logits_padded(x, weight, bias)
    return _SoftmaxCE.apply(lp.reshape(-1, lp.shape[-1]), target.reshape(-1), weight.shape[0], ignore_index)


def softmax_cross_entropy(logits, target, ignore_index: int = -100):
    """mean CE over the last axis of `logits` ([..., V], any leading shape, any strides) against integer targets.
    Rows are re-homed once into a buffer with an 8-element-aligned stride when they are not already laid out so."""
    V = logits.shape[-1]
    l2 = logits.reshape(-1, V)
    Vp = (V + 7) // 8 * 8
    if Vp != V or not l2.is_contiguous():
        l2 = torch.nn.functional.pad(l2, (0, Vp - V))
    return _SoftmaxCE.apply(l2, target.reshape(-1), V, ignore_index)


# ---------------------------------------------------------------------------------------------
# attention core on an already projected, packed q|k|v buffer (the divided space-time attention of
# src/meant/timesformer_pytorch.py regroups tokens between the projection and the core)
class _AttentionCore(torch.autograd.Function):
    """softmax(q k^T * scale [+ key padding]) v per group of S rows, on packed [G*S, 3*H*Dh] (q | k | v).  `tables` =
    (qa, qb, ka, kb) float [S, R] rotate q and k first (meant_rotary_qk); the backward applies the adjoint inside the
    attention backward kernel and returns the gradient w.r.t. the UNROTATED buffer."""

    @staticmethod
    def forward(ctx, qkv, tables, G, S, H, scale, causal, key_mask):
        _need_gpu(qkv)
        dt = _dt(qkv)
        D3 = qkv.shape[-1]
        D = D3 // 3
        Dh = D // H
        qa, qb, ka, kb = tables if tables is not None else (None, None, None, None)
        R = qa.shape[1] if qa is not None else 0
        q2 = qkv.reshape(G * S, D3)
        if qa is not None:
            q2 = q2.clone() if q2.data_ptr() == qkv.data_ptr() else q2
            check(lib.meant_rotary_qk(_p(q2), G * S, S, H, Dh, R, _p(qa), _p(qb), _p(ka), _p(kb), 0, dt, _stream()), "rotary_qk")
        else:
            q2 = _c(q2)
        o = torch.empty((G * S, D), device=qkv.device, dtype=qkv.dtype)
        lse = torch.empty((G, H, S, 2), device=qkv.device, dtype=torch.float32)
        km = _c(key_mask.float()) if key_mask is not None else None
        wsb = lib.meant_attn_fwd_ws(G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=qkv.device, dtype=torch.uint8)
        check(lib.meant_attn_fwd(_p(q2), _p(o), _p(lse), _p(km), G, S, H, Dh, float(scale), int(causal), dt, _p(ws), wsb, _stream()), "attn_fwd")
        ctx.save_for_backward(q2, o, lse, km)
        ctx.tables, ctx.meta, ctx.in_shape = tables, (G, S, H, Dh, float(scale), int(causal)), qkv.shape
        return o

    @staticmethod
    def backward(ctx, do):
        q2, o, lse, km = ctx.saved_tensors
        G, S, H, Dh, scale, causal = ctx.meta
        do2 = _c(do)
        dt = _dt(do2)
        dqkv = torch.empty_like(q2)
        wsb = lib.meant_attn_ws(G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=do2.device, dtype=torch.uint8)
        qa, qb, ka, kb = ctx.tables if ctx.tables is not None else (None, None, None, None)
        R = qa.shape[1] if qa is not None else 0
        check(lib.meant_attn_bwd(_p(q2), _p(o), _p(do2), _p(lse), _p(km), _p(dqkv), G, S, H, Dh, scale, causal,
                                 _p(qa), _p(qb), _p(ka), _p(kb), R, dt, _p(ws), wsb, _stream()), "attn_bwd")
        return dqkv.view(ctx.in_shape), None, None, None, None, None, None, None


class _AttentionCoreScoreDropout(torch.autograd.Function):
    """_AttentionCore with dropout on the score matrix (meant/xPosAttention.py:59: after the causal fill and the padding term, before
    the softmax).  The fused kernels never see a score matrix; this runs the materialised fp32 core (meant_attn_drop_fwd / _bwd): rotary
    by meant_rotary_qk in front, its adjoint behind the backward.  No reference model sets the probability above zero."""

    @staticmethod
    def forward(ctx, qkv, tables, G, S, H, scale, causal, key_mask, drop_p, seed):
        _need_gpu(qkv)
        dt = _dt(qkv)
        D3 = qkv.shape[-1]
        D = D3 // 3
        Dh = D // H
        qa, qb, ka, kb = tables if tables is not None else (None, None, None, None)
        R = qa.shape[1] if qa is not None else 0
        q2 = qkv.reshape(G * S, D3)
        if qa is not None:
            q2 = q2.clone() if q2.data_ptr() == qkv.data_ptr() else q2
            check(lib.meant_rotary_qk(_p(q2), G * S, S, H, Dh, R, _p(qa), _p(qb), _p(ka), _p(kb), 0, dt, _stream()), "rotary_qk")
        else:
            q2 = _c(q2)
        o = torch.empty((G * S, D), device=qkv.device, dtype=qkv.dtype)
        lse = torch.empty((G, H, S, 2), device=qkv.device, dtype=torch.float32)
        km = _c(key_mask.float()) if key_mask is not None else None
        wsb = lib.meant_attn_drop_ws(G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=qkv.device, dtype=torch.uint8)
        check(lib.meant_attn_drop_fwd(_p(q2), _p(o), _p(lse), _p(km), G, S, H, Dh, float(scale), int(causal), float(drop_p), int(seed), dt,
                                      _p(ws), wsb, _stream()), "attn_drop_fwd")
        ctx.save_for_backward(q2, o, lse, km)
        ctx.tables, ctx.meta, ctx.in_shape = tables, (G, S, H, Dh, float(scale), int(causal), float(drop_p), int(seed)), qkv.shape
        return o

    @staticmethod
    def backward(ctx, do):
        q2, o, lse, km = ctx.saved_tensors
        G, S, H, Dh, scale, causal, drop_p, seed = ctx.meta
        do2 = _c(do)
        dt = _dt(do2)
        dqkv = torch.empty_like(q2)
        wsb = lib.meant_attn_drop_ws(G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=do2.device, dtype=torch.uint8)
        check(lib.meant_attn_drop_bwd(_p(q2), _p(o), _p(do2), _p(lse), _p(km), _p(dqkv), G, S, H, Dh, scale, causal, drop_p, seed, dt,
                                      _p(ws), wsb, _stream()), "attn_drop_bwd")
        if ctx.tables is not None:
            qa, qb, ka, kb = ctx.tables
            check(lib.meant_rotary_qk(_p(dqkv), G * S, S, H, Dh, qa.shape[1], _p(qa), _p(qb), _p(ka), _p(kb), 1, dt, _stream()), "rotary_qk^T")
        return dqkv.view(ctx.in_shape), None, None, None, None, None, None, None, None, None


def attention_core_score_dropout(qkv, G, S, H, scale, drop_p, seed, tables=None, causal=False, key_mask=None):
    """attention_core with dropout(drop_p) on the score matrix; `seed` selects the mask (forward and backward share it)"""
    return _AttentionCoreScoreDropout.apply(qkv, tables, int(G), int(S), int(H), float(scale), bool(causal), key_mask, float(drop_p), int(seed))


def attention_core(qkv, G, S, H, scale, tables=None, causal=False, key_mask=None):
    """qkv: [G*S, 3*H*Dh] (or any shape with that many elements per row) -> o [G*S, H*Dh]"""
    return _AttentionCore.apply(qkv, tables, int(G), int(S), int(H), float(scale), bool(causal), key_mask)


class _GEGLU(torch.autograd.Function):
    """a * gelu(g) on h = [a | g] (src/meant/timesformer_pytorch.py:60-63), one pass forward and one backward"""

    @staticmethod
    def forward(ctx, h):
        _need_gpu(h)
        h2 = _c(h).view(-1, h.shape[-1])
        rows, w2 = h2.shape
        y = torch.empty((rows, w2 // 2), device=h.device, dtype=h.dtype)
        check(lib.meant_geglu_fwd(_p(h2), _p(y), rows, w2 // 2, _dt(h2), _stream()), "geglu_fwd")
        ctx.save_for_backward(h2)
        ctx.shape = h.shape
        return y.view(*h.shape[:-1], w2 // 2)

    @staticmethod
    def backward(ctx, dy):
        (h2,) = ctx.saved_tensors
        rows, w2 = h2.shape
        d2 = _c(dy).view(rows, w2 // 2)
        dh = torch.empty_like(h2)
        check(lib.meant_geglu_bwd(_p(h2), _p(d2), _p(dh), rows, w2 // 2, _dt(h2), _stream()), "geglu_bwd")
        return dh.view(ctx.shape)


def geglu(h):
    return _GEGLU.apply(h)


def gather_rows(src2d, idx, fill=None):
    """dst[r] = src2d[idx[r]] (idx int32 on the device; -1 -> zeros, -2 -> `fill` row); no autograd"""
    _need_gpu(src2d, idx)
    src2d = _c(src2d)
    n, W = idx.numel(), src2d.shape[1]
    dst = torch.empty((n, W), device=src2d.device, dtype=src2d.dtype)
    check(lib.meant_gather_rows(_p(src2d), _p(idx), _p(fill), _p(dst), n, W, _dt(src2d), _stream()), "gather_rows")
    return dst


class _DividedAttention(torch.autograd.Function):
    """One half of the divided space-time pair between the q|k|v projection and the output projection
    (src/meant/timesformer_pytorch.py:108-145), entirely on the HIP path:
      forward : rows regrouped by an index table (meant_gather_rows; the cls row in front of every group) -> rotary in place
                (identity row for the cls position) -> flash attention core -> rows back to token order by the inverse
                table -> the cls query's attention over all tokens written into row 0 (meant_attn_cls_fwd);
      backward: the same tables in reverse, the attention backward with the rotary adjoint inside, meant_group_scatter
                (copy of the group rows, sum of the cls rows), and the cls backward ADDING into that buffer.
    qkv [b, L, 3*H*Dh] -> out [b, L, H*Dh].  plan = (index [G, S] int32, idx_in [b*G*S], idx_out [b*L], idx_dog [b*G*S])."""

    @staticmethod
    def forward(ctx, qkv, plan, tables, H, scale, group_mask, cls_mask):
        _need_gpu(qkv)
        qkv = _c(qkv)
        b, L, D3 = qkv.shape
        D = D3 // 3
        Dh = D // H
        index, idx_in, idx_out, idx_dog = plan
        G, S = index.shape
        dt = _dt(qkv)
        q2 = qkv.view(b * L, D3)
        if tables is not None:                                                  # regrouped and rotated in one pass: [b G S, 3 D]
            qa, qb, ka, kb = tables
            grouped = torch.empty((b * G * S, D3), device=qkv.device, dtype=qkv.dtype)
            check(lib.meant_gather_rows_rot(_p(q2), _p(idx_in), _p(grouped), b * G * S, S, H, Dh, qa.shape[1], _p(qa), _p(qb), _p(ka), _p(kb),
                                            dt, _stream()), "gather_rows_rot")
        else:                                                                   # rotary_emb=False: learned positions were added to x instead
            grouped = gather_rows(q2, idx_in)
        og = torch.empty((b * G * S, D), device=qkv.device, dtype=qkv.dtype)
        lse = torch.empty((b * G, H, S, 2), device=qkv.device, dtype=torch.float32)
        km = _c(group_mask.float()) if group_mask is not None else None         # [b G, S]
        wsb = lib.meant_attn_fwd_ws(b * G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=qkv.device, dtype=torch.uint8)
        check(lib.meant_attn_fwd(_p(grouped), _p(og), _p(lse), _p(km), b * G, S, H, Dh, float(scale), 0, dt, _p(ws), wsb, _stream()), "attn_fwd")
        out = gather_rows(og, idx_out).view(b, L, D)                             # row 0 of every sequence: zeros for now
        stats = torch.empty((b, H, 2), device=qkv.device, dtype=torch.float32)
        cm = _c(cls_mask.float()) if cls_mask is not None else None             # [b, L]
        check(lib.meant_attn_cls_fwd(_p(q2), _p(out), L * D, _p(stats), _p(cm), b, L, H, Dh, float(scale), dt, _stream()), "attn_cls_fwd")
        ctx.save_for_backward(qkv, grouped, og, lse, km, out, stats, cm)
        ctx.plan, ctx.tables, ctx.meta = plan, tables, (b, L, D, H, Dh, G, S, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, grouped, og, lse, km, out, stats, cm = ctx.saved_tensors
        b, L, D, H, Dh, G, S, scale = ctx.meta
        index, idx_in, idx_out, idx_dog = ctx.plan
        dout = _c(dout)
        dt = _dt(dout)
        dog = gather_rows(dout.view(b * L, D), idx_dog)                          # position 0 of every group: zeros (its output was dropped)
        dgrouped = torch.empty_like(grouped)
        wsb = lib.meant_attn_ws(b * G, S, H, Dh, dt)
        ws = torch.empty(max(wsb, 16), device=dout.device, dtype=torch.uint8)
        qa, qb, ka, kb = ctx.tables if ctx.tables is not None else (None, None, None, None)
        check(lib.meant_attn_bwd(_p(grouped), _p(og), _p(dog), _p(lse), _p(km), _p(dgrouped), b * G, S, H, Dh, scale, 0,
                                 _p(qa), _p(qb), _p(ka), _p(kb), qa.shape[1] if qa is not None else 0, dt, _p(ws), wsb, _stream()), "attn_bwd")
        dqkv = torch.empty_like(qkv)
        check(lib.meant_group_scatter(_p(dgrouped), _p(index), _p(dqkv), b, L, G, S, 3 * D, dt, _stream()), "group_scatter")
        check(lib.meant_attn_cls_bwd(_p(qkv), _p(out), L * D, _p(dout), L * D, _p(stats), _p(cm), _p(dqkv), b, L, H, Dh, scale, dt, _stream()),
              "attn_cls_bwd")
        return dqkv, None, None, None, None, None, None


def divided_attention(qkv, plan, tables, H, scale, group_mask=None, cls_mask=None):
    return _DividedAttention.apply(qkv, plan, tables, int(H), float(scale), group_mask, cls_mask)


class _TokenShift(torch.autograd.Function):
    """PreTokenShift (src/meant/timesformer_pytorch.py:28-53) on x [b, 1 + f n, d]"""

    @staticmethod
    def forward(ctx, x, f, n):
        _need_gpu(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(lib.meant_token_shift(_p(x), _p(y), x.shape[0], f, n, x.shape[2], 0, _dt(x), _stream()), "token_shift")
        ctx.fn = (f, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib.meant_token_shift(_p(dy), _p(dx), dy.shape[0], ctx.fn[0], ctx.fn[1], dy.shape[2], 1, _dt(dy), _stream()), "token_shift")
        return dx, None, None


def token_shift(x, f, n):
    return _TokenShift.apply(x, int(f), int(n))


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        _need_gpu(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(lib.meant_dropout(_p(x), _p(y), x.numel(), p, seed, _dt(x), _stream()), "dropout")
        ctx.args = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib.meant_dropout(_p(dy), _p(dx), dy.numel(), ctx.args[0], ctx.args[1], _dt(dy), _stream()), "dropout")
        return dx, None, None


def dropout(x, p: float, seed: int):
    """inverted dropout with a counter-based mask (statistically nn.Dropout; the mask is not torch's)"""
    return x if p <= 0.0 else _Dropout.apply(x, float(p), int(seed))


class _ClsConcat(torch.autograd.Function):
    """torch.cat((cls_token expanded over the batch, tokens), dim=1) (src/meant/timesformer_pytorch.py:211-213) as one row
    gather: x[b, 0] = cls, x[b, 1 + t] = tokens[b, t]"""

    @staticmethod
    def forward(ctx, cls_token, tokens, idx_fwd, idx_bwd):
        _need_gpu(cls_token, tokens)
        b, n, d = tokens.shape
        fill = _c(cls_token.detach().to(tokens.dtype).reshape(d))
        x = gather_rows(_c(tokens).view(b * n, d), idx_fwd, fill).view(b, n + 1, d)
        ctx.idx_bwd, ctx.shape, ctx.cls_shape = idx_bwd, (b, n, d), cls_token.shape
        return x

    @staticmethod
    def backward(ctx, dx):
        b, n, d = ctx.shape
        dx = _c(dx)
        dtok = gather_rows(dx.view(b * (n + 1), d), ctx.idx_bwd).view(b, n, d)
        dcls = dx[:, 0, :].float().sum(dim=0).view(ctx.cls_shape)          # b rows of d: plumbing-sized
        return dcls, dtok, None, None


def cls_concat(cls_token, tokens, idx_fwd, idx_bwd):
    return _ClsConcat.apply(cls_token, tokens, idx_fwd, idx_bwd)
