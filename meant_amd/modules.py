"""Host-side mirror of the reference's nn.Module surface for the encoder hot path.

Same class names, constructor / forward signatures and state_dict keys as the reference
(top-level tree), so `meant.meant(...)`, checkpoints and the training
drivers drop in.  The arithmetic is NOT here: every forward below is a short sequence of
calls into libmeant_hip.so through meant_amd.ops (fused QKV+rotary+attention, Linear with
bias/GELU/residual/sigmoid epilogues, RMSNorm with fused dropout, mean-pool+concat, temporal
attention).  There is no eager fallback: tensors must live on an MI355X.

Reference map (file:line under the reference root):
  RMSNorm              utils/rms_norm.py:16-57
  RotaryEmbedding      meant/rotary_embedding_torch.py:58-147  (only what the path uses)
  attention            meant/attention.py:11-62
  xPosAttention        meant/xPosAttention.py:11-66
  temporal             meant/temporal.py:12-60
  visionEncoder / languageEncoder / temporalEncoder / meant    meant/meant.py:35-238
  meant_vision         meant/meant_vision.py:79-165
  meant_tweet          meant/meant_tweet.py:85-167
  meant_vqa            meant/meant_vqa.py:143-234
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import ops
from ._lib import EPI_NONE, EPI_GELU, EPI_SIGMOID


# Compose the encoder's Linear(d,d) with the q/k/v Linears that follow it without any nonlinearity
# (meant/meant.py:60-61,102-103): same function and gradients, one per-token projection less.  Set to False to
# run the two Linears separately, exactly as the reference's module list does.
COMPOSE_PRE_LINEAR = True

# The last Linear (+ residual) of the LAST encoder layer of a stack feeds nothing but the sequence mean-pool
# (meant/meant.py:231, meant_vision.py:159, meant_tweet.py:161, meant_vqa.py:228): mean_s(h W^T + b + x) is evaluated as
# mean_s(h) W^T + b + mean_s(x) (ops.pool_linear_cat) -- same function, same gradients, three token-sized GEMMs less per
# stack.  Set to False to run the module list literally.
POOL_LAST_LINEAR = True

# Run the vision stack of `meant` on a second HIP stream, concurrently with the language stack
# (MEANT_TWO_STREAMS=0 in the environment turns it off).
import os as _os
TWO_STREAMS = _os.environ.get("MEANT_TWO_STREAMS", "1") != "0"
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


# With TWO_STREAMS the LANGUAGE stack -- the longer of the two, i.e. the step's critical path -- runs on a high-priority stream of its
# own, so that its kernels are placed ahead of the vision stack's when both are waiting for CUs: 43.00 -> 42.90 ms median over three
# alternating pairs of 30-step runs on one box (+0.2 %; MEANT_LANG_PRIORITY=0 runs it on the caller's stream as before)
# Default: on for a single-process run, OFF wherever a gradient collective runs beside the backward pass (a multi-process launch,
# WORLD_SIZE > 1, or its one-GPU rehearsal MEANT_REDUCE_ALWAYS=1).  Measured in round 4 (profiles/r04_stream_queues.txt,
# tools/lab/stream_cliff*.sh: one-rank RCCL group on one MI355X, three alternating runs each, stream -> hardware-queue map from
# rocprofv3 --kernel-trace): plain step 37.7 ms; with the reducer's collectives and the priority stream 39.0 ms on the runtime's
# default four hardware queues (main and the language stream end up sharing one) and 42.4 ms -- the "fifth-stream cliff" of round 3 --
# once every stream has a queue of its own (GPU_MAX_HW_QUEUES=8): three compute queues, one of them high priority, plus the
# collective's.  Without the priority stream (language stack on the caller's stream, two compute queues + the collective's):
# 39.5 ms on four queues, 38.3 ms on eight, the best of all: bench.py exports GPU_MAX_HW_QUEUES=8 for such launches.
# MEANT_LANG_PRIORITY=1 forces the priority stream on.
_multi = int(_os.environ.get("WORLD_SIZE", "1") or "1") > 1 or _os.environ.get("MEANT_REDUCE_ALWAYS") == "1"
LANG_PRIORITY = _os.environ.get("MEANT_LANG_PRIORITY", "0" if _multi else "1") != "0"
_HI = {}


def _hi_stream(device):
    key = (device.type, device.index)
    if key not in _HI:
        _HI[key] = torch.cuda.Stream(device=device, priority=-1)
    return _HI[key]


# ------------------------------------------------------------------------------------------
# precision tier selection
def resolve_compute_dtype(module: nn.Module, like: Optional[torch.Tensor]) -> torch.dtype:
    """bf16 when the caller runs under torch.autocast (the reference trains under fp16 autocast,
    in_loop_train.py:215) or hands over half-precision pixels; otherwise fp32.  A model-level
    override is `model.compute_dtype = torch.bfloat16`."""
    forced = getattr(module, "compute_dtype", None)
    if forced is not None:
        return forced
    if torch.is_autocast_enabled():
        return torch.bfloat16
    if like is not None and like.dtype in (torch.bfloat16, torch.float16):
        return torch.bfloat16
    return torch.float32


# ------------------------------------------------------------------------------------------
class RMSNorm(nn.Module):
    """utils/rms_norm.py:16-57.  The full-width, bias-free form (p = -1, bias = False) that every MEANT model uses runs on the packed
    / fused kernels; the partial form (0 <= p <= 1: statistics over the first int(d p) elements, :44-50) and the bias form (a learned
    `offset`, :35-37, :54-55) run on the generic one-row-per-wave kernels (meant_rmsnorm_partial_*), without the fusions."""

    def __init__(self, d, p=-1., eps=1e-8, bias=False):
        super().__init__()
        self.eps, self.d, self.p, self.bias = eps, d, p, bias
        self.scale = nn.Parameter(torch.ones(d))
        if bias:
            self.offset = nn.Parameter(torch.zeros(d))

    def forward(self, x, drop_p: float = 0.0, seed: int = 0):
        partial = 0. <= self.p <= 1.
        if partial or self.bias:
            d_part = int(self.d * self.p) if partial else self.d
            if d_part < 1:
                raise ValueError(f"meant_amd.RMSNorm: p = {self.p} leaves no element to take the statistics over (d = {self.d})")
            y = ops.rmsnorm_partial(x, self.scale, self.offset if self.bias else None, d_part, self.eps)
            return ops.dropout(y, drop_p, seed) if drop_p > 0.0 else y
        return ops.rmsnorm(x, self.scale, self.eps, drop_p, seed)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps)


class Linear(nn.Linear):
    def forward(self, x, residual=None, epilogue=EPI_NONE):
        return ops.linear(x, self.weight, self.bias, residual, epilogue)


class RotaryEmbedding(nn.Module):
    """The subset of meant/rotary_embedding_torch.py:58-147 the path needs: `freqs` (frozen
    Parameter) and the xPos `scale` buffer under the reference's names, plus the per-length
    tables the HIP rotary kernel consumes.  Tables are built the reference's way (fp32 angles
    pos*freq, :141; xPos power (pos - S//2)/scale_base, :121; block-concatenated scale, :125)."""

    def __init__(self, dim, custom_freqs=None, freqs_for='lang', theta=10000, max_freq=10, num_freqs=1,
                 learned_freq=False, use_xpos=False, xpos_scale_base=512):
        super().__init__()
        if custom_freqs is not None:
            freqs = custom_freqs
        elif freqs_for == 'lang':
            freqs = 1. / (theta ** (torch.arange(0, dim, 2)[:(dim // 2)].float() / dim))
        elif freqs_for == 'pixel':
            freqs = torch.linspace(1., max_freq / 2, dim // 2) * math.pi
        elif freqs_for == 'constant':
            freqs = torch.ones(num_freqs).float()
        else:
            raise ValueError(f'unknown modality {freqs_for}')
        # learned_freq (rotary_embedding_torch.py:67,85: `freqs` is then a trainable Parameter; no reference model sets it): the fused
        # kernels rotate inside the projection GEMM's epilogue and undo it inside the attention backward, neither of which returns a
        # gradient for the tables -- an attention module whose rotary has learned frequencies therefore takes the UNFUSED route
        # (`learned_rotary_attention` below: projection GEMM, rotation as differentiable tensor ops on the device, attention core without tables)
        self.learned_freq = bool(learned_freq)
        self.freqs = nn.Parameter(freqs, requires_grad=self.learned_freq)
        self.use_xpos = use_xpos
        self.scale_base = xpos_scale_base
        if use_xpos:
            self.register_buffer('scale', (torch.arange(0, dim, 2) + 0.4 * dim) / (1.4 * dim))
        else:
            self.register_buffer('scale', None)
        self._tables = {}

    def __getstate__(self):
        st = self.__dict__.copy()
        st['_tables'] = {}
        return st

    @property
    def rot_dim(self) -> int:
        return 2 * self.freqs.numel()

    def tables_autograd(self, seq_len: int) -> tuple:
        """the same four tables as differentiable functions of `freqs`, on the parameter's device (learned_freq)"""
        f = self.freqs.float()
        pos = torch.arange(seq_len, device=f.device, dtype=f.dtype)
        ang = torch.repeat_interleave(pos[:, None] * f[None, :], 2, dim=-1)
        cos, sin = ang.cos(), ang.sin()
        if not self.use_xpos:
            return cos, sin, cos, sin
        power = (torch.arange(seq_len, device=f.device) - seq_len // 2) / self.scale_base
        s = self.scale.float()[None, :] ** power[:, None]
        s = torch.cat((s, s), dim=-1)
        si = s ** -1
        return cos * s, sin * s, cos * si, sin * si

    def tables(self, seq_len: int, device) -> tuple:
        """(qa, qb, ka, kb) float32 [S, R] on `device`: out = t*a + rot(t)*b."""
        key = (seq_len, str(device), self.freqs._version, self.freqs.data_ptr())
        hit = self._tables.get(key)
        if hit is not None:
            return hit
        with torch.no_grad():
            f = self.freqs.detach().float().cpu()
            pos = torch.arange(seq_len)
            ang = torch.repeat_interleave(pos.to(f.dtype)[:, None] * f[None, :], 2, dim=-1)
            cos, sin = ang.cos(), ang.sin()
            if self.use_xpos:
                power = (pos - seq_len // 2) / self.scale_base
                s = self.scale.detach().float().cpu()[None, :] ** power[:, None]
                s = torch.cat((s, s), dim=-1)
                si = s ** -1
                t = (cos * s, sin * s, cos * si, sin * si)
            else:
                t = (cos, sin, cos, sin)
            t = tuple(x.contiguous().to(device) for x in t)
            if not self.use_xpos:
                t = (t[0], t[1], t[0], t[1])
        self._tables = {k: v for k, v in self._tables.items() if k[2:] == key[2:]}
        self._tables[key] = t
        return t


def _rotate_pairs(t, a, b):
    """t [G, S, H, Dh]; a, b [S, R]: out[..., :R] = t * a + rot(t) * b with rot(t)[2j] = -t[2j+1], rot(t)[2j+1] = t[2j]
    (what meant_rotary_qk / the projection GEMM's epilogue compute), in fp32, rounded back to t's dtype; columns past R pass through"""
    R = a.shape[1]
    tr = t[..., :R].float()
    pr = tr.reshape(*tr.shape[:-1], R // 2, 2)
    rot = torch.stack((-pr[..., 1], pr[..., 0]), dim=-1).reshape(tr.shape)
    out = tr * a[None, :, None, :] + rot * b[None, :, None, :]
    return torch.cat((out.to(t.dtype), t[..., R:]), dim=-1)


def learned_rotary_attention(x, mod, rot: RotaryEmbedding, key_mask, causal: bool, pre=None):
    """softmax((rot q)(rot k)^T / sqrt(dim) [+ mask]) v for a rotary embedding with LEARNED frequencies: the q|k|v projection as one GEMM
    (ops.linear), the rotation as tensor ops whose tables carry the gradient to `rot.freqs`, the attention core (HIP) on the rotated
    buffer without tables.  `mod` holds the q / v / k Linears under the reference's names (the one called `v` produces the keys)."""
    G, S, d = x.shape
    H = mod.num_heads
    wqkv = torch.cat([mod.q.weight, mod.v.weight, mod.k.weight], dim=0)
    bqkv = torch.cat([mod.q.bias, mod.v.bias, mod.k.bias], dim=0)
    if pre is not None:
        wqkv, bqkv = ops.compose_linear(pre.weight, pre.bias, wqkv, bqkv)
    D = wqkv.shape[0] // 3
    qkv = ops.linear(x.reshape(G * S, d), wqkv, bqkv).view(G, S, 3, H, D // H)
    qa, qb, ka, kb = rot.tables_autograd(S)
    q, k = _rotate_pairs(qkv[:, :, 0], qa, qb), _rotate_pairs(qkv[:, :, 1], ka, kb)
    packed = torch.stack((q, k, qkv[:, :, 2]), dim=2).reshape(G * S, 3 * D)
    return ops.attention_core(packed, G, S, H, 1.0 / math.sqrt(D), None, causal, key_mask).view(G, S, D)


def score_dropout_attention(x, mod, rot: RotaryEmbedding, key_mask, causal: bool, p: float, seed: int, pre=None):
    """xPosAttention with droput > 0 in train mode (meant/xPosAttention.py:59 drops SCORES, after the masks and before the softmax):
    the q|k|v projection as one GEMM, then the materialised attention core with the dropout inside (ops.attention_core_score_dropout),
    rotary by tables as everywhere else.  With learned rotary frequencies the rotation is done here, differentiably."""
    G, S, d = x.shape
    H = mod.num_heads
    wqkv = torch.cat([mod.q.weight, mod.v.weight, mod.k.weight], dim=0)
    bqkv = torch.cat([mod.q.bias, mod.v.bias, mod.k.bias], dim=0)
    if pre is not None:
        wqkv, bqkv = ops.compose_linear(pre.weight, pre.bias, wqkv, bqkv)
    D = wqkv.shape[0] // 3
    qkv = ops.linear(x.reshape(G * S, d), wqkv, bqkv)
    tables = None
    if rot is not None and getattr(rot, 'learned_freq', False):
        q5 = qkv.view(G, S, 3, H, D // H)
        qa, qb, ka, kb = rot.tables_autograd(S)
        qkv = torch.stack((_rotate_pairs(q5[:, :, 0], qa, qb), _rotate_pairs(q5[:, :, 1], ka, kb), q5[:, :, 2]), dim=2).reshape(G * S, 3 * D)
    elif rot is not None:
        tables = rot.tables(S, x.device)
    return ops.attention_core_score_dropout(qkv, G, S, H, 1.0 / math.sqrt(D), p, seed, tables, causal, key_mask).view(G, S, D)


# ------------------------------------------------------------------------------------------
class attention(nn.Module):
    """meant/attention.py:11-62: MHA over patches, pixel rotary on q,k, no mask, scale 1/sqrt(dim).
    The Linear named `v` produces KEYS and the one named `k` produces VALUES (:36-37)."""

    def __init__(self, num_heads, dim, pos_emb: RotaryEmbedding, mask=False, droput=0.):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = int(dim / num_heads)
        self.dropout = nn.Dropout(droput)
        self.pos_emb = pos_emb
        self.mask = mask
        self.softmax = nn.Softmax(dim=-1)
        self.multi_mad = Linear(self.num_heads * self.Dh, self.dim)
        self.q = Linear(self.dim, self.Dh * self.num_heads)
        self.v = Linear(self.dim, self.Dh * self.num_heads)
        self.k = Linear(self.dim, self.Dh * self.num_heads)

    def core(self, x, pre=None):
        """pre: an nn.Linear that the caller would have applied to x right before this module (composed into q/k/v)"""
        if self.pos_emb is not None and getattr(self.pos_emb, 'learned_freq', False):
            return learned_rotary_attention(x, self, self.pos_emb, None, bool(self.mask), pre)
        tables = self.pos_emb.tables(x.shape[1], x.device) if self.pos_emb is not None else None
        prew = (pre.weight, pre.bias) if pre is not None else None
        return ops.qkv_attention(x, self.q.weight, self.q.bias, self.v.weight, self.v.bias, self.k.weight, self.k.bias,
                                 tables, None, bool(self.mask), self.num_heads, pre=prew)

    def forward(self, input):
        out = self.multi_mad(self.core(input))
        if self.training and self.dropout.p > 0:
            out = self.dropout(out)
        return out


class xPosAttention(nn.Module):
    """meant/xPosAttention.py:11-66: as `attention` with xPos on q,k (:39), always-causal mask
    (:43-50) and the additive (1-mask)*-1e9 key-padding term (:54-56)."""

    def __init__(self, num_heads, dim, xPos: RotaryEmbedding, mask=True, droput=0.):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = int(dim / num_heads)
        self.dropout = nn.Dropout(droput)
        self.xPos = xPos
        self.mask = mask
        self.softmax = nn.Softmax(dim=-1)
        self.multi_mad = Linear(self.num_heads * self.Dh, self.dim)
        self.q = Linear(self.dim, self.Dh * self.num_heads)
        self.v = Linear(self.dim, self.Dh * self.num_heads)
        self.k = Linear(self.dim, self.Dh * self.num_heads)

    def core(self, x, attention_mask=None, pre=None):
        if self.training and self.dropout.p > 0:              # xPosAttention.py:59: dropout on the SCORES (no reference model sets it)
            return score_dropout_attention(x, self, self.xPos, attention_mask, bool(self.mask), float(self.dropout.p), _seed(), pre)
        if getattr(self.xPos, 'learned_freq', False):
            return learned_rotary_attention(x, self, self.xPos, attention_mask, bool(self.mask), pre)
        tables = self.xPos.tables(x.shape[1], x.device)
        prew = (pre.weight, pre.bias) if pre is not None else None
        return ops.qkv_attention(x, self.q.weight, self.q.bias, self.v.weight, self.v.bias, self.k.weight, self.k.bias,
                                 tables, attention_mask, bool(self.mask), self.num_heads, pre=prew)

    def forward(self, input, attention_mask=None):
        return self.multi_mad(self.core(input, attention_mask))


class temporal(nn.Module):
    """meant/temporal.py:12-60: the query is the last lag step only (:39), keys/values all L steps."""

    def __init__(self, num_heads, dim, mask=False, droput=0.):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = int(dim / num_heads)
        self.dropout = nn.Dropout(droput)
        self.mask = mask
        self.softmax = nn.Softmax(dim=-1)
        self.multi_mad = Linear(self.num_heads * self.Dh, self.dim)
        self.atten_size = self.Dh * self.num_heads
        self.q = Linear(self.dim, self.atten_size)
        self.v = Linear(self.dim, self.atten_size)
        self.k = Linear(self.dim, self.atten_size)

    def forward(self, input):
        o = ops.temporal_attention(input, self.q.weight, self.q.bias, self.v.weight, self.v.bias, self.k.weight, self.k.bias,
                                   self.num_heads)
        return self.multi_mad(o)


# ------------------------------------------------------------------------------------------
def _seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class visionEncoder(nn.Module):
    """meant/meant.py:35-75."""

    def __init__(self, dim, num_heads, flash=False):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.posEmbed = RotaryEmbedding(dim=math.floor(dim / num_heads / 2), freqs_for='pixel')
        atten = attention(num_heads, dim, self.posEmbed)          # `flash` ignored: single HIP backend
        self.encode = nn.ModuleList([RMSNorm(dim), Linear(dim, dim), atten, RMSNorm(dim), Linear(dim, dim)])
        self.encode2 = nn.ModuleList([RMSNorm(dim), Linear(dim, dim), nn.GELU(), RMSNorm(dim), Linear(dim, dim)])

    def forward(self, input, pool: bool = False, fold=None):
        """pool=True: return (h, res) instead of encode2[-1](h) + res, for ops.pool_linear_cat.
        fold: the stack's RMSNorm-into-Linear decision (an explicit argument so that a recomputation under
        torch.utils.checkpoint repeats the forward's path); None = by ops.fold_wanted()"""
        e, e2 = self.encode, self.encode2
        n, res = ops.rmsnorm_fork(input, e[0].scale, e[0].eps)      # residual gradient is folded into this norm's backward
        if COMPOSE_PRE_LINEAR and e[1].bias is not None:
            h = e[2].multi_mad(e[2].core(n, pre=e[1]))               # Linear(d,d) composed into the q/k/v projections
        else:
            h = e[2](e[1](n))
        h = e[3](h)
        x1 = e[4](h, residual=res)
        fold = ops.norm_linear_ok(x1, e2[1].weight, fold)                 # encode2[0] rides encode2[1]'s GEMM as a per-row factor
        if pool and ops.pooled_norm_ok(x1, x1.shape[1]):             # the two norms beside the mean-pool emit the means themselves
            if fold:
                return ops.norm_linear_gelu_norm_pooled(x1, e2[0].scale, e2[0].eps, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps)
            n, xm = ops.rmsnorm_fork_pooled(x1, e2[0].scale, e2[0].eps)
            return ops.linear_gelu_rmsnorm_pooled(n, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps), xm
        if fold:
            h, res = ops.norm_linear_gelu_norm(x1, e2[0].scale, e2[0].eps, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps)
        else:
            n, res = ops.rmsnorm_fork(x1, e2[0].scale, e2[0].eps)
            h = ops.linear_gelu_rmsnorm(n, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps)
        if pool:
            return h, res
        return e2[4](h, residual=res)


class languageEncoder(nn.Module):
    """meant/meant.py:78-120.  Dropout(dropout) at encode[4] and the default-p Dropout() at
    encode2[4] (:105,:107) are fused into the preceding RMSNorm kernel in training mode."""

    def __init__(self, dim, num_heads, dropout=0.0, flash=False):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.xPos = RotaryEmbedding(dim=48, use_xpos=True)
        att = xPosAttention(num_heads, dim, self.xPos)
        self.encode = nn.ModuleList([RMSNorm(dim), Linear(dim, dim), att, RMSNorm(dim), nn.Dropout(dropout), Linear(dim, dim)])
        self.encode2 = nn.ModuleList([RMSNorm(dim), Linear(dim, dim), nn.GELU(), RMSNorm(dim), nn.Dropout(), Linear(dim, dim)])

    def forward(self, input, attention_mask=None, pool: bool = False, fold=None):
        """pool=True: return (h, res) instead of encode2[-1](h) + res, for ops.pool_linear_cat; fold: as visionEncoder.forward"""
        e, e2 = self.encode, self.encode2
        p1 = e[4].p if self.training else 0.0
        p2 = e2[4].p if self.training else 0.0
        n, res = ops.rmsnorm_fork(input, e[0].scale, e[0].eps)
        if COMPOSE_PRE_LINEAR and e[1].bias is not None:
            h = e[2].multi_mad(e[2].core(n, attention_mask, pre=e[1]))
        else:
            h = e[2](e[1](n), attention_mask)
        h = e[3](h, drop_p=p1, seed=_seed() if p1 > 0 else 0)
        x1 = e[5](h, residual=res)
        fold = ops.norm_linear_ok(x1, e2[1].weight, fold)                 # encode2[0] rides encode2[1]'s GEMM as a per-row factor
        seed2 = _seed() if p2 > 0 else 0
        if pool and ops.pooled_norm_ok(x1, x1.shape[1]):             # the two norms beside the mean-pool emit the means themselves
            if fold:
                return ops.norm_linear_gelu_norm_pooled(x1, e2[0].scale, e2[0].eps, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps, p2, seed2)
            n, xm = ops.rmsnorm_fork_pooled(x1, e2[0].scale, e2[0].eps)
            return ops.linear_gelu_rmsnorm_pooled(n, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps, p2, seed2), xm
        if fold:
            h, res = ops.norm_linear_gelu_norm(x1, e2[0].scale, e2[0].eps, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps, p2, seed2)
        else:
            n, res = ops.rmsnorm_fork(x1, e2[0].scale, e2[0].eps)
            h = ops.linear_gelu_rmsnorm(n, e2[1].weight, e2[1].bias, e2[3].scale, e2[3].eps, p2, seed2)
        if pool:
            return h, res
        return e2[5](h, residual=res)


class temporalEncoder(nn.Module):
    """meant/meant.py:124-145; `norms=False` gives the variant of meant/meant_vision.py:79-105 and
    meant/meant_tweet.py:85-110 (no RMSNorms, so temp_encode has 3 entries)."""

    def __init__(self, dim, num_heads, lag, norms=True):
        super().__init__()
        self.dim, self.num_heads, self.lag = dim, num_heads, lag
        self.temp_embedding = nn.Parameter(torch.randn(1, lag, dim))
        if norms:
            mods = [RMSNorm(dim), Linear(dim, dim), temporal(num_heads, dim), RMSNorm(dim), Linear(dim, dim)]
        else:
            mods = [Linear(dim, dim), temporal(num_heads, dim), Linear(dim, dim)]
        self.temp_encode = nn.ModuleList(mods)

    def forward(self, x):
        x = ops.add_rowvec(x, self.temp_embedding)
        for mod in self.temp_encode:
            x = mod(x)
        return x


class _Patchify(nn.Module):
    """einops Rearrange('b c (h p1) (w p2) -> b (h w) (p1 p2 c)') of meant/meant.py:194, emitting the
    compute dtype.  Parameter-free, sits at index 0 so the Linear keeps the key `patchEmbed.1.*`.
    Accepts raw float64 / uint8 / float32 / bf16 pixels; `set_normalization(mean, std)` folds the data set's global
    (x - mean) / std (in_loop_train.py:591-593) into the same pass (not part of the state_dict, like the reference's)."""

    def __init__(self, p):
        super().__init__()
        self.p = p
        self.out_dtype = torch.float32
        self.norm_mean, self.norm_std = 0.0, 1.0

    def set_normalization(self, mean: float, std: float):
        self.norm_mean, self.norm_std = float(mean), float(std)

    def forward(self, images):
        return ops.patchify(images, self.p, self.out_dtype, self.norm_mean, self.norm_std)


class _PatchEmbed(nn.Sequential):
    def __init__(self, patch_dim, dim, p):
        super().__init__(_Patchify(p), Linear(patch_dim, dim))

    def forward(self, images, dtype=torch.float32):
        self[0].out_dtype = dtype
        return self[1](self[0](images))


def _run_encoder(enc, x, *args, checkpoint: bool = False, **kw):
    """one encoder layer, optionally under activation recomputation: with `model.activation_checkpointing = True` only
    the layer input is kept and the layer is re-run in backward (torch.utils.checkpoint restores the CPU RNG state, from
    which the dropout seeds are drawn, so the recomputed masks are the forward's).  At the reference's CLI default of
    12 encoder layers, 128 samples per GPU would otherwise save ~330 GB of activations."""
    if checkpoint and torch.is_grad_enabled() and x.requires_grad:
        from torch.utils.checkpoint import checkpoint as _ckpt
        return _ckpt(enc, x, *args, use_reentrant=False, preserve_rng_state=True, **kw)
    return enc(x, *args, **kw)


def _ckpt_setting(model):
    """model.activation_checkpointing: False / True (every encoder layer is recomputed in backward) / an int n (only the first n
    layers of each stack are: their recomputation comes last in backward, when the later layers' activations are already
    freed, so the peak is the (layers - n) kept layers -- 12 layers at 128 samples per GPU fit the 288 GB with n = 4 and pay a
    third of the full recomputation)"""
    v = getattr(model, "activation_checkpointing", False)
    return v if isinstance(v, bool) else int(v)


def _run_stack(encoders, x, *args, checkpoint=False):
    """all encoder layers of one stack; returns either the token tensor, or -- with POOL_LAST_LINEAR -- the
    (h, res, weight, bias) part that ops.pool_linear_cat pools.  checkpoint: bool, or the number of leading layers to recompute"""
    n = len(encoders)
    # big stacks fold their norms into the consumer Linears.  Decided ONCE per stack and passed down as an argument: the layers run
    # under torch.utils.checkpoint are re-run in backward, after the OTHER stack may have left a different process-wide hint
    fold = ops.stack_fold(n, x.numel() // x.shape[-1], x.shape[-1])
    for i, enc in enumerate(encoders):
        ck = checkpoint if isinstance(checkpoint, bool) else i < checkpoint
        last = enc.encode2[-1]
        if POOL_LAST_LINEAR and i == n - 1 and isinstance(last, Linear):
            h, res = _run_encoder(enc, x, *args, checkpoint=ck, pool=True, fold=fold)
            return (h, res, last.weight, last.bias)
        x = _run_encoder(enc, x, *args, checkpoint=ck, fold=fold)
    return x


def _pool_parts(compute_dtype, *stacks):
    """stacks: outputs of _run_stack (a part tuple or a token tensor) -> pooled, concatenated features [G, sum d]"""
    if all(isinstance(st, tuple) for st in stacks):
        if all(st[0].dim() == 3 for st in stacks):        # (h, x, W, b) token tensors
            return ops.pool_linear_cat(list(stacks))
        # (mean_s h, mean_s x, W, b): the norm kernels pooled already; a stack whose shape they do not cover is pooled here
        parts = [st if st[0].dim() == 2 else (ops.meanpool_f32(st[0]), ops.meanpool_f32(st[1]), st[2], st[3]) for st in stacks]
        return ops.pooled_linear_cat(parts, compute_dtype)
    toks = []
    for st in stacks:
        if isinstance(st, tuple):                         # a stack without the pooled tail beside one with it: literal way
            h, res, w, b = st
            if h.dim() == 2:
                raise RuntimeError("meant_amd: one stack pooled its last layer, the other did not")
            st = ops.linear(h, w, b, residual=res)
        toks.append(st)
    return ops.meanpool_cat(*toks)


POOL_OWN_STREAM = _os.environ.get("MEANT_POOL_OWN_STREAM", "1") == "1"


def _pool_own_stream(compute_dtype, st):
    """the pooled tail of ONE stack (mean over the sequence + the last Linear on the means) where the stack itself ran, instead of
    both stacks' after the streams have joined: the vision stack finishes long before the language stack, so its two small fp32
    products (and their three in backward, which autograd replays on this stream) leave the step's serial stretch.
    Same values: the parts are rounded to the compute dtype before the concatenation instead of after it."""
    if not POOL_OWN_STREAM or not (isinstance(st, tuple) and st[0].dim() == 2):
        return st
    return ops.pooled_linear_cat([st], compute_dtype)


def _embed(mods, ids, dtype):
    """meant/meant.py:210-211.  A plain nn.Embedding is served by the HIP gather kernel; any other
    user module (e.g. HF RobertaEmbeddings) is called as is and its output cast."""
    x = ids
    for mod in mods:
        if isinstance(mod, nn.Embedding) and mod.max_norm is None and not mod.sparse and x.dtype in (torch.int64, torch.int32):
            x = ops.embedding(x, mod.weight, dtype)
        else:
            x = mod(x)
    if x.dtype != dtype:
        x = x.to(dtype)
    return x


HEAD_F32_MAX_CLASSES = 64


def _head(mods, x):
    """[norm, Linear, Sigmoid] heads (meant/meant.py:204): the sigmoid rides the GEMM epilogue.
    A head of at most HEAD_F32_MAX_CLASSES outputs is evaluated in fp32 in either tier (its [B, classes] product has no MFMA tile
    to fill; the bf16 tier ran it on the exact-f32 generic kernel already and only STORED the probabilities as bf16).  That
    rounding -- 2^-9 of a probability the loss differentiates -- was the largest single error of the tier's gradients: on the
    full-dims golden one flipped output bit moved every gradient norm by 4 % (tools/debug_golden_c3.py)."""
    x = mods[0](x)
    if x.dtype != torch.float32 and mods[1].weight.shape[0] <= HEAD_F32_MAX_CLASSES:
        x = x.float()
    if isinstance(mods[2], nn.Sigmoid):
        return mods[1](x, epilogue=EPI_SIGMOID)
    return mods[2](mods[1](x))


class meant(nn.Module):
    """meant/meant.py:148-238."""

    def __init__(self, text_dim, image_dim, price_dim, height, width, patch_res, lag, num_classes, embedding, flash=False,
                 num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.lag, self.text_dim, self.image_dim = lag, text_dim, image_dim
        self.dim = text_dim + image_dim
        self.num_heads = num_heads
        self.channels = channels
        self.patch_dim = channels * patch_res * patch_res
        self.n = int((height * width) / (patch_res ** 2))
        self.embedding = nn.ModuleList([embedding])
        self.patchEmbed = _PatchEmbed(self.patch_dim, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag)])
        self.mlpHead = nn.ModuleList([RMSNorm(self.dim), Linear(self.dim, num_classes), nn.Sigmoid()])
        self.compute_dtype = None

    def forward(self, tweets, images, attention_mask=None):
        dt = resolve_compute_dtype(self, images)
        B = images.shape[0]
        if images.is_cuda and TWO_STREAMS:
            ops.set_index_stream(images.device, _side_stream(images.device))     # the id sort rides the vision stream
        words = _embed(self.embedding, tweets.reshape(B * self.lag, tweets.shape[2]), dt)
        if attention_mask is not None:
            attention_mask = attention_mask.reshape(B * self.lag, attention_mask.shape[2])
        # The language and the vision stacks are independent until the pooled concat: with TWO_STREAMS the vision
        # stack runs on a second HIP stream so that kernels with different bottlenecks (HBM-bound norms, load-path-bound
        # GEMMs, issue-bound attention) of the two stacks can share the chip.  Autograd replays each backward op on the
        # stream of its forward op, so the backward passes overlap the same way.
        ck = _ckpt_setting(self)
        side = _side_stream(images.device) if (TWO_STREAMS and images.is_cuda and not ck) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                img = self.patchEmbed(images.reshape(B * self.lag, *images.shape[2:]), dt)
                img = _run_stack(self.visionEncoders, img, checkpoint=ck)
                img = _pool_own_stream(dt, img)
        if side is not None and LANG_PRIORITY:
            hi = _hi_stream(images.device)
            hi.wait_stream(main)
            with torch.cuda.stream(hi):
                words = _run_stack(self.languageEncoders, words, attention_mask, checkpoint=ck)
                words = _pool_own_stream(dt, words)
            main.wait_stream(hi)
            for tt in (words if isinstance(words, tuple) else (words,)):
                if tt is not None and tt.is_cuda:
                    tt.record_stream(main)
        else:
            words = _run_stack(self.languageEncoders, words, attention_mask, checkpoint=ck)
        if side is not None:
            main.wait_stream(side)
            for tt in (img if isinstance(img, tuple) else (img,)):
                if tt is not None and tt.is_cuda:
                    tt.record_stream(main)
        else:
            img = self.patchEmbed(images.reshape(B * self.lag, *images.shape[2:]), dt)
            img = _run_stack(self.visionEncoders, img, checkpoint=ck)
        done = [torch.is_tensor(st) and st.dim() == 2 for st in (words, img)]       # pooled on its own stream already
        if any(done):
            fused = torch.cat([st if d_ else _pool_parts(dt, st) for st, d_ in zip((words, img), done)], dim=1).view(B, self.lag, self.dim)
        else:
            fused = _pool_parts(dt, words, img).view(B, self.lag, self.dim)
        for enc in self.temporal_encoding:
            fused = enc(fused)
        return _head(self.mlpHead, fused).squeeze(dim=1).float()


class meant_vision(nn.Module):
    """meant/meant_vision.py:107-165."""

    def __init__(self, image_dim, price_dim, height, width, patch_res, lag, num_classes, flash=False, num_heads=8,
                 num_encoders=1, channels=4):
        super().__init__()
        self.dim = image_dim
        self.num_heads = num_heads
        self.channels = channels
        self.patch_dim = channels * patch_res * patch_res
        self.n = int((height * width) / (patch_res ** 2))
        self.patchEmbed = _PatchEmbed(self.patch_dim, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag, norms=False)])
        self.mlpHead = nn.ModuleList([LayerNorm(self.dim), Linear(self.dim, num_classes), nn.Sigmoid()])
        self.compute_dtype = None

    def forward(self, images):
        dt = resolve_compute_dtype(self, images)
        B, L = images.shape[0], images.shape[1]
        img = self.patchEmbed(images.reshape(B * L, *images.shape[2:]), dt)
        img = _run_stack(self.visionEncoders, img)
        fused = _pool_parts(dt, img).view(B, L, self.dim)
        for enc in self.temporal_encoding:
            fused = enc(fused)
        return _head(self.mlpHead, fused).squeeze(dim=1).float()


class meant_tweet(nn.Module):
    """meant/meant_tweet.py:114-167 (with the evident intent of its broken :81, == meant/meant.py:109-120)."""

    def __init__(self, text_dim, price_dim, lag, num_classes, embedding, flash=False, num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.dim = text_dim
        self.num_heads = num_heads
        self.embedding = nn.ModuleList([embedding])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag, norms=False)])
        self.mlpHead = nn.ModuleList([LayerNorm(self.dim), Linear(self.dim, num_classes), nn.Sigmoid()])
        self.lag = lag
        self.compute_dtype = None

    def forward(self, tweets, attention_mask=None):
        dt = resolve_compute_dtype(self, None)
        B = tweets.shape[0]
        words = _embed(self.embedding, tweets.reshape(B * self.lag, tweets.shape[2]), dt)
        attention_mask = attention_mask.reshape(B * self.lag, attention_mask.shape[2])     # required, as in :150
        words = _run_stack(self.languageEncoders, words, attention_mask)
        fused = _pool_parts(dt, words).view(B, self.lag, self.dim)
        for enc in self.temporal_encoding:
            fused = enc(fused)
        return _head(self.mlpHead, fused).squeeze(dim=1).float()


class meant_vqa(nn.Module):
    """meant/meant_vqa.py:143-234: no lag axis, no cross-attention in the forward that runs; the
    multimodal_* blocks exist only so that the state_dict matches (:199-200)."""

    def __init__(self, text_dim, image_dim, price_dim, height, width, patch_res, lag, num_classes, embedding, flash=False,
                 num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.lag, self.text_dim, self.image_dim = lag, text_dim, image_dim
        self.dim = text_dim + image_dim
        self.num_heads = num_heads
        self.channels = channels
        self.patch_dim = channels * patch_res * patch_res
        self.n = int((height * width) / (patch_res ** 2))
        self.embedding = nn.ModuleList([embedding])
        self.patchEmbed = _PatchEmbed(self.patch_dim, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.multimodal_embedding = nn.Sequential(nn.Linear(1, self.dim), nn.GELU(), RMSNorm(self.dim), nn.Linear(self.dim, self.dim))
        self.multimodal_encoding = nn.ModuleList([visionEncoder(self.dim, num_heads, flash=flash)])
        self.mlpHead = nn.ModuleList([RMSNorm(self.dim), Linear(self.dim, num_classes), nn.Sigmoid()])
        self.compute_dtype = None

    def forward(self, tweets, images, attention_mask=None):
        dt = resolve_compute_dtype(self, images)
        words = _embed(self.embedding, tweets, dt)
        words = _run_stack(self.languageEncoders, words, attention_mask)
        img = self.patchEmbed(images, dt)
        img = _run_stack(self.visionEncoders, img)
        fused = _pool_parts(dt, words, img)
        return _head(self.mlpHead, fused).float()


# single backend: the reference's flash variants are aliases (meant/flash_attention.py, meant/xPosAttention_flash.py)
flash_attention = attention
xPosAttention_flash = xPosAttention


# ------------------------------------------------------------------------------------------
def _is_roberta_lm_head(h) -> bool:
    """HF `RobertaLMHead`: dense -> gelu -> layer_norm -> decoder (weight usually tied to the word embedding)"""
    return all(hasattr(h, a) for a in ("dense", "layer_norm", "decoder")) and isinstance(getattr(h, "decoder"), nn.Linear)


class meant_language_pretrainer(nn.Module):
    """pretrain_mlm.py:74-88: masked-language-model pretrainer around `num_encoders` languageEncoders.
    `embedding` and `lm_head` are the caller's modules (the reference passes HF `RobertaForMaskedLM(...).roberta.
    embeddings` and `.lm_head`, :318-319) and keep their own parameter names, so `state_dict` keys match the
    reference's (`embedding.0.*`, `languageEncoders.i.*`, `mlm_head.*`).  A head with RobertaLMHead's structure
    runs on the HIP path (GELU in the GEMM epilogue, LayerNorm kernel, padded vocabulary GEMM); any other head is
    called as is.  `forward` returns the logits like the reference; `loss(...)` is forward + CrossEntropyLoss
    (pretrain_mlm.py:160,178) with the logits kept in the padded buffer of the vocabulary GEMM end to end."""

    def __init__(self, num_encoders, mlm_input_dim, embedding, lm_head, flash=False, lag=5, text_dim=768, num_heads=8):
        super().__init__()
        self.embedding = nn.ModuleList([embedding])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads, flash=flash) for _ in range(num_encoders)])
        self.mlm_head = lm_head
        self.lag = lag

    def _encode(self, words, attention_mask):
        dt = resolve_compute_dtype(self, None)
        x = _embed(self.embedding, words, dt)
        fold = ops.stack_fold(len(self.languageEncoders), x.numel() // x.shape[-1], x.shape[-1])
        for enc in self.languageEncoders:
            x = enc(x, attention_mask=attention_mask, fold=fold)
        return x

    def _head_features(self, x):
        h = self.mlm_head
        y = ops.linear(x, h.dense.weight, h.dense.bias, None, EPI_GELU)
        return ops.layernorm(y, h.layer_norm.weight, h.layer_norm.bias, h.layer_norm.eps)

    def forward(self, words, attention_mask=None):       # (meant/hf_wrapper.py:120 calls it without a mask)
        x = self._encode(words, attention_mask)
        h = self.mlm_head
        if _is_roberta_lm_head(h):
            return ops.vocab_linear(self._head_features(x), h.decoder.weight, h.decoder.bias)
        return h(x)

    def loss(self, words, attention_mask, labels, ignore_index: int = -100):
        x = self._encode(words, attention_mask)
        h = self.mlm_head
        if _is_roberta_lm_head(h):
            return ops.vocab_linear_cross_entropy(self._head_features(x), h.decoder.weight, h.decoder.bias, labels, ignore_index)
        return ops.softmax_cross_entropy(h(x), labels, ignore_index)


# ------------------------------------------------------------------------------------------
def _is_vit_mim_decoder(dec) -> bool:
    """HF ViTForMaskedImageModeling.decoder: Sequential(Conv2d(hidden, stride^2 * C_out, kernel 1), PixelShuffle(stride))"""
    return (isinstance(dec, nn.Sequential) and len(dec) == 2 and isinstance(dec[0], nn.Conv2d) and dec[0].kernel_size == (1, 1)
            and dec[0].stride == (1, 1) and dec[0].groups == 1 and isinstance(dec[1], nn.PixelShuffle))


class meant_vision_pretrainer(nn.Module):
    """pretrain_mim.py:77-99: masked-image-modelling pretrainer.  patchify + Linear, ONE visionEncoder
    (`num_encoders` is ignored, as in the reference, :86), then the caller's decoder on the (B, d, sqrt(n), sqrt(n))
    feature map.  `state_dict` keys: `patchEmbed.1.*`, `visionEncoders.0.*`, `decoder.*` (the caller's module, kept as
    is).  When the decoder is HF's ViT masked-image-modelling decoder -- a 1x1 convolution followed by PixelShuffle --
    the convolution runs as the per-token GEMM it is (weight [stride^2 C, d, 1, 1] viewed as [stride^2 C, d]) on the
    HIP path and the pixel shuffle is a view/permute of its output; any other decoder is called as is."""

    def __init__(self, num_encoders, decoder, mlm_input_dim, patch_res=16, channels=4, height=224, width=224, image_dim=768, num_heads=8):
        super().__init__()
        self.channels = channels
        self.patch_dim = channels * patch_res * patch_res
        self.n = int((height * width) / (patch_res ** 2))
        self.patchEmbed = _PatchEmbed(self.patch_dim, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads, flash=True)])
        self.decoder = decoder

    def forward(self, images):
        dt = resolve_compute_dtype(self, images)
        x = self.patchEmbed(images, dt)                                  # [B, n, d]
        fold = ops.stack_fold(len(self.visionEncoders), x.numel() // x.shape[-1], x.shape[-1])
        for enc in self.visionEncoders:
            x = enc(x, fold=fold)
        b, n, c = x.shape
        hw = math.floor(n ** 0.5)
        dec = self.decoder
        if _is_vit_mim_decoder(dec):
            conv, r = dec[0], dec[1].upscale_factor
            y = ops.linear(x, conv.weight.view(conv.out_channels, c), conv.bias)          # [B, n, r*r*C_out], token-major
            co = conv.out_channels // (r * r)
            # PixelShuffle: out[b, co, h*r + i, w*r + j] = y[b, (h, w), co*r*r + i*r + j]
            y = y.view(b, hw, hw, co, r, r).permute(0, 3, 1, 4, 2, 5).reshape(b, co, hw * r, hw * r)
            return y
        return dec(x.permute(0, 2, 1).reshape(b, c, hw, hw))


# ------------------------------------------------------------------------------------------
# divided space-time attention as the fork's TimeSformer defines it
# (src/meant/timesformer_pytorch.py:89-259, src/utils/rotary.py).  state_dict keys follow the reference:
# to_patch_embedding.*, cls_token, frame_rot_emb.inv_freqs, image_rot_emb.scales, layers.i.{0,1}.{norm.*, fn.to_qkv.weight,
# fn.to_out.0.*}, layers.i.2.{norm.*, fn.net.0.*, fn.net.3.*}, to_out.{0,1}.*.
class _TSAttention(nn.Module):
    """One half of the divided pair (timesformer_pytorch.py:89-148).  On the device end to end: one projection GEMM for all
    tokens, ops.divided_attention (row gathers by index tables, rotary with an identity row for the cls position, the
    flash attention core on the regrouped buffer, the cls query's attention as its own kernel), output projection with the
    block's `+ x` in the GEMM epilogue."""

    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        inner = dim_head * heads
        self.to_qkv = Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(Linear(inner, dim), nn.Identity())

    def forward(self, x, plan, tables, group_mask=None, cls_mask=None, residual=None, drop_p=0.0):
        qkv = self.to_qkv(x)                                             # [b, 1 + f n, 3 inner]
        out = ops.divided_attention(qkv, plan, tables, self.heads, self.scale, group_mask, cls_mask)
        if drop_p > 0.0:                                                 # Dropout after to_out (:101) sits before the block's `+ x`
            return ops.dropout(self.to_out[0](out), drop_p, _seed()) + residual
        return self.to_out[0](out, residual=residual)                    # the block's `+ x` rides the GEMM epilogue


class _TSPreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn, self.norm = fn, LayerNorm(dim)


class _TSPreTokenShift(nn.Module):
    """PreTokenShift (timesformer_pytorch.py:34-53): shifts thirds of the features of the patch tokens one frame forward /
    backward in time before calling `fn`; keeps the reference's module nesting (state_dict keys `...fn.fn.*`)"""

    def __init__(self, frames, fn):
        super().__init__()
        self.frames, self.fn = frames, fn

    def forward(self, x, *args, **kw):
        return self.fn(ops.token_shift(x, self.frames, (x.shape[1] - 1) // self.frames), *args, **kw)


class _TSFeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(Linear(dim, dim * mult * 2), nn.Identity(), nn.Identity(), Linear(dim * mult, dim))

    def forward(self, x, residual=None, drop_p=0.0):
        h = ops.geglu(self.net[0](x))
        if drop_p > 0.0:                                                 # Dropout between GEGLU and the second Linear (:70)
            h = ops.dropout(h, drop_p, _seed())
        return self.net[3](h, residual=residual)


class TimeSformer(nn.Module):
    """src/meant/timesformer_pytorch.py:152-259 with rotary embeddings and no token shift (the variants the fork's models
    use, src/meant/meant_vision.py:130-162), with the optional frame mask.  `meant_forward(video, mask)` returns all
    tokens [b, 1 + f n, dim], `forward` the class logits of the cls token."""

    def __init__(self, *, dim, num_frames, num_classes, image_size=224, patch_size=16, channels=3, depth=12, heads=8, dim_head=64,
                 attn_dropout=0., ff_dropout=0., rotary_emb=True, shift_tokens=False):
        super().__init__()
        self.use_rotary_emb = bool(rotary_emb)
        self.shift_tokens, self.attn_dropout, self.ff_dropout = bool(shift_tokens), float(attn_dropout), float(ff_dropout)
        assert image_size % patch_size == 0, "Image dimensions must be divisible by the patch size."
        self.heads, self.patch_size, self.dim_head, self.num_frames = heads, patch_size, dim_head, num_frames
        self.to_patch_embedding = Linear(channels * patch_size ** 2, dim)
        self.cls_token = nn.Parameter(torch.randn(1, dim))
        if rotary_emb:
            self.frame_rot_emb = nn.Module()
            self.frame_rot_emb.register_buffer("inv_freqs", 1.0 / (10000 ** (torch.arange(0, dim_head, 2).float() / dim_head)))
            self.image_rot_emb = nn.Module()
            self.image_rot_emb.register_buffer("scales", torch.logspace(0., math.log(10 / 2) / math.log(2), dim_head // 4, base=2))
        else:                                            # learned positions added to the tokens (:186, :220-221)
            self.pos_emb = nn.Embedding(num_frames * (image_size // patch_size) ** 2 + 1, dim)
        wrap = (lambda fn: _TSPreTokenShift(num_frames, fn)) if shift_tokens else (lambda fn: fn)      # :196-199
        self.layers = nn.ModuleList([nn.ModuleList([_TSPreNorm(dim, wrap(_TSAttention(dim, dim_head, heads))),
                                                    _TSPreNorm(dim, wrap(_TSAttention(dim, dim_head, heads))),
                                                    _TSPreNorm(dim, wrap(_TSFeedForward(dim)))]) for _ in range(depth)])
        self.to_out = nn.Sequential(LayerNorm(dim), Linear(dim, num_classes))
        self._cache = {}

    def _plan(self, b, f, hp, wp, device):
        """rotary tables with an identity row for the cls position, and the int32 index tables of both regroupings for a
        batch of b videos: index [G, S] (token of group g, position s; position 0 = cls), idx_in [b G S] (rows of the
        [b L] token matrix that make up the regrouped buffer), idx_out [b L] (row of the regrouped buffer that holds
        token t's output; -1 for the cls token, whose output comes from its own kernel), idx_dog [b G S] (= idx_in with
        -1 at position 0: the group attention's output at the cls position is dropped)"""
        key = (b, f, hp, wp, str(device))
        hit = self._cache.get(key)
        if hit is not None:
            return hit
        n = hp * wp
        L = 1 + f * n
        with torch.no_grad():
            fr = ang = None
            if self.use_rotary_emb:
                inv = self.frame_rot_emb.inv_freqs.detach().float().cpu()
                fr = torch.arange(f).float()[:, None] * inv[None, :]
                fr = torch.cat((fr, fr), dim=-1)                                             # [f, Dh] (src/utils/rotary.py:57-60)
                sc = self.image_rot_emb.scales.detach().float().cpu()[None, :]
                hs = torch.linspace(-1., 1., hp)[:, None] * sc * math.pi
                ws = torch.linspace(-1., 1., wp)[:, None] * sc * math.pi
                ang = torch.cat((hs[:, None, :].expand(hp, wp, -1), ws[None, :, :].expand(hp, wp, -1)), dim=-1).reshape(n, -1)
                ang = ang.repeat_interleave(2, dim=-1)                                        # [n, Dh] (:46-48)

            def tabs(a):
                if a is None:
                    return None
                cos = torch.cat((torch.ones(1, a.shape[1]), a.cos())).contiguous().to(device)   # row 0: cls, not rotated
                sin = torch.cat((torch.zeros(1, a.shape[1]), a.sin())).contiguous().to(device)
                return (cos, sin, cos, sin)

            def tables_of(index):                                                             # index [G, S] long
                G, S = index.shape
                boff = (torch.arange(b) * L)[:, None, None]
                idx_in = (boff + index[None]).reshape(-1)
                idx_dog = (boff + index[None]).clone()
                idx_dog[:, :, 0] = -1
                pos = torch.full((L,), -1, dtype=torch.long)
                pos[index[:, 1:].reshape(-1)] = (torch.arange(G)[:, None] * S + torch.arange(1, S)[None, :]).reshape(-1)
                idx_out = (torch.arange(b) * (G * S))[:, None] + pos[None, :]
                idx_out[:, 0] = -1
                i32 = lambda t: t.to(torch.int32).contiguous().to(device)
                return (i32(index), i32(idx_in), i32(idx_out.reshape(-1)), i32(idx_dog.reshape(-1)))
            tok = 1 + torch.arange(f * n).view(f, n)
            zero = torch.zeros(1, dtype=torch.long)
            idx_time = torch.stack([torch.cat((zero, tok[:, j])) for j in range(n)])           # [n, 1 + f]
            idx_space = torch.stack([torch.cat((zero, tok[i, :])) for i in range(f)])          # [f, 1 + n]
            # cls concatenation at the entry: x[b, 0] = cls (-2: the fill row), x[b, 1 + t] = tokens[b, t]; and its inverse
            cat_fwd = torch.cat((torch.full((b, 1), -2, dtype=torch.long), torch.arange(b * f * n).view(b, f * n)), dim=1)
            cat_bwd = (torch.arange(b) * L)[:, None] + 1 + torch.arange(f * n)[None, :]
            plan = (tabs(fr), tabs(ang), tables_of(idx_time), tables_of(idx_space),
                    cat_fwd.reshape(-1).to(torch.int32).to(device), cat_bwd.reshape(-1).to(torch.int32).to(device))
        self._cache = {key: plan}
        return plan

    def meant_forward(self, video, mask=None):
        """mask: optional bool [b, f], False = frame absent (timesformer_pytorch.py:241-253): hidden from the time
        attention's keys and from the cls query; the space attention masks only the cls query"""
        b, f, c, hh, ww = video.shape
        p = self.patch_size
        assert hh % p == 0 and ww % p == 0, f"height {hh} and width {ww} of video must be divisible by the patch size {p}"
        hp, wp = hh // p, ww // p
        n = hp * wp
        dt = resolve_compute_dtype(self, video)
        tokens = self.to_patch_embedding(ops.patchify(video.reshape(b * f, c, hh, ww), p, dt)).view(b, f * n, -1)
        t_time, t_space, p_time, p_space, cat_fwd, cat_bwd = self._plan(b, f, hp, wp, video.device)
        x = ops.cls_concat(self.cls_token, tokens, cat_fwd, cat_bwd)
        if not self.use_rotary_emb:
            x = ops.add_rowvec(x, self.pos_emb.weight[:x.shape[1]].view(1, x.shape[1], -1))  # x += pos_emb(arange(L)), :220-221
        time_mask = cls_mask = None
        if mask is not None:
            m = mask.to(video.device).float()
            one = torch.ones(b, 1, device=video.device)
            time_mask = torch.cat((one, m), dim=1).repeat_interleave(n, dim=0)               # [(b n), 1 + f]: every patch position's group
            cls_mask = torch.cat((one, m.repeat_interleave(n, dim=1)), dim=1)               # [b, 1 + f n]
        pa = self.attn_dropout if self.training else 0.0
        pf = self.ff_dropout if self.training else 0.0
        for ta, sa, ff in self.layers:                   # .fn is the block itself, or PreTokenShift around it (:196-199)
            x = ta.fn(ta.norm(x), p_time, t_time, time_mask, cls_mask, residual=x, drop_p=pa)
            x = sa.fn(sa.norm(x), p_space, t_space, None, cls_mask, residual=x, drop_p=pa)
            x = ff.fn(ff.norm(x), residual=x, drop_p=pf)
        return x

    def forward(self, video, mask=None):
        return self.to_out(self.meant_forward(video, mask=mask)[:, 0])
