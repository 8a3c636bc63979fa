"""CPU oracle for the MEANT encoder hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU, autograd) restatement of the arithmetic of
biirving/meant's multimodal encoder forward pass.  It exists so that the hand-written
HIP path in ``meant_amd/`` has something independent to be checked against on a GPU
box where the reference itself is not present.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
    may import this module;  nothing under ``meant_amd/`` does;
  * it is never the thing that is measured as the product and never a fallback.

Pinning: every class below is checked (tests/test_oracle_golden.py) against golden
vectors that ``oracle/gen_golden.py`` produced by importing the reference's own
modules from /root/reference in the build container (fixtures in tests/golden/).
Parity vs the vendored ``meant/rotary_embedding_torch.py`` is therefore pinned;
parity vs pip ``rotary-embedding-torch==0.5.3`` (not in the tree) is unpinned.

Every function cites the reference file:line whose behaviour it restates.  The code
is written from the formulas, not transcribed: no einops, no
rotary library, explicit index arithmetic.  ``state_dict`` keys match the reference
so one set of weights drives the reference, the oracle and the HIP modules.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1  RMSNorm                                         reference: utils/rms_norm.py:17-57
# --------------------------------------------------------------------------------------
class RMSNorm(nn.Module):
    """y = scale * x / (||x||_2 / sqrt(d) + eps); eps is added to the RMS, outside the
    square root (utils/rms_norm.py:52-53).  Only the p<0, bias=False form is on the
    hot path (utils/rms_norm.py:41-43,57).  The partial form (0 <= p <= 1: the statistics are those of the
    first int(d * p) elements only, :44-50) and the bias form (a learned `offset` added to the result, :35-37,
    :54-55) are restated too, under the reference's parameter names."""

    def __init__(self, d: int, p: float = -1., eps: float = 1e-8, bias: bool = False):
        super().__init__()
        self.d, self.p, self.eps, self.bias = d, p, eps, bias
        self.scale = nn.Parameter(torch.ones(d))
        if bias:
            self.offset = nn.Parameter(torch.zeros(d))

    def forward(self, x):
        if self.p < 0. or self.p > 1.:
            part, d_x = x, self.d
        else:
            d_x = int(self.d * self.p)
            part = x[..., :d_x]
        l2 = torch.sqrt(torch.sum(part * part, dim=-1, keepdim=True))
        rms = l2 * (d_x ** -0.5)
        y = self.scale * (x / (rms + self.eps))
        return y + self.offset if self.bias else y


# --------------------------------------------------------------------------------------
# a2-a5  rotary / xPos tables        reference: meant/rotary_embedding_torch.py:58-147
# --------------------------------------------------------------------------------------
class RotaryTable(nn.Module):
    """Holds ``freqs`` (a frozen Parameter, :85) and, for xPos, the ``scale`` buffer
    (:90-92) under the reference's state_dict names."""

    def __init__(self, dim: int, kind: str, use_xpos: bool = False,
                 theta: float = 10000.0, max_freq: float = 10.0, scale_base: float = 512.0):
        super().__init__()
        half = dim // 2
        if kind == "lang":      # :75   theta^-(2j/dim)
            j2 = torch.arange(0, dim, 2)[:half].float()
            freqs = 1.0 / (theta ** (j2 / dim))
        elif kind == "pixel":   # :77   pi * linspace(1, max_freq/2, dim//2)
            freqs = torch.linspace(1.0, max_freq / 2, half) * math.pi
        else:
            raise ValueError(kind)
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self.use_xpos = use_xpos
        self.scale_base = scale_base
        if use_xpos:            # :90-92
            self.register_buffer("scale", (torch.arange(0, dim, 2) + 0.4 * dim) / (1.4 * dim))
        else:
            self.scale = None

    @property
    def rot_dim(self) -> int:
        return 2 * self.freqs.numel()

    def cos_sin(self, seq_len: int):
        """(S, rot_dim) cos / sin with each frequency repeated on two adjacent lanes
        (:140-142); angles are formed in fp32 (:141)."""
        pos = torch.arange(seq_len, device=self.freqs.device).to(self.freqs.dtype)
        ang = pos[:, None] * self.freqs[None, :]                  # (S, R/2)
        ang = torch.repeat_interleave(ang, 2, dim=-1)             # lanes (2j, 2j+1) share j
        return ang.cos(), ang.sin()

    def xpos_scale(self, seq_len: int):
        """(S, rot_dim) zeta^((pos - S//2)/scale_base), laid out as two concatenated
        blocks, i.e. lane c uses zeta[c mod R/2] (:121-125) -- not interleaved."""
        pos = torch.arange(seq_len, device=self.freqs.device)
        power = (pos - seq_len // 2) / self.scale_base
        s = self.scale[None, :] ** power[:, None]                 # (S, R/2)
        return torch.cat((s, s), dim=-1)


def rotate_pairs(t, cos, sin, scale=None):
    """apply_rotary_emb + rotate_half (:31-44): on lanes [0, R) of the last axis,
    out[2j] = t[2j] c - t[2j+1] s ; out[2j+1] = t[2j+1] c + t[2j] s (times scale);
    lanes >= R pass through untouched."""
    R = cos.shape[-1]
    head, tail = t[..., :R], t[..., R:]
    even, odd = head[..., 0::2], head[..., 1::2]
    swapped = torch.stack((-odd, even), dim=-1).reshape(head.shape)
    if scale is None:
        out = head * cos + swapped * sin
    else:
        out = head * cos * scale + swapped * sin * scale
    return torch.cat((out, tail), dim=-1)


def _split_heads(x, h):
    b, s, d = x.shape
    return x.view(b, s, h, d // h).permute(0, 2, 1, 3)            # b h s dh


def _merge_heads(x):
    b, h, s, dh = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, s, h * dh)


# --------------------------------------------------------------------------------------
# a6  spatial attention over patches                 reference: meant/attention.py:14-62
# --------------------------------------------------------------------------------------
class attention(nn.Module):
    """K comes from the Linear called ``v`` and V from the one called ``k`` (:36-37);
    the scale is 1/sqrt(Dh*H) = 1/sqrt(dim) (:43); pixel rotary on q and k with raster
    positions 0..N-1 (:39-40); never masked."""

    def __init__(self, num_heads: int, dim: int, pos_emb: RotaryTable):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = dim // num_heads
        self.pos_emb = pos_emb
        self.multi_mad = nn.Linear(self.num_heads * self.Dh, dim)
        self.q = nn.Linear(dim, self.Dh * num_heads)
        self.v = nn.Linear(dim, self.Dh * num_heads)
        self.k = nn.Linear(dim, self.Dh * num_heads)

    def forward(self, x):
        H = self.num_heads
        q, k, v = _split_heads(self.q(x), H), _split_heads(self.v(x), H), _split_heads(self.k(x), H)
        cos, sin = self.pos_emb.cos_sin(x.shape[1])
        q, k = rotate_pairs(q, cos, sin), rotate_pairs(k, cos, sin)
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(self.Dh * H)
        w = torch.softmax(scores, dim=-1)
        return self.multi_mad(_merge_heads(w @ v))


# --------------------------------------------------------------------------------------
# a7  causal xPos text attention                 reference: meant/xPosAttention.py:13-66
# --------------------------------------------------------------------------------------
class xPosAttention(nn.Module):
    """As ``attention`` with xPos rotation (q * zeta^p, k * zeta^-p, :39), an
    always-on causal -inf mask (:43-50) and an additive (1-mask)*-1e9 key padding
    term (:54-56)."""

    def __init__(self, num_heads: int, dim: int, xPos: RotaryTable):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = dim // num_heads
        self.xPos = xPos
        self.multi_mad = nn.Linear(self.num_heads * self.Dh, dim)
        self.q = nn.Linear(dim, self.Dh * num_heads)
        self.v = nn.Linear(dim, self.Dh * num_heads)
        self.k = nn.Linear(dim, self.Dh * num_heads)

    def forward(self, x, attention_mask: Optional[torch.Tensor] = None):
        H, S = self.num_heads, x.shape[1]
        q, k, v = _split_heads(self.q(x), H), _split_heads(self.v(x), H), _split_heads(self.k(x), H)
        cos, sin = self.xPos.cos_sin(S)
        zeta = self.xPos.xpos_scale(S)
        q = rotate_pairs(q, cos, sin, zeta)
        k = rotate_pairs(k, cos, sin, zeta ** -1)
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(self.Dh * H)
        future = torch.ones(S, S, dtype=torch.bool, device=x.device).triu(1)
        scores = scores.masked_fill(future, float("-inf"))
        if attention_mask is not None:
            scores = scores + (1 - attention_mask[:, None, None, :]) * -1e9
        w = torch.softmax(scores, dim=-1)
        return self.multi_mad(_merge_heads(w @ v))


# --------------------------------------------------------------------------------------
# a8  temporal (lag-axis) attention                   reference: meant/temporal.py:15-60
# --------------------------------------------------------------------------------------
class temporal(nn.Module):
    """Single query = last lag step (:39); keys/values = all L steps; no rotary, no
    mask; same k/v naming swap and 1/sqrt(dim) scale (:38-39,44)."""

    def __init__(self, num_heads: int, dim: int):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        self.Dh = dim // num_heads
        self.atten_size = self.Dh * num_heads
        self.multi_mad = nn.Linear(self.atten_size, dim)
        self.q = nn.Linear(dim, self.atten_size)
        self.v = nn.Linear(dim, self.atten_size)
        self.k = nn.Linear(dim, self.atten_size)

    def forward(self, x):
        H = self.num_heads
        q = _split_heads(self.q(x[:, -1:, :]), H)                 # b h 1 dh
        k, v = _split_heads(self.v(x), H), _split_heads(self.k(x), H)
        scores = (q @ k.transpose(-1, -2)) / math.sqrt(self.Dh * H)
        w = torch.softmax(scores, dim=-1)
        return self.multi_mad(_merge_heads(w @ v))                # b 1 dim


# --------------------------------------------------------------------------------------
# a9-a11  encoder blocks                               reference: meant/meant.py:35-145
# --------------------------------------------------------------------------------------
class visionEncoder(nn.Module):
    """meant/meant.py:35-75: [RMSNorm, Linear, attention, RMSNorm, Linear] + residual,
    then [RMSNorm, Linear, GELU(erf), RMSNorm, Linear] + residual."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.posEmbed = RotaryTable(math.floor(dim / num_heads / 2), "pixel")      # :46-48
        self.encode = nn.ModuleList([RMSNorm(dim), nn.Linear(dim, dim),
                                     attention(num_heads, dim, self.posEmbed),
                                     RMSNorm(dim), nn.Linear(dim, dim)])
        self.encode2 = nn.ModuleList([RMSNorm(dim), nn.Linear(dim, dim), nn.GELU(),
                                      RMSNorm(dim), nn.Linear(dim, dim)])

    def forward(self, x):
        h = x
        for m in self.encode:
            h = m(h)
        x1 = h + x
        h = x1
        for m in self.encode2:
            h = m(h)
        return h + x1


class languageEncoder(nn.Module):
    """meant/meant.py:78-120.  Indices follow the reference's ModuleLists, which hold
    Dropout(0.0) at encode[4] and Dropout(0.5) at encode2[4]; the oracle is an
    eval-mode oracle, so both are identities here (kept as nn.Identity to preserve the
    state_dict numbering)."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.xPos = RotaryTable(48, "lang", use_xpos=True)                         # :88-92
        self.encode = nn.ModuleList([RMSNorm(dim), nn.Linear(dim, dim),
                                     xPosAttention(num_heads, dim, self.xPos),
                                     RMSNorm(dim), nn.Identity(), nn.Linear(dim, dim)])
        self.encode2 = nn.ModuleList([RMSNorm(dim), nn.Linear(dim, dim), nn.GELU(),
                                      RMSNorm(dim), nn.Identity(), nn.Linear(dim, dim)])

    def forward(self, x, attention_mask=None):
        h = x
        for m in self.encode:
            h = m(h, attention_mask) if isinstance(m, xPosAttention) else m(h)
        x1 = h + x
        h = x1
        for m in self.encode2:
            h = m(h)
        return h + x1


class temporalEncoder(nn.Module):
    """meant/meant.py:124-145 (norms=True) and the norm-less variants of
    meant/meant_vision.py:79-105 / meant/meant_tweet.py:85-110 (norms=False)."""

    def __init__(self, dim: int, num_heads: int, lag: int, norms: bool = True):
        super().__init__()
        self.temp_embedding = nn.Parameter(torch.randn(1, lag, dim))
        if norms:
            mods = [RMSNorm(dim), nn.Linear(dim, dim), temporal(num_heads, dim),
                    RMSNorm(dim), nn.Linear(dim, dim)]
        else:
            mods = [nn.Linear(dim, dim), temporal(num_heads, dim), nn.Linear(dim, dim)]
        self.temp_encode = nn.ModuleList(mods)

    def forward(self, x):
        x = x + self.temp_embedding                                # :141-142 (broadcast over b)
        for m in self.temp_encode:
            x = m(x)
        return x


def patchify(images, p: int):
    """einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (meant/meant.py:194): channel is
    the fastest-varying index inside a patch vector."""
    b, c, H, W = images.shape
    x = images.view(b, c, H // p, p, W // p, p)                   # b c h p1 w p2
    x = x.permute(0, 2, 4, 3, 5, 1)                               # b h w p1 p2 c
    return x.reshape(b, (H // p) * (W // p), p * p * c)


class _PatchEmbed(nn.Sequential):
    """Index 0 is the parameter-free rearrange, index 1 the Linear, so the key is
    ``patchEmbed.1.weight`` as in the reference (meant/meant.py:193-195)."""

    def __init__(self, patch_dim, dim, p):
        super().__init__(nn.Identity(), nn.Linear(patch_dim, dim))
        self.p = p

    def forward(self, images):
        return self[1](patchify(images, self.p))


# --------------------------------------------------------------------------------------
# a12  full model                                     reference: meant/meant.py:148-238
# --------------------------------------------------------------------------------------
class meant(nn.Module):
    def __init__(self, text_dim, image_dim, price_dim, height, width, patch_res, lag,
                 num_classes, embedding, flash=False, num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.lag = lag
        self.dim = text_dim + image_dim
        self.embedding = nn.ModuleList([embedding])
        self.patchEmbed = _PatchEmbed(channels * patch_res * patch_res, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads) for _ in range(num_encoders)])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag)])
        self.mlpHead = nn.ModuleList([RMSNorm(self.dim), nn.Linear(self.dim, num_classes), nn.Sigmoid()])

    def forward(self, tweets, images, attention_mask=None):
        B, L = images.shape[0], self.lag
        words = tweets.reshape(B * L, tweets.shape[2])                              # :209
        for m in self.embedding:
            words = m(words)
        if attention_mask is not None:
            attention_mask = attention_mask.reshape(B * L, attention_mask.shape[2])  # :215
        for enc in self.languageEncoders:
            words = enc(words, attention_mask)
        words = words.view(B, L, words.shape[1], words.shape[2])
        img = images.reshape(B * L, *images.shape[2:])                               # :223
        img = self.patchEmbed(img)
        for enc in self.visionEncoders:
            img = enc(img)
        img = img.view(B, L, img.shape[1], img.shape[2])
        fused = torch.cat((words.mean(dim=2), img.mean(dim=2)), dim=2)               # :231
        for enc in self.temporal_encoding:
            fused = enc(fused)
        for m in self.mlpHead:
            fused = m(fused)
        return fused.squeeze(dim=1)


# --------------------------------------------------------------------------------------
# a13  meant_vision                             reference: meant/meant_vision.py:107-165
# --------------------------------------------------------------------------------------
class meant_vision(nn.Module):
    def __init__(self, image_dim, price_dim, height, width, patch_res, lag, num_classes,
                 flash=False, num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.dim = image_dim
        self.patchEmbed = _PatchEmbed(channels * patch_res * patch_res, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag, norms=False)])
        self.mlpHead = nn.ModuleList([nn.LayerNorm(self.dim), nn.Linear(self.dim, num_classes), nn.Sigmoid()])

    def forward(self, images):
        B, L = images.shape[0], images.shape[1]
        img = self.patchEmbed(images.reshape(B * L, *images.shape[2:]))
        for enc in self.visionEncoders:
            img = enc(img)
        fused = img.view(B, L, img.shape[1], img.shape[2]).mean(dim=2)
        for enc in self.temporal_encoding:
            fused = enc(fused)
        for m in self.mlpHead:
            fused = m(fused)
        return fused.squeeze(dim=1)


# --------------------------------------------------------------------------------------
# a14  meant_tweet                               reference: meant/meant_tweet.py:114-167
# --------------------------------------------------------------------------------------
class meant_tweet(nn.Module):
    """The reference's own languageEncoder copy ends in a NameError
    (meant/meant_tweet.py:81); the evident intent (== meant/meant.py:109-120) is what is
    restated, and what gen_golden.py patches in on the reference side."""

    def __init__(self, text_dim, price_dim, lag, num_classes, embedding, flash=False,
                 num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.dim, self.lag = text_dim, lag
        self.embedding = nn.ModuleList([embedding])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads) for _ in range(num_encoders)])
        self.temporal_encoding = nn.ModuleList([temporalEncoder(self.dim, num_heads, lag, norms=False)])
        self.mlpHead = nn.ModuleList([nn.LayerNorm(self.dim), nn.Linear(self.dim, num_classes), nn.Sigmoid()])

    def forward(self, tweets, attention_mask=None):
        B, L = tweets.shape[0], self.lag
        words = tweets.reshape(B * L, tweets.shape[2])
        attention_mask = attention_mask.reshape(B * L, attention_mask.shape[2])       # :150 (required)
        for m in self.embedding:
            words = m(words)
        for enc in self.languageEncoders:
            words = enc(words, attention_mask)
        fused = words.view(B, L, words.shape[1], words.shape[2]).mean(dim=2)
        for enc in self.temporal_encoding:
            fused = enc(fused)
        for m in self.mlpHead:
            fused = m(fused)
        return fused.squeeze(dim=1)


# --------------------------------------------------------------------------------------
# a15  meant_vqa                                   reference: meant/meant_vqa.py:143-234
# --------------------------------------------------------------------------------------
class meant_vqa(nn.Module):
    """No lag axis and no cross-attention in the forward that actually runs
    (:205-234): concat of the two mean-pools -> RMSNorm -> Linear -> Sigmoid.  The
    ``multimodal_embedding`` / ``multimodal_encoding`` blocks are constructed (so their
    weights are in the state_dict, :199-200) and never called."""

    def __init__(self, text_dim, image_dim, price_dim, height, width, patch_res, lag,
                 num_classes, embedding, flash=False, num_heads=8, num_encoders=1, channels=4):
        super().__init__()
        self.dim = text_dim + image_dim
        self.embedding = nn.ModuleList([embedding])
        self.patchEmbed = _PatchEmbed(channels * patch_res * patch_res, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads) for _ in range(num_encoders)])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads) for _ in range(num_encoders)])
        self.multimodal_embedding = nn.Sequential(nn.Linear(1, self.dim), nn.GELU(), RMSNorm(self.dim),
                                                  nn.Linear(self.dim, self.dim))
        self.multimodal_encoding = nn.ModuleList([visionEncoder(self.dim, num_heads)])
        self.mlpHead = nn.ModuleList([RMSNorm(self.dim), nn.Linear(self.dim, num_classes), nn.Sigmoid()])

    def forward(self, tweets, images, attention_mask=None):
        words = tweets
        for m in self.embedding:
            words = m(words)
        for enc in self.languageEncoders:
            words = enc(words, attention_mask)
        img = self.patchEmbed(images)
        for enc in self.visionEncoders:
            img = enc(img)
        fused = torch.cat((words.mean(dim=1), img.mean(dim=1)), dim=1)
        for m in self.mlpHead:
            fused = m(fused)
        return fused


# --------------------------------------------------------------------------------------
# deterministic weights / inputs shared by gen_golden.py, the tests and bench.py
# (SURVEY.md section 8c "fixture recipe": numpy RandomState only, no torch RNG)
# --------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------
# a16 / 8f-4  divided space-time attention          reference: src/meant/timesformer_pytorch.py
# --------------------------------------------------------------------------------------
def _rot_pairs_full(t, sin, cos):
    """src/utils/rotary.py:7-19: t * cos + rotate_every_two(t) * sin on the first sin.shape[-1] lanes (adjacent pairs
    (x0, x1) -> (-x1, x0)), remaining lanes untouched"""
    r = sin.shape[-1]
    a, rest = t[..., :r], t[..., r:]
    x0, x1 = a[..., 0::2], a[..., 1::2]
    rot = torch.stack((-x1, x0), dim=-1).flatten(-2)
    return torch.cat((a * cos + rot * sin, rest), dim=-1)


class _TSAttention(nn.Module):
    """src/meant/timesformer_pytorch.py:89-148.  One attention of the divided pair: the cls token attends to every
    token (:124); the patch tokens are regrouped along time ('(b n) f d') or space ('(b f) n d'), rotated (:130-131,
    cls excluded), and attend within their group to [cls, group] (:134-141)."""

    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        self.heads, self.scale = heads, dim_head ** -0.5
        inner = dim_head * heads
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Identity())      # Dropout(0.) at index 1 in the reference

    def forward(self, x, mode, f, n, rot, mask=None, cls_mask=None):
        """mask: bool [b, 1 + f] over a time group's keys (cls, frames) -- the time attention only (:250, :256); cls_mask: bool
        [b, 1 + f n] over the keys of the cls query (:252-253, :256-257); masked scores are filled with -max (:82-84)"""
        b, _, _ = x.shape
        h = self.heads
        q, k, v = self.to_qkv(x).chunk(3, dim=-1)
        split = lambda t: t.reshape(b, -1, h, t.shape[-1] // h).permute(0, 2, 1, 3)       # b h tokens d
        q, k, v = split(q) * self.scale, split(k), split(v)
        sim0 = q[:, :, :1] @ k.transpose(-1, -2)                                             # b h 1 tokens
        if cls_mask is not None:
            sim0 = sim0.masked_fill(~cls_mask[:, None, None, :], -torch.finfo(sim0.dtype).max)
        cls_out = torch.softmax(sim0, dim=-1) @ v                                            # b h 1 d
        d = q.shape[-1]

        def group(t):                                                                        # patch tokens -> groups
            t = t[:, :, 1:].reshape(b, h, f, n, d)
            return t.permute(0, 1, 3, 2, 4) if mode == "time" else t                         # b h n f d | b h f n d
        qg, kg, vg = group(q), group(k), group(v)
        sin, cos = rot
        qg, kg = _rot_pairs_full(qg, sin, cos), _rot_pairs_full(kg, sin, cos)
        G = qg.shape[2]
        ck = k[:, :, :1].unsqueeze(2).expand(b, h, G, 1, d)
        cv = v[:, :, :1].unsqueeze(2).expand(b, h, G, 1, d)
        kg, vg = torch.cat((ck, kg), dim=3), torch.cat((cv, vg), dim=3)
        sim = qg @ kg.transpose(-1, -2)                                                      # b h G L L
        if mask is not None:
            sim = sim.masked_fill(~mask[:, None, None, None, :], -torch.finfo(sim.dtype).max)
        og = torch.softmax(sim, dim=-1) @ vg                                                 # b h G L d
        if mode == "time":
            og = og.permute(0, 1, 3, 2, 4)                                                   # b h f n d
        out = torch.cat((cls_out, og.reshape(b, h, f * n, d)), dim=2)
        return self.to_out(out.permute(0, 2, 1, 3).reshape(b, 1 + f * n, h * d))


class _TSPreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn, self.norm = fn, nn.LayerNorm(dim)


class _TSFeedForward(nn.Module):
    """:66-77: Linear(d, 8d) -> GEGLU (x * gelu(gates), :60-63) -> Dropout(0) -> Linear(4d, d)"""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim * mult * 2), nn.Identity(), nn.Identity(), nn.Linear(dim * mult, dim))

    def forward(self, x):
        a, g = self.net[0](x).chunk(2, dim=-1)
        return self.net[3](a * F.gelu(g))


class _TSPreTokenShift(nn.Module):
    """src/meant/timesformer_pytorch.py:28-53: of the patch tokens [b, f, n, d] the first d // 3 features are taken from the
    next frame (shift -1: F.pad drops the first frame and appends zeros), the second third is kept, the third third comes
    from the previous frame (shift +1); features past 3 * (d // 3) and the cls token pass through"""

    def __init__(self, frames, fn):
        super().__init__()
        self.frames, self.fn = frames, fn

    def forward(self, x, *args, **kw):
        f, d = self.frames, x.shape[-1]
        cls_x, t = x[:, :1], x[:, 1:]
        b = t.shape[0]
        t = t.reshape(b, f, -1, d)
        c = d // 3
        z = torch.zeros_like(t[:, :1, :, :c])
        nxt = torch.cat((t[:, 1:, :, :c], z), dim=1)                    # out[frame] = in[frame + 1]
        prv = torch.cat((z, t[:, :-1, :, 2 * c:3 * c]), dim=1)          # out[frame] = in[frame - 1]
        t = torch.cat((nxt, t[..., c:2 * c], prv, t[..., 3 * c:]), dim=-1).reshape(b, -1, d)
        return self.fn(torch.cat((cls_x, t), dim=1), *args, **kw)


class TimeSformer(nn.Module):
    """src/meant/timesformer_pytorch.py:152-259 with rotary_emb=True, shift_tokens=False (what the fork's callers use,
    src/meant/meant_vision.py:130-162), with the optional frame mask.  Frame rotary: angle = frame * 10000^(-2j/Dh) laid out
    cat(freqs, freqs) (src/utils/rotary.py:51-62); axial rotary: logspace(0, log2(max_freq/2), Dh/4, base 2) * pi *
    linspace(-1, 1) along h then w, each angle repeated on a lane pair (:21-49); both rotate adjacent pairs."""

    def __init__(self, *, dim, num_frames, num_classes, image_size=224, patch_size=16, channels=3, depth=12, heads=8, dim_head=64,
                 shift_tokens=False, rotary_emb=True):
        super().__init__()
        self.heads, self.patch_size, self.dim_head = heads, patch_size, dim_head
        self.use_rotary_emb = rotary_emb
        self.to_patch_embedding = nn.Linear(channels * patch_size ** 2, dim)
        self.cls_token = nn.Parameter(torch.randn(1, dim))
        if rotary_emb:
            self.frame_rot_emb = nn.Module()
            self.frame_rot_emb.register_buffer("inv_freqs", 1.0 / (10000 ** (torch.arange(0, dim_head, 2).float() / dim_head)))
            self.image_rot_emb = nn.Module()
            self.image_rot_emb.register_buffer("scales", torch.logspace(0., math.log(10 / 2) / math.log(2), dim_head // 4, base=2))
        else:
            self.pos_emb = nn.Embedding(num_frames * (image_size // patch_size) ** 2 + 1, dim)           # :186
        wrap = (lambda fn: _TSPreTokenShift(num_frames, fn)) if shift_tokens else (lambda fn: fn)      # :196-199
        self.layers = nn.ModuleList([nn.ModuleList([_TSPreNorm(dim, wrap(_TSAttention(dim, dim_head, heads))),
                                                    _TSPreNorm(dim, wrap(_TSAttention(dim, dim_head, heads))),
                                                    _TSPreNorm(dim, wrap(_TSFeedForward(dim)))]) for _ in range(depth)])
        self.to_out = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))

    def rotary_tables(self, f, hp, wp):
        fr = torch.arange(f).float()[:, None] * self.frame_rot_emb.inv_freqs[None, :]
        fr = torch.cat((fr, fr), dim=-1)                                                      # [f, Dh]
        sc = self.image_rot_emb.scales[None, :]
        hs = torch.linspace(-1., 1., hp)[:, None] * sc * math.pi
        ws = torch.linspace(-1., 1., wp)[:, None] * sc * math.pi
        ang = torch.cat((hs[:, None, :].expand(hp, wp, -1), ws[None, :, :].expand(hp, wp, -1)), dim=-1).reshape(hp * wp, -1)
        ang = ang.repeat_interleave(2, dim=-1)                                                # [n, Dh]
        return (fr.sin(), fr.cos()), (ang.sin(), ang.cos())

    def meant_forward(self, video, mask=None):
        """mask: optional bool [b, f], False = frame absent (:241-253)"""
        b, f, c, hh, ww = video.shape
        p = self.patch_size
        hp, wp = hh // p, ww // p
        n = hp * wp
        tok = video.reshape(b, f, c, hp, p, wp, p).permute(0, 1, 3, 5, 4, 6, 2).reshape(b, f * n, p * p * c)   # (p1 p2 c)
        x = torch.cat((self.cls_token[None].expand(b, -1, -1), self.to_patch_embedding(tok)), dim=1)
        if self.use_rotary_emb:
            frame_rot, image_rot = self.rotary_tables(f, hp, wp)
        else:
            x = x + self.pos_emb(torch.arange(x.shape[1]))                                         # :220-221
            ident = (torch.zeros(1, 2), torch.ones(1, 2))                                            # sin 0, cos 1: no rotation
            frame_rot = image_rot = ident
        frame_mask = cls_mask = None
        if mask is not None:
            one = torch.ones(b, 1, dtype=torch.bool)
            frame_mask = torch.cat((one, mask), dim=1)                                            # [b, 1 + f]
            cls_mask = torch.cat((one, mask.repeat_interleave(n, dim=1)), dim=1)                  # [b, 1 + f n] ('b f -> b (f n)')
        for ta, sa, ff in self.layers:
            x = ta.fn(ta.norm(x), "time", f, n, frame_rot, mask=frame_mask, cls_mask=cls_mask) + x
            x = sa.fn(sa.norm(x), "space", f, n, image_rot, cls_mask=cls_mask) + x
            x = ff.fn(ff.norm(x)) + x
        return x

    def forward(self, video, mask=None):
        return self.to_out(self.meant_forward(video, mask=mask)[:, 0])


class meant_language_pretrainer(nn.Module):
    """pretrain_mlm.py:74-88 -- the MLM pretrainer: a caller-supplied embedding module (the reference passes HF
    `RobertaForMaskedLM(...).roberta.embeddings`, :318), `num_encoders` languageEncoders (default num_heads = 8) and a
    caller-supplied vocabulary head (HF `lm_head`, :319); returns the head's logits.  `mlm_input_dim` and `lag` are
    accepted and unused, as in the reference.  The loss is nn.CrossEntropyLoss() over logits.view(-1, V) with the
    -100 labels ignored (:160,178)."""

    def __init__(self, num_encoders, mlm_input_dim, embedding, lm_head, flash=False, lag=5, text_dim=768, num_heads=8):
        super().__init__()
        self.embedding = nn.ModuleList([embedding])
        self.languageEncoders = nn.ModuleList([languageEncoder(text_dim, num_heads) for _ in range(num_encoders)])
        self.mlm_head = lm_head
        self.lag = lag

    def forward(self, words, attention_mask):
        for mod in self.embedding:
            words = mod(words)
        for enc in self.languageEncoders:
            words = enc(words, attention_mask=attention_mask)
        return self.mlm_head(words)


class meant_vision_pretrainer(nn.Module):
    """pretrain_mim.py:77-99 -- masked-image-modelling pretrainer: patchify + Linear, ONE visionEncoder (the
    reference ignores `num_encoders`, :86), tokens reshaped to a (B, d, sqrt(n), sqrt(n)) feature map (:95-98) and a
    caller-supplied decoder (the reference passes HF `ViTForMaskedImageModeling(...).decoder`, :338-339: 1x1 conv to
    stride^2 * 3 channels + PixelShuffle).  Loss: nn.L1Loss() against the first 3 channels of the target image (:162,204)."""

    def __init__(self, num_encoders, decoder, mlm_input_dim, patch_res=16, channels=4, height=224, width=224, image_dim=768, num_heads=8):
        super().__init__()
        self.channels = channels
        self.patch_dim = channels * patch_res * patch_res
        self.n = int((height * width) / (patch_res ** 2))
        self.patchEmbed = _PatchEmbed(self.patch_dim, image_dim, patch_res)
        self.visionEncoders = nn.ModuleList([visionEncoder(image_dim, num_heads)])
        self.decoder = decoder

    def forward(self, images):
        x = self.patchEmbed(images)
        for enc in self.visionEncoders:
            x = enc(x)
        b, n, c = x.shape
        hw = math.floor(n ** 0.5)
        return self.decoder(x.permute(0, 2, 1).reshape(b, c, hw, hw))


def mim_decoder(hidden=128, stride=16, image=32):
    """the decoder the reference's MIM driver takes from HF ViTForMaskedImageModeling (pretrain_mim.py:338), tiny config"""
    from transformers import ViTConfig, ViTForMaskedImageModeling
    cfg = ViTConfig(hidden_size=hidden, num_hidden_layers=1, num_attention_heads=2, intermediate_size=2 * hidden, image_size=image,
                    patch_size=stride, num_channels=3, encoder_stride=stride)
    return ViTForMaskedImageModeling._from_config(cfg).decoder


def mlm_parts(vocab=120, hidden=128, max_pos=40):
    """the embedding module and vocabulary head the reference's MLM driver hands to its pretrainer
    (pretrain_mlm.py:297-319: RobertaForMaskedLM._from_config -> .roberta.embeddings / .lm_head), tiny config"""
    from transformers import RobertaConfig, RobertaForMaskedLM
    cfg = RobertaConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=1, num_attention_heads=2, intermediate_size=2 * hidden,
                        max_position_embeddings=max_pos, pad_token_id=1, type_vocab_size=1, layer_norm_eps=1e-5,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    rob = RobertaForMaskedLM._from_config(cfg)
    return rob.roberta.embeddings, rob.lm_head



def fill_weights_(model: nn.Module, seed: int = 1234) -> nn.Module:
    import numpy as np
    rs = np.random.RandomState(seed)
    sd = model.state_dict()
    with torch.no_grad():
        for key in sorted(sd):
            t = sd[key]
            if key.endswith("freqs") or key.endswith("xPos.scale") or key.endswith("pos_emb.scale"):
                continue                                            # analytic tables
            if not t.is_floating_point():
                continue
            z = torch.from_numpy(rs.standard_normal(tuple(t.shape)).astype("float32"))
            if key.endswith(".scale") or (key.endswith(".weight") and t.dim() == 1):
                v = 1.0 + 0.1 * z                                   # RMSNorm / LayerNorm gains
            elif key.endswith("bias"):
                v = 0.02 * z
            elif key.endswith("temp_embedding"):
                v = 0.5 * z
            else:
                v = z / math.sqrt(t.shape[-1])
            t.copy_(v.to(t.dtype))
    return model


def cross_entropy_on_probs(out, target):
    """in_loop_train.py:232 -- CrossEntropyLoss applied to the already-Sigmoid-ed output."""
    return F.cross_entropy(out, target)
