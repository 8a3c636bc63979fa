"""The single-pass attention backward (csrc/attn_bwd1.hip: Dh = 64, S <= 256 or causal S <= 512) through the C ABI:
against the two-pass kernels it replaces (option attn_bwd1 = 0), against an fp32 PyTorch evaluation of the reference's
softmax(QK^T * scale + mask) V (meant/attention.py:43-57, meant/xPosAttention.py:41-63), which route a shape takes, and bit-identical
results run after run (the kernel has no atomics: anything else is a race)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16 = 1


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _case(G, S, H, causal, mask_kind, rotary, seed, dev):
    import meant_amd
    Dh, D = 64, 64 * H
    gen = torch.Generator().manual_seed(seed)
    qkv = torch.randn(G * S, 3 * D, generator=gen).to(dev).bfloat16()
    do = torch.randn(G * S, D, generator=gen).to(dev).bfloat16()
    rs = np.random.RandomState(seed)
    m = np.ones((G, S), dtype=np.float32)
    if mask_kind == "suffix":                        # trailing padding of random length, as bench.py draws it
        for g in range(G):
            p = rs.randint(0, max(1, (3 * S) // 4))
            if p:
                m[g, S - p:] = 0
    elif mask_kind == "holes":                       # padding in the middle: dead 64-key tiles between live ones
        for g in range(G):
            a = rs.randint(1, max(2, S // 2))
            b = rs.randint(a, S)
            m[g, a:b] = 0
    elif mask_kind == "upper_dead":                  # nothing alive past key 200: a causal S > 256 item loses its whole upper key half
        m[:, min(200, S - 1):] = 0
    mask = torch.from_numpy(m).to(dev) if mask_kind != "none" else None
    tables = None
    if rotary:
        rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True) if causal else meant_amd.RotaryEmbedding(dim=32, freqs_for="pixel")
        tables = rot.tables(S, dev)
    return qkv, do, mask, tables, 1.0 / math.sqrt(D)


def _run(qkv, do, mask, tables, scale, G, S, H, causal, one_pass):
    from meant_amd import _lib
    from meant_amd._lib import lib, check
    Dh, D = 64, 64 * H
    st = torch.cuda.current_stream().cuda_stream
    o = torch.empty(G * S, D, device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty(G, H, S, 2, device=qkv.device)
    wsb = lib.meant_attn_ws(G, S, H, Dh, BF16)
    ws = torch.empty(max(wsb, 16), device=qkv.device, dtype=torch.uint8)
    mp = mask.data_ptr() if mask is not None else None
    check(lib.meant_attn_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), mp, G, S, H, Dh, scale, causal, BF16, ws.data_ptr(), wsb, st), "attn_fwd")
    dqkv = torch.full_like(qkv, float("nan"))
    tp = [t.data_ptr() for t in tables] if tables is not None else [None] * 4
    R = tables[0].shape[1] if tables is not None else 0
    _lib.set_option("attn_bwd1", 1 if one_pass else 0)
    try:
        _lib.route_reset()
        check(lib.meant_attn_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), mp, dqkv.data_ptr(), G, S, H, Dh, scale, causal,
                                 tp[0], tp[1], tp[2], tp[3], R, BF16, ws.data_ptr(), wsb, st), "attn_bwd")
        torch.cuda.synchronize()
        routes = (_lib.route_count("attn_bwd1"), _lib.route_count("attn_bwd"))
    finally:
        _lib.set_option("attn_bwd1", 1)
    return o, dqkv, routes


SHAPES = [
    # G, S, H, causal, mask, rotary
    (5, 196, 2, 0, "none", True),          # the vision shape of the step, fewer items than workgroups
    (300, 196, 3, 0, "none", True),        # more items than one round of persistent workgroups
    (7, 512, 2, 1, "suffix", True),        # the text shape: two key halves, partial dQ blocks through the scratch buffer
    (40, 512, 4, 1, "suffix", True),
    (6, 512, 2, 1, "upper_dead", True),    # upper half dead: no partial, zeros for its dK / dV
    (6, 512, 2, 1, "holes", False),        # dead tiles between live ones
    (4, 17, 2, 1, "suffix", True), (4, 63, 2, 0, "none", False), (4, 64, 2, 1, "none", True), (4, 65, 2, 1, "suffix", False),
    (3, 130, 2, 1, "suffix", True), (3, 255, 2, 0, "holes", False), (3, 256, 2, 1, "none", True),
    (3, 257, 2, 1, "suffix", True), (3, 300, 2, 1, "holes", True), (3, 384, 2, 1, "suffix", False), (3, 511, 2, 1, "suffix", True),
    (3, 129, 1, 0, "suffix", False),
]


@pytest.mark.parametrize("G,S,H,causal,mask_kind,rotary", SHAPES)
def test_single_pass_equals_two_pass(dev, G, S, H, causal, mask_kind, rotary):
    """same inputs through both backward forms: every shape here must take the single-pass kernel when it is on, and the two
    results agree to the rounding of their bf16 outputs (the sums are formed in a different order, nothing else differs)"""
    case = _case(G, S, H, causal, mask_kind, rotary, 1000 + S + G, dev)
    _, d1, r1 = _run(*case, G, S, H, causal, True)
    _, d2, r2 = _run(*case, G, S, H, causal, False)
    assert r1[0] == 1, f"single-pass route not taken: {r1}"
    assert r2 == (0, 1), f"two-pass route expected with the option off: {r2}"
    assert not torch.isnan(d1).any() and not torch.isnan(d2).any()
    D = 64 * H
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = d1[:, sl].float(), d2[:, sl].float()
        scale_ = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2 ** -7 * max(scale_, 1e-6), name          # two bf16 roundings of the largest value
        assert ((a - b).norm() / max(b.norm().item(), 1e-12)).item() <= 2e-4, name


@pytest.mark.parametrize("G,S,H,causal,mask_kind", [(3, 196, 2, 0, "none"), (3, 512, 2, 1, "suffix"), (2, 300, 2, 1, "holes")])
def test_single_pass_against_fp32_torch(dev, G, S, H, causal, mask_kind):
    """forward + backward of the eager reference in fp32 on the same bf16 inputs (no rotary: the adjoint has its own tests)"""
    qkv, do, mask, _, scale = _case(G, S, H, causal, mask_kind, False, 77 + S, dev)
    o, dqkv, routes = _run(qkv, do, mask, None, scale, G, S, H, causal, True)
    assert routes[0] == 1
    D = 64 * H
    x = qkv.float().view(G, S, 3, H, 64).requires_grad_()
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)          # [G, H, S, 64]
    s = (q @ k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + (1.0 - mask)[:, None, None, :] * -1e9
    if causal:
        s = s.masked_fill(torch.ones(S, S, device=dev, dtype=torch.bool).triu(1), float("-inf"))
    ref = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(G * S, D)
    ref.backward(do.float())
    g = x.grad.view(G * S, 3 * D)
    assert (o.float() - ref.detach()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = dqkv[:, sl].float(), g[:, sl]
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item(), name
        assert abs(a.norm().item() - b.norm().item()) <= 1e-2 * b.norm().item(), name


def test_shapes_outside_its_range_take_two_passes(dev):
    """non-causal sequences longer than 256 (two key halves would need more partial blocks than a wave can park) and anything
    past 512 stay on the dQ + dK/dV pair"""
    for (G, S, H, causal) in ((2, 300, 2, 0), (2, 600, 2, 1)):
        case = _case(G, S, H, causal, "none", False, 5, dev)
        _, d, routes = _run(*case, G, S, H, causal, True)
        assert routes == (0, 1), (S, causal, routes)
        assert not torch.isnan(d).any()


@pytest.mark.parametrize("G,S,H,causal,mask_kind", [(96, 512, 4, 1, "suffix"), (96, 196, 4, 0, "none")])
def test_single_pass_is_bit_reproducible(dev, G, S, H, causal, mask_kind):
    """24 runs on the same inputs, bit-identical outputs.  (A data race shows up here as one 32 x 32 block of dQ that differs now
    and then: the form of the dQ product that kept LDS reads in flight across statement boundaries did exactly that.)"""
    case = _case(G, S, H, causal, mask_kind, True, 4242, dev)
    first = None
    for rep in range(24):
        _, d, routes = _run(*case, G, S, H, causal, True)
        assert routes[0] == 1
        if first is None:
            first = d.clone()
        else:
            ne = (d != first)
            assert not ne.any(), f"run {rep}: {int(ne.sum())} elements differ from the first run"


@pytest.mark.parametrize("G,S,H,causal,mask_kind", [(96, 512, 4, 1, "suffix"), (96, 196, 4, 0, "none")])
def test_forward_and_two_pass_backward_are_bit_reproducible(dev, G, S, H, causal, mask_kind):
    """the same 24-run check on the flash forward and on the dQ + dK/dV pair (their LDS reads are separate asm statements with the
    wait behind them: the pattern that bit the single-pass kernel's dQ product; tools/isa_inflight_check.py looks for it statically)"""
    case = _case(G, S, H, causal, mask_kind, True, 99, dev)
    first = None
    for rep in range(24):
        o, d, routes = _run(*case, G, S, H, causal, False)
        assert routes == (0, 1)
        if first is None:
            first = (o.clone(), d.clone())
        else:
            assert not (o != first[0]).any(), f"run {rep}: forward output differs"
            ne = (d != first[1])
            assert not ne.any(), f"run {rep}: {int(ne.sum())} gradient elements differ from the first run"
