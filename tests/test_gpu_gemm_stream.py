"""Element-level parity of the STREAMING 256x256 NT GEMM (`gemm_bf16_nt256s_kernel`, the kernel that carries the headline
number) and of the 256x256 dW kernel, at shapes that provably route to them with more than one tile per workgroup, the
dynamic tile draw live, XCD stealing, the ragged head + tail split, every epilogue, the rotary instantiation of the fused
q|k|v projection and the vocabulary GEMM of the MLM head.  Reference: fp32 on the CPU on the bf16-rounded inputs
(the arithmetic of meant/meant.py:59-64,101-107 nn.Linear call sites, meant/attention.py:36-40, pretrain_mlm.py:74-89,160).

Every case asserts its ROUTE through the library's launch counters, so that a change of the dispatch thresholds cannot
silently orphan these tests again."""
import math

import numpy as np
import pytest
import torch

from tests.util import assert_close, assert_grad_close

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def L():
    from meant_amd import _lib
    saved = {k: _lib.get_option(k) for k in ("nt_dynamic", "nt_grid_cap", "nt_stream", "deterministic", "nt_ragged")}
    for k, v in (("nt_dynamic", 1), ("nt_grid_cap", 0), ("nt_stream", 1), ("deterministic", 0), ("nt_ragged", 1)):
        _lib.set_option(k, v)                        # the route assertions below are about the default dispatch, whatever the environment says
    _lib.route_reset()
    yield _lib
    for k, v in saved.items():
        _lib.set_option(k, v)


def _rand(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype("float32"))


def _linear_case(L, dev, M_, N, K, epi, dynamic, cap):
    from meant_amd import ops
    from meant_amd._lib import EPI_NONE, EPI_GELU, EPI_SIGMOID
    L.set_option("nt_dynamic", dynamic)
    L.set_option("nt_grid_cap", cap)
    rs = np.random.RandomState(M_ % 1000 + N + K)
    x, w, b = _rand(rs, M_, K), _rand(rs, N, K, scale=1 / math.sqrt(K)), _rand(rs, N, scale=0.1)
    res, dy = _rand(rs, M_, N), _rand(rs, M_, N)
    xq, wq, resq, dyq = [v.to(BF).float() for v in (x, w, res, dy)]
    xr, wr, br, rr = xq.clone().requires_grad_(), wq.clone().requires_grad_(), b.clone().requires_grad_(), resq.clone().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, br)
    if epi == "gelu":
        yr = torch.nn.functional.gelu(yr)
    elif epi == "sigmoid":
        yr = torch.sigmoid(yr)
    elif epi == "residual":
        yr = yr + rr
    yr.backward(dyq)
    xh = x.to(dev).to(BF).requires_grad_()
    wh, bh = wq.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    rh = res.to(dev).to(BF).requires_grad_()
    e = {"none": EPI_NONE, "gelu": EPI_GELU, "sigmoid": EPI_SIGMOID, "residual": EPI_NONE}[epi]
    L.route_reset()
    yh = ops.linear(xh, wh, bh, rh if epi == "residual" else None, e)
    fwd_routes = {r: L.route_count(r) for r in ("nt256s", "nt_split", "nt128", "nt256")}
    yh.backward(dy.to(dev).to(BF))
    torch.cuda.synchronize()
    assert_close(yh, yr, 3e-2 * max(1.0, yr.abs().max().item()), "y")
    assert_grad_close(xh.grad, xr.grad, 2e-2, "dx")
    assert_grad_close(wh.grad, wr.grad, 2e-2, "dw")
    assert_grad_close(bh.grad, br.grad, 2e-2, "db")
    if epi == "residual":
        assert_grad_close(rh.grad, rr.grad, 2e-2, "dres")
    return fwd_routes


# (M, N, K): 36864 rows = 144 row tiles.  N = 768 -> 432 tiles on 256 workgroups; N = 3072 -> 1536 (6 per workgroup);
# K = 3072 -> 48 K-steps per tile; K = 768 -> 12 (the dynamic draw needs >= 8).
@pytest.mark.parametrize("dynamic", [1, 0], ids=["dyn", "fixed"])
@pytest.mark.parametrize("M_,N,K,epi", [
    (36864, 768, 768, "none"), (36864, 768, 768, "residual"), (36864, 768, 768, "sigmoid"),
    (32768, 3072, 768, "gelu"), (32768, 768, 3072, "residual"), (36864, 2304, 768, "none"),
])
def test_streaming_linear(L, dev, M_, N, K, epi, dynamic):
    r = _linear_case(L, dev, M_, N, K, epi, dynamic, 0)
    assert r["nt256s"] == 1 and r["nt128"] == 0 and r["nt256"] == 0, r      # forward took the streaming kernel
    assert L.route_count("nt256s") == 2                                       # ... and so did dX
    assert L.route_count("tn256") == 1                                        # dW on the 256 x 256 TN kernel


@pytest.mark.parametrize("epi", ["none", "residual", "gelu"])
def test_streaming_linear_few_workgroups_and_steals(L, dev, epi):
    """64 workgroups (8 per XCD) on 432 tiles: ~7 tiles per workgroup through the dynamic draw; then the same with
    nt_dynamic = 3, where only XCD 0 draws from its own counter and the tiles of the other seven counters can only be
    handed out by the steal path (gemm_bf16.hip, 'this XCD is dry'): the counter of stolen tiles must move, and the values
    are checked against the oracle in both runs."""
    r = _linear_case(L, dev, 36864, 768, 768, epi, 1, 64)
    assert r["nt256s"] == 1
    before = L.lib.meant_debug_nt_steals()
    r = _linear_case(L, dev, 36864, 768, 768, epi, 3, 64)
    assert r["nt256s"] == 1
    stolen = L.lib.meant_debug_nt_steals() - before
    # forward and dX: 2 x (432 tiles - 64 first tiles - XCD 0's own share of the rest)
    assert stolen >= 2 * (432 - 64) * 3 // 4, f"only {stolen} tiles went through the steal path"


@pytest.mark.parametrize("ragged", [1, 0], ids=["overlap", "split"])
@pytest.mark.parametrize("dynamic", [1, 0], ids=["dyn", "fixed"])
@pytest.mark.parametrize("M_,epi", [(33025, "none"), (33024 + 100, "residual"), (33024 + 255, "gelu")])
def test_streaming_linear_ragged_rows(L, dev, M_, epi, dynamic, ragged):
    """M not a multiple of 256.  Default (nt_ragged = 1): all 130 row tiles on the streaming kernel, the last one moved up to end at
    row M (it recomputes rows of its neighbour bit-identically).  nt_ragged = 0 (what aliasing operands fall back to): head (129
    row tiles) on the streaming kernel, the < 256 trailing rows on the 128 x 128 kernel."""
    L.set_option("nt_ragged", ragged)
    try:
        L.route_reset()
        r = _linear_case(L, dev, M_, 768, 768, epi, dynamic, 0)
        ov = L.route_count("nt_overlap")
    finally:
        L.set_option("nt_ragged", 1)
    if ragged:
        assert r["nt_split"] == 0 and r["nt256s"] == 1 and r["nt128"] == 0 and ov >= 1, (r, ov)
    else:
        assert r["nt_split"] == 1 and r["nt256s"] == 1 and r["nt128"] == 1 and ov == 0, (r, ov)


def test_ragged_rows_are_bit_identical_between_the_two_schemes(L, dev):
    """the overlapped last tile and the head + tail split compute every element from the same operands in the same K order"""
    import meant_amd.ops as ops
    from meant_amd._lib import EPI_GELU
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((33024 + 77, 768)).astype("float32")).to(dev).to(BF)
    w = torch.from_numpy((rs.standard_normal((768, 768)) * 0.05).astype("float32")).to(dev)
    b = torch.from_numpy(rs.standard_normal(768).astype("float32")).to(dev)
    with torch.no_grad():
        y1 = ops.linear(x, w, b, None, EPI_GELU)
        L.set_option("nt_ragged", 0)
        try:
            y0 = ops.linear(x, w, b, None, EPI_GELU)
        finally:
            L.set_option("nt_ragged", 1)
    # rows of the streaming head: same kernel, same arithmetic -> identical bits; the tail rows come from the 128 x 128 kernel in the
    # split scheme (16x16x32 MFMA, different summation tree) and only have to agree to rounding
    assert torch.equal(y1[:33024], y0[:33024])
    assert (y1[33024:].float() - y0[33024:].float()).abs().max().item() <= 2e-2


def test_ragged_rows_with_an_in_place_residual_fall_back_to_the_split(L, dev):
    """y = x W^T + y through the C ABI with residual == y (in place): the overlapped last tile would read rows its neighbour
    has already overwritten, so the launcher must take the head + tail split -- and the result must be right"""
    M_, N, K = 33024 + 60, 768, 768
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.standard_normal((M_, K)).astype("float32")).to(dev).to(BF)
    w = torch.from_numpy((rs.standard_normal((N, K)) * 0.05).astype("float32")).to(dev).to(BF)
    y0 = torch.from_numpy(rs.standard_normal((M_, N)).astype("float32")).to(dev).to(BF)
    y = y0.clone()
    L.route_reset()
    L.check(L.lib.meant_linear_fwd(x.data_ptr(), K, w.data_ptr(), None, y.data_ptr(), N, y.data_ptr(), N, None, M_, N, K, L.EPI_RESIDUAL, 1,
                                   torch.cuda.current_stream().cuda_stream), "linear_fwd")
    torch.cuda.synchronize()
    assert L.route_count("nt_split") == 1 and L.route_count("nt_overlap") == 0
    ref = x.float() @ w.float().t() + y0.float()
    assert_close(y, ref, 3e-2 * max(1.0, ref.abs().max().item()), "in-place residual")


def _rot_ref(t, A, B):
    """out[c] = t[c] A[pos, c] + rot(t)[c] B[pos, c], rot(t)[2j] = -t[2j+1], rot(t)[2j+1] = t[2j]   (include/meant_hip.h)"""
    r = torch.empty_like(t)
    r[..., 0::2] = -t[..., 1::2]
    r[..., 1::2] = t[..., 0::2]
    return t * A + r * B


@pytest.mark.parametrize("dynamic,cap", [(1, 0), (0, 0), (1, 64)], ids=["dyn", "fixed", "dyn-64wg"])
@pytest.mark.parametrize("G,S,H,Dh,R", [(72, 512, 12, 64, 48), (192, 196, 12, 64, 32)])
def test_streaming_qkv_projection_with_rotary(L, dev, G, S, H, Dh, R, dynamic, cap):
    """meant_qkv_proj_fwd through the raw C ABI at M = G*S rows (36864 / 37632 = 144 / 147 row tiles x 9 column tiles):
    the rotary instantiation of the streaming kernel walking several tiles per workgroup, checked element by element."""
    import meant_amd
    from meant_amd._lib import check, BF16
    L.set_option("nt_dynamic", dynamic)
    L.set_option("nt_grid_cap", cap)
    d, D = 768, H * Dh
    M_ = G * S
    rs = np.random.RandomState(G + S)
    x, w, b = _rand(rs, M_, d), _rand(rs, 3 * D, d, scale=1 / math.sqrt(d)), _rand(rs, 3 * D, scale=0.1)
    if S == 512:
        rot = meant_amd.RotaryEmbedding(dim=R, use_xpos=True)
    else:
        rot = meant_amd.RotaryEmbedding(dim=R, freqs_for="pixel")
    qa, qb, ka, kb = rot.tables(S, dev)
    assert qa.shape == (S, R)
    xq, wq = x.to(BF).float(), w.to(BF).float()
    ref = torch.nn.functional.linear(xq, wq, b).view(G, S, 3, H, Dh)
    tabs = [v.cpu()[None, :, None, :] for v in (qa, qb, ka, kb)]
    ref[:, :, 0, :, :R] = _rot_ref(ref[:, :, 0, :, :R].clone(), tabs[0], tabs[1])
    ref[:, :, 1, :, :R] = _rot_ref(ref[:, :, 1, :, :R].clone(), tabs[2], tabs[3])
    ref = ref.reshape(M_, 3 * D)
    xh, wh, bh = x.to(dev).to(BF), w.to(dev).to(BF), b.to(dev)
    out = torch.empty((M_, 3 * D), device=dev, dtype=BF)
    L.route_reset()
    check(L.lib.meant_qkv_proj_fwd(xh.data_ptr(), d, wh.data_ptr(), bh.data_ptr(), out.data_ptr(), M_, d, S, H, Dh, R, qa.data_ptr(),
                                   qb.data_ptr(), ka.data_ptr(), kb.data_ptr(), BF16, torch.cuda.current_stream().cuda_stream), "qkv_proj_fwd")
    torch.cuda.synchronize()
    assert L.route_count("nt256s_rot") >= 1 and L.route_count("nt128") == 0 and L.route_count("nt256") == 0
    assert_close(out, ref, 3e-2 * max(1.0, ref.abs().max().item()), "qkv")
    # q, k and v sections separately, relative to their own scale (the xPos scale shrinks q at late positions)
    for sec, name in enumerate("qkv"):
        a, r_ = out[:, sec * D:(sec + 1) * D], ref[:, sec * D:(sec + 1) * D]
        assert_grad_close(a, r_, 1.5e-2, name)


@pytest.mark.parametrize("dynamic", [1, 0], ids=["dyn", "fixed"])
def test_vocab_gemm_cross_entropy_at_streaming_scale(L, dev, dynamic):
    """pretrain_mlm.py:88,160: logits = h W^T + b over V = 64001 (padded to 64256 = 251 column tiles), mean CE with
    ignore_index -100, at T = 11264 tokens: 44 x 251 = 11044 tiles through the streaming kernel forward, its dX
    (K = 64256: 1004 K-steps per tile) and the 256 x 256 dW kernel, against F.cross_entropy in fp32 on the CPU."""
    from meant_amd import ops
    L.set_option("nt_dynamic", dynamic)
    T, V, d = 11264, 64001, 768
    rs = np.random.RandomState(7)
    x, w, b = _rand(rs, T, d), _rand(rs, V, d, scale=1 / math.sqrt(d)), _rand(rs, V, scale=0.1)
    tgt = torch.from_numpy(rs.randint(0, V, size=T))
    tgt[rs.rand(T) < 0.85] = -100                       # 15 % of the positions carry a label
    tgt[0] = V - 1                                      # the last real row of W next to the padding
    xq, wq = x.to(BF).float(), w.to(BF).float()
    xr, wr, br = xq.clone().requires_grad_(), wq.clone().requires_grad_(), b.clone().requires_grad_()
    # the HIP path rounds the logits to bf16 before the loss (they are the GEMM's output tensor): mirror that
    logits = torch.nn.functional.linear(xr, wr, br)
    lr = torch.nn.functional.cross_entropy(logits, tgt, ignore_index=-100)
    lr.backward()
    xh = x.to(dev).to(BF).requires_grad_()
    wh, bh = torch.nn.Parameter(wq.to(dev)), torch.nn.Parameter(b.to(dev))
    L.route_reset()
    lh = ops.vocab_linear_cross_entropy(xh, wh, bh, tgt.to(dev))
    assert L.route_count("nt256s") == 1 and L.route_count("nt128") == 0
    lh.backward()
    torch.cuda.synchronize()
    assert L.route_count("nt256s") == 2 and L.route_count("tn256") == 1
    assert abs(lh.item() - lr.item()) <= 1e-2 * abs(lr.item()), (lh.item(), lr.item())
    assert wh.grad.shape == (V, d) and bh.grad.shape == (V,)
    assert_grad_close(xh.grad, xr.grad, 2e-2, "dx")
    assert_grad_close(wh.grad, wr.grad, 2e-2, "dW")
    assert_grad_close(bh.grad, br.grad, 2e-2, "db")
    # second call: the padded bf16 / transposed copies of W come from the cache (keyed on the parameter), nothing accumulates
    n0 = len(ops.weights)
    ops.vocab_linear_cross_entropy(xh, wh, bh, tgt.to(dev)).backward()
    assert len(ops.weights) == n0


def test_deterministic_parameter_gradients(L, dev):
    """option "deterministic": dW / dbias by ordered partial sums -> two backward passes are bit-identical
    (with the default float atomics they differ in the last bits)"""
    from meant_amd import ops
    rs = np.random.RandomState(3)
    M_, N, K = 36864, 768, 768
    x, w, b, dy = _rand(rs, M_, K), _rand(rs, N, K, scale=1 / math.sqrt(K)), _rand(rs, N, scale=0.1), _rand(rs, M_, N)
    xh, dyh = x.to(dev).to(BF), dy.to(dev).to(BF)

    def grads():
        wh, bh = w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
        ops.linear(xh, wh, bh).backward(dyh)
        return wh.grad.clone(), bh.grad.clone()

    L.set_option("deterministic", 1)
    L.route_reset()
    g1, g2 = grads(), grads()
    assert L.route_count("tn256_det") == 2 and L.route_count("tn256") == 0
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    L.set_option("deterministic", 0)
    g3 = grads()
    assert L.route_count("tn256") == 1
    assert_grad_close(g3[0], g1[0], 1e-5, "dW atomics vs ordered")
    assert_grad_close(g3[1], g1[1], 1e-5, "db atomics vs ordered")
    # ragged token count (tail rows through the exact generic kernel) and a small shape on the 128 x 128 kernel
    L.set_option("deterministic", 1)
    for (m2, n2, k2) in [(4096 + 37, 768, 768), (1000, 128, 192)]:
        x2, dy2 = _rand(rs, m2, k2).to(dev).to(BF), _rand(rs, m2, n2).to(dev).to(BF)
        w2 = _rand(rs, n2, k2).to(dev)
        outs = []
        for _ in range(2):
            wh, bh = w2.clone().requires_grad_(), torch.zeros(n2, device=dev).requires_grad_()
            ops.linear(x2, wh, bh).backward(dy2)
            outs.append((wh.grad.clone(), bh.grad.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ref_w = dy2.float().cpu().t() @ x2.float().cpu()
        assert_grad_close(outs[0][0], ref_w, 1e-3, "dW deterministic")
