"""GPU parity of each HIP op (through the C ABI) against the CPU oracle / golden vectors."""
import math

import numpy as np
import pytest
import torch

from tests.util import DTYPES, IDS, TOL, t, assert_close, assert_grad_close, pair, compare_param_grads, norm_floor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def M():
    import meant_amd
    return meant_amd


@pytest.fixture(scope="module")
def O():
    from oracle import meant_oracle
    return meant_oracle


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("rows,d", [(51, 768), (7, 1536), (130, 128), (3, 256), (1000, 768)])
def test_rmsnorm(M, O, dev, golden, dtype, rows, d):
    rs = np.random.RandomState(rows + d)
    x = t(rs.standard_normal((rows, d)).astype("float32"))
    g = t((1 + 0.1 * rs.standard_normal(d)).astype("float32"))
    dy = t(rs.standard_normal((rows, d)).astype("float32"))
    ref = O.RMSNorm(d)
    hip = M.RMSNorm(d).to(dev)
    with torch.no_grad():
        ref.scale.copy_(g)
        hip.scale.copy_(g)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    yr.backward(dy)
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh)
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert yh.dtype == dtype
    assert_close(yh, yr, tol["out"] * 4, "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    assert_grad_close(hip.scale.grad, ref.scale.grad, tol["gelem"], "dscale")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_rmsnorm_partial_and_bias_forms_golden(M, dev, golden, dtype, tag):
    """meant_amd.RMSNorm(d, p, bias) -- the forms of utils/rms_norm.py:44-57 that no MEANT model constructs -- against the reference
    class's own outputs and gradients (fixture from oracle/gen_golden.py), both tiers; state_dict carries `offset` like the reference"""
    g = golden("rmsnorm_partial_bias")
    d, p, bias = int(g[f"{tag}_cfg"][0]), float(g[f"{tag}_cfg"][1]), bool(g[f"{tag}_cfg"][2])
    hip = M.RMSNorm(d, p=p, bias=bias).to(dev)
    assert set(hip.state_dict()) == ({"scale", "offset"} if bias else {"scale"})
    with torch.no_grad():
        hip.scale.copy_(t(g[f"{tag}_scale"]))
        if bias:
            hip.offset.copy_(t(g[f"{tag}_offset"]))
    x = t(g[f"{tag}_x"]).to(dev).to(dtype).requires_grad_()
    y = hip(x)
    y.backward(t(g[f"{tag}_dy"]).to(dev).to(dtype))
    tol = TOL[dtype]
    exact = dtype == torch.float32
    assert_close(y, t(g[f"{tag}_y"]), 1e-5 if exact else tol["out"] * 4, "y")
    assert_grad_close(x.grad, t(g[f"{tag}_dx"]), 1e-5 if exact else tol["gelem"], "dx")
    assert_grad_close(hip.scale.grad, t(g[f"{tag}_dscale"]), 1e-4 if exact else tol["gelem"], "dscale")
    if bias:
        assert_grad_close(hip.offset.grad, t(g[f"{tag}_doffset"]), 1e-5 if exact else tol["gelem"], "doffset")


def test_rmsnorm_golden(M, dev, golden):
    g = golden("rmsnorm_768")
    hip = M.RMSNorm(768).to(dev)
    with torch.no_grad():
        hip.scale.copy_(t(g["scale"]))
    x = t(g["x"]).to(dev).requires_grad_()
    y = hip(x)
    y.backward(t(g["dy"]).to(dev))
    assert_close(y, t(g["y"]), 1e-5, "y")
    assert_close(x.grad, t(g["dx"]), 1e-5, "dx")
    assert_grad_close(hip.scale.grad, t(g["dscale"]), 1e-4, "dscale")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_rmsnorm_dropout_is_consistent(M, dev, dtype):
    """train-mode fused dropout: forward mask == backward mask, keep-rate ~ 1-p, scaling 1/(1-p)."""
    from meant_amd import ops
    x = torch.randn(256, 768, device=dev).to(dtype).requires_grad_()
    scale = torch.ones(768, device=dev, requires_grad=True)
    y = ops.rmsnorm(x, scale, 1e-8, 0.5, 1234)
    y0 = ops.rmsnorm(x, scale, 1e-8, 0.0, 0)
    kept = (y != 0)
    rate = kept.float().mean().item()
    assert 0.48 < rate < 0.52
    assert_close(y[kept], 2 * y0[kept], 2e-2 if dtype == torch.bfloat16 else 1e-5, "scaled")
    y.backward(torch.ones_like(y))
    # gradient flows only through kept elements: d/dscale_j = sum_rows mask * 2 * x_j * r
    gs = (kept.float() * 2 * (y0.detach().float())).sum(0)
    assert_grad_close(scale.grad, gs, 2e-2 if dtype == torch.bfloat16 else 1e-4, "dscale")
    y2 = ops.rmsnorm(x, scale, 1e-8, 0.5, 1234)
    assert torch.equal(y, y2)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_layernorm(M, dev, dtype):
    rs = np.random.RandomState(5)
    x = t(rs.standard_normal((9, 768)).astype("float32"))
    dy = t(rs.standard_normal((9, 768)).astype("float32"))
    ref = torch.nn.LayerNorm(768)
    hip = M.LayerNorm(768)
    with torch.no_grad():
        ref.weight.copy_(t((1 + 0.1 * rs.standard_normal(768)).astype("float32")))
        ref.bias.copy_(t((0.1 * rs.standard_normal(768)).astype("float32")))
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    xr = x.clone().requires_grad_()
    ref(xr).backward(dy)
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh)
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(yh, ref(x), tol["out"] * 4, "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    assert_grad_close(hip.weight.grad, ref.weight.grad, tol["gelem"], "dgamma")
    assert_grad_close(hip.bias.grad, ref.bias.grad, tol["gelem"], "dbeta")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("M_,N,K", [(300, 768, 768), (64, 2304, 768), (1000, 128, 1024), (5, 2, 1536), (257, 3129, 256),
                                     (128, 128, 128), (4097, 768, 768), (2048, 768, 256), (1536, 512, 128), (1300, 256, 128)])
@pytest.mark.parametrize("epi", ["none", "gelu", "residual", "sigmoid"])
def test_linear(M, dev, dtype, M_, N, K, epi):
    from meant_amd import ops
    from meant_amd._lib import EPI_NONE, EPI_GELU, EPI_SIGMOID
    if epi != "none" and (M_, N, K) not in [(300, 768, 768), (5, 2, 1536), (1000, 128, 1024), (2048, 768, 256), (1300, 256, 128)]:
        pytest.skip("epilogues are covered on four shapes")
    rs = np.random.RandomState(M_ + N)
    x = t(rs.standard_normal((M_, K)).astype("float32"))
    w = t((rs.standard_normal((N, K)) / math.sqrt(K)).astype("float32"))
    b = t((0.1 * rs.standard_normal(N)).astype("float32"))
    res = t(rs.standard_normal((M_, N)).astype("float32"))
    dy = t(rs.standard_normal((M_, N)).astype("float32"))
    # reference in fp32 on CPU, on the dtype-rounded inputs
    xq, wq, resq, dyq = [v.to(dtype).float() for v in (x, w, res, dy)]
    xr, wr, br, rr = xq.clone().requires_grad_(), wq.clone().requires_grad_(), b.clone().requires_grad_(), resq.clone().requires_grad_()
    yr = torch.nn.functional.linear(xr, wr, br)
    if epi == "gelu":
        yr = torch.nn.functional.gelu(yr)
    elif epi == "sigmoid":
        yr = torch.sigmoid(yr)
    elif epi == "residual":
        yr = yr + rr
    yr.backward(dyq)
    xh = x.to(dev).to(dtype).requires_grad_()
    wh = wq.to(dev).requires_grad_()          # fp32 master weights that are exactly representable in `dtype`
    bh = b.to(dev).requires_grad_()
    rh = res.to(dev).to(dtype).requires_grad_()
    e = {"none": EPI_NONE, "gelu": EPI_GELU, "sigmoid": EPI_SIGMOID, "residual": EPI_NONE}[epi]
    yh = ops.linear(xh, wh, bh, rh if epi == "residual" else None, e)
    yh.backward(dy.to(dev).to(dtype))
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert yh.dtype == dtype and yh.shape == (M_, N)
    assert_close(yh, yr, tol * max(1.0, yr.abs().max().item()) , "y")
    gt = 1e-3 if dtype == torch.float32 else 2e-2
    assert_grad_close(xh.grad, xr.grad, gt, "dx")
    assert_grad_close(wh.grad, wr.grad, gt, "dw")
    assert_grad_close(bh.grad, br.grad, gt, "db")
    if epi == "residual":
        assert_grad_close(rh.grad, rr.grad, gt, "dres")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_rotary_against_golden(M, dev, golden, dtype):
    """rotary kernel on a packed q|k|v buffer vs vectors produced by the reference's rotary library"""
    from meant_amd import ops
    from meant_amd._lib import lib, check
    g = golden("rotary_xpos48")
    rot = M.RotaryEmbedding(dim=48, use_xpos=True)
    for S in (16, 64, 512):
        q, k = t(g[f"q{S}"]), t(g[f"k{S}"])                      # [1, 2, S, 64] (b h s dh)
        H, Dh = 2, 64
        qkv = torch.zeros(S, 3 * H * Dh)
        qkv[:, : H * Dh] = q[0].permute(1, 0, 2).reshape(S, H * Dh)
        qkv[:, H * Dh: 2 * H * Dh] = k[0].permute(1, 0, 2).reshape(S, H * Dh)
        qkv[:, 2 * H * Dh:] = 7.0
        buf = qkv.to(dev).to(dtype).contiguous()
        orig = buf.clone()
        qa, qb, ka, kb = rot.tables(S, dev)
        dt = 0 if dtype == torch.float32 else 1
        check(lib.meant_rotary_qk(buf.data_ptr(), S, S, H, Dh, 48, qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(),
                                  0, dt, torch.cuda.current_stream().cuda_stream))
        rq = buf[:, : H * Dh].float().cpu().view(S, H, Dh).permute(1, 0, 2)
        rk = buf[:, H * Dh: 2 * H * Dh].float().cpu().view(S, H, Dh).permute(1, 0, 2)
        tol = 2e-5 if dtype == torch.float32 else 4e-2
        assert_close(rq, t(g[f"rq{S}"])[0], tol, f"q S={S}")
        assert_close(rk, t(g[f"rk{S}"])[0], tol, f"k S={S}")
        assert torch.equal(buf[:, 2 * H * Dh:], orig[:, 2 * H * Dh:])            # v untouched
        # adjoint: <R x, y> == <x, R^T y>
        y = torch.randn(buf.shape, generator=torch.Generator().manual_seed(S)).to(dev).to(dtype)
        yt = y.clone()
        check(lib.meant_rotary_qk(yt.data_ptr(), S, S, H, Dh, 48, qa.data_ptr(), qb.data_ptr(), ka.data_ptr(), kb.data_ptr(),
                                  1, dt, torch.cuda.current_stream().cuda_stream))
        terms = (buf.double() * y.double())[:, : 2 * H * Dh]
        lhs = terms.sum().item()
        rhs = (orig.double() * yt.double())[:, : 2 * H * Dh].sum().item()
        # both sides carry one rounding to `dtype` per element (the rotated x, the adjoint-rotated y): independent errors of
        # relative size eps add up like a random walk over the terms
        eps = 2.0 ** -23 if dtype == torch.float32 else 2.0 ** -8
        assert abs(lhs - rhs) <= 6 * eps * terms.pow(2).sum().sqrt().item() + 1e-6


def _attn_pair(M, O, kind, H, d, dev):
    if kind == "pixel":
        ref = O.attention(H, d, O.RotaryTable(math.floor(d / H / 2), "pixel"))
        hip = M.attention(H, d, M.RotaryEmbedding(dim=math.floor(d / H / 2), freqs_for="pixel"))
    else:
        ref = O.xPosAttention(H, d, O.RotaryTable(48, "lang", use_xpos=True))
        hip = M.xPosAttention(H, d, M.RotaryEmbedding(dim=48, use_xpos=True))
    return pair(ref, hip, 4321, dev)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("name,kind", [("attention_h2_d128_n196", "pixel"), ("xposattention_h2_d128_s80", "xpos"),
                                       ("xposattention_h2_d128_s512", "xpos")])
def test_attention_modules_golden(M, O, dev, golden, dtype, name, kind):
    g = golden(name)
    ref, hip = _attn_pair(M, O, kind, 2, 128, dev)
    x = t(g["x"]).to(dev).to(dtype).requires_grad_()
    args = (t(g["mask"]).to(dev),) if kind == "xpos" else ()
    y = hip(x, *args)
    y.backward(t(g["dy"]).to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(y, t(g["y"]), tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(x.grad, t(g["dx"]), tol["gelem"], "dx")
    params = dict(hip.named_parameters())
    floor = norm_floor(g["grad_norms"], dtype)
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(nm)]
        a = p.grad.double().norm().item()
        assert abs(a - refn) <= tol["gnorm"] * max(refn, floor), (nm, a, refn)
        if refn > floor:
            ga = t(g["grad__" + str(nm)])
            got = p.grad if p.grad.numel() <= 4096 else p.grad[:4]
            assert_grad_close(got, ga, tol["gelem"] * 2, str(nm))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("kind,G,S,H,d", [("pixel", 3, 196, 12, 768), ("xpos", 2, 512, 12, 768), ("xpos", 5, 64, 2, 128),
                                          ("pixel", 2, 4, 2, 128), ("xpos", 3, 100, 4, 256), ("xpos", 2, 1, 2, 128),
                                          ("pixel", 1, 300, 2, 128),
                                          # head dims 96 (the reference's default 8 heads: native), 128 (native), 80 (padded to 128
                                          # in the bf16 tier)
                                          ("xpos", 2, 512, 8, 768), ("pixel", 3, 196, 8, 768), ("xpos", 3, 200, 2, 256),
                                          ("pixel", 2, 130, 1, 128), ("xpos", 2, 100, 4, 320),
                                          # long sequences: 18 key tiles, and more than 64 (the per-group tile masks no longer fit two
                                          # 64-bit words: the kernels read the per-tile flags instead)
                                          ("xpos", 2, 1100, 2, 128), ("xpos", 2, 4200, 2, 128), ("pixel", 1, 4200, 2, 128)])
def test_attention_modules_vs_oracle(M, O, dev, dtype, kind, G, S, H, d):
    """real head geometry incl. ragged lengths (not multiples of any tile), S=1, fully padded rows"""
    ref, hip = _attn_pair(M, O, kind, H, d, dev)
    rs = np.random.RandomState(S + d)
    x = t(rs.standard_normal((G, S, d)).astype("float32"))
    dy = t(rs.standard_normal((G, S, d)).astype("float32"))
    mask = torch.ones(G, S)
    if kind == "xpos" and S > 1:
        mask[0, S // 2:] = 0
        mask[G - 1, :] = 0           # everything padded: softmax falls back to uniform over the causal window
    xq, dyq = x.to(dtype).float(), dy.to(dtype).float()
    xr = xq.clone().requires_grad_()
    yr = ref(xr, mask) if kind == "xpos" else ref(xr)
    yr.backward(dyq)
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh, mask.to(dev)) if kind == "xpos" else hip(xh)
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(yh, yr, tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    compare_param_grads(ref, hip, dtype, kind)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("kind,G,S,H,d", [("pixel", 2, 100, 2, 128), ("xpos", 3, 96, 2, 128), ("xpos", 2, 300, 4, 384)])
def test_attention_with_learned_rotary_frequencies(M, O, dev, dtype, kind, G, S, H, d):
    """RotaryEmbedding(learned_freq=True) (rotary_embedding_torch.py:67,85): `freqs` trains.  The module then takes the unfused
    route (projection GEMM, rotation as tensor ops, attention core without tables); outputs, input gradient, every parameter
    gradient AND d loss / d freqs against the oracle, whose tables are plain differentiable functions of its `freqs`."""
    if kind == "pixel":
        ref = O.attention(H, d, O.RotaryTable(math.floor(d / H / 2), "pixel"))
        hip = M.attention(H, d, M.RotaryEmbedding(dim=math.floor(d / H / 2), freqs_for="pixel", learned_freq=True))
    else:
        ref = O.xPosAttention(H, d, O.RotaryTable(48, "lang", use_xpos=True))
        hip = M.xPosAttention(H, d, M.RotaryEmbedding(dim=48, use_xpos=True, learned_freq=True))
    ref, hip = pair(ref, hip, 977, dev)
    rot_ref = ref.pos_emb if kind == "pixel" else ref.xPos
    rot_hip = hip.pos_emb if kind == "pixel" else hip.xPos
    rot_ref.freqs.requires_grad_(True)
    assert rot_hip.freqs.requires_grad and rot_hip.learned_freq
    rs = np.random.RandomState(S + d)
    x = t(rs.standard_normal((G, S, d)).astype("float32"))
    dy = t(rs.standard_normal((G, S, d)).astype("float32"))
    mask = torch.ones(G, S)
    if kind == "xpos":
        mask[0, S // 2:] = 0
    xr = x.to(dtype).float().clone().requires_grad_()
    yr = ref(xr, mask) if kind == "xpos" else ref(xr)
    yr.backward(dy.to(dtype).float())
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh, mask.to(dev)) if kind == "xpos" else hip(xh)
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(yh, yr, tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    compare_param_grads(ref, hip, dtype, kind + " learned_freq")
    gf, gr = rot_hip.freqs.grad, rot_ref.freqs.grad
    assert gf is not None and gr is not None and gr.abs().max().item() > 0
    assert_grad_close(gf, gr, tol["gelem"], "d freqs")


def _hash_uniform_np(seed: int, idx):
    """common.h:hash_uniform on the host: splitmix-style hash of (seed, index) -> [0, 1), in the kernel's float arithmetic"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("G,S,H,d,p", [(3, 80, 2, 128, 0.25), (2, 200, 4, 256, 0.1), (2, 64, 2, 192, 0.5)])
def test_xpos_attention_with_dropout_on_the_scores(M, O, dev, dtype, G, S, H, d, p):
    """meant/xPosAttention.py:59: `scores = self.dropout(scores)` -- after the causal fill and the key-padding term, before the softmax
    (p = 0 in every reference model).  The HIP path (materialised fp32 core) against the same arithmetic in eager fp32 on the host with
    the SAME mask, rebuilt from the kernel's counter-based hash: a dropped score is 0 (a dropped FUTURE position becomes visible),
    a kept one is divided by 1 - p.  Outputs, input gradient and every parameter gradient."""
    ref = O.xPosAttention(H, d, O.RotaryTable(48, "lang", use_xpos=True))
    hip = M.xPosAttention(H, d, M.RotaryEmbedding(dim=48, use_xpos=True), droput=p)
    ref, hip = pair(ref, hip, 611, dev)
    hip.train()
    rs = np.random.RandomState(S + d)
    x = t(rs.standard_normal((G, S, d)).astype("float32"))
    dy = t(rs.standard_normal((G, S, d)).astype("float32"))
    mask = torch.ones(G, S)
    mask[0, S // 2:] = 0
    # the module draws ONE seed per forward from torch's CPU generator (modules._seed)
    torch.manual_seed(4242)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    idx = np.arange(G * H * S * S, dtype=np.uint64).reshape(G, H, S, S)
    keep = torch.from_numpy(_hash_uniform_np(seed, idx) >= np.float32(p))
    assert 0.5 * (1 - p) < keep.float().mean().item() < min(1.0, 1.5 * (1 - p))

    def reference(xin):
        Dh = d // H
        q, k, v = (O._split_heads(f(xin), H) for f in (ref.q, ref.v, ref.k))          # the Linear called `v` makes the keys
        cos, sin = ref.xPos.cos_sin(S)
        zeta = ref.xPos.xpos_scale(S)
        q, k = O.rotate_pairs(q, cos, sin, zeta), O.rotate_pairs(k, cos, sin, zeta ** -1)
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(Dh * H)
        sc = sc.masked_fill(torch.ones(S, S, dtype=torch.bool).triu(1), float("-inf"))
        sc = sc + (1 - mask[:, None, None, :]) * -1e9
        sc = torch.where(keep, sc / (1 - p), torch.zeros_like(sc))                      # nn.Dropout on the scores, this mask
        return ref.multi_mad(O._merge_heads(torch.softmax(sc, dim=-1) @ v))

    xr = x.to(dtype).float().clone().requires_grad_()
    yr = reference(xr)
    yr.backward(dy.to(dtype).float())
    torch.manual_seed(4242)
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh, mask.to(dev))
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(yh, yr, tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    compare_param_grads(ref, hip, dtype, "xpos score dropout")
    hip.eval()                                          # eval mode: no dropout, the fused path
    ye = hip(xh.detach(), mask.to(dev))
    assert_close(ye, ref(x.to(dtype).float(), mask), tol["out"] * (1 if dtype == torch.float32 else 4), "eval y")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("H,d", [(2, 128), (2, 192), (1, 128)], ids=["dh64", "dh96", "dh128"])
def test_text_attention_padding_patterns(M, O, dev, dtype, H, d):
    """the bf16 kernels skip key tiles that are all padding when key 0 is live; every pattern that does or does not
    qualify (suffix at and off tile boundaries, prefix, hole, one live key, nothing live) must match the oracle"""
    G, S = 7, 320
    ref, hip = _attn_pair(M, O, "xpos", H, d, dev)
    rs = np.random.RandomState(99)
    x = t(rs.standard_normal((G, S, d)).astype("float32"))
    dy = t(rs.standard_normal((G, S, d)).astype("float32"))
    mask = torch.ones(G, S)
    mask[0, 64:] = 0            # suffix on a tile boundary: tiles 1..4 skipped
    mask[1, 131:] = 0           # suffix off a boundary: tile 2 partial, 3..4 skipped
    mask[2, :70] = 0            # prefix: key 0 dead -> nothing may be skipped (rows 0..69 see only padding)
    mask[3, 64:192] = 0         # hole: tiles 1..2 skipped in the middle
    mask[4, 1:] = 0             # one live key
    mask[5, :] = 0              # nothing live
    xq, dyq = x.to(dtype).float(), dy.to(dtype).float()
    xr = xq.clone().requires_grad_()
    yr = ref(xr, mask)
    yr.backward(dyq)
    xh = x.to(dev).to(dtype).requires_grad_()
    yh = hip(xh, mask.to(dev))
    yh.backward(dy.to(dev).to(dtype))
    tol = TOL[dtype]
    assert_close(yh, yr, tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
    compare_param_grads(ref, hip, dtype, "xpos")


def test_head_dim_96_runs_on_the_native_kernels(M, O, dev):
    """the reference's default of 8 heads (meant/meant.py:149: Dh = 96 at d = 768) takes the 96-wide instantiation of the
    flash kernels -- six k-steps for QK^T, three 32-column output blocks, an unpadded [T, 3 * 768] projection buffer --
    not the zero-padded 128-wide one and not the fp32 detour; checked through the library's route counters, with values
    against the oracle on a causal padded case and an unmasked ragged one"""
    from meant_amd import _lib
    for kind, G, S in (("xpos", 3, 200), ("pixel", 2, 196)):
        ref, hip = _attn_pair(M, O, kind, 8, 768, dev)
        rs = np.random.RandomState(S)
        x = t(rs.standard_normal((G, S, 768)).astype("float32"))
        dy = t(rs.standard_normal((G, S, 768)).astype("float32"))
        mask = torch.ones(G, S)
        mask[0, S // 3:] = 0
        xq, dyq = x.bfloat16().float(), dy.bfloat16().float()
        xr = xq.clone().requires_grad_()
        yr = ref(xr, mask) if kind == "xpos" else ref(xr)
        yr.backward(dyq)
        _lib.route_reset()
        xh = x.to(dev).bfloat16().requires_grad_()
        yh = hip(xh, mask.to(dev)) if kind == "xpos" else hip(xh)
        yh.backward(dy.to(dev).bfloat16())
        assert _lib.route_count("attn_fwd_d96") == 1 and _lib.route_count("attn_bwd_d96") == 1
        assert _lib.route_count("attn_fwd_d128") == 0 and _lib.route_count("attn_generic") == 0
        tol = TOL[torch.bfloat16]
        assert_close(yh, yr, tol["out"] * 4, "y")
        assert_grad_close(xh.grad, xr.grad, tol["gelem"], "dx")
        compare_param_grads(ref, hip, torch.bfloat16, kind)


@pytest.mark.parametrize("kind,G,S,H,d", [("pixel", 5, 13, 12, 768), ("xpos", 6, 13, 2, 128), ("xpos", 3, 16, 8, 768), ("pixel", 3, 16, 2, 256),
                                          ("xpos", 4, 7, 1, 128), ("pixel", 9, 2, 3, 192)])
def test_short_sequences_take_the_one_wave_kernels(M, O, dev, kind, G, S, H, d):
    """S <= 16 (the time half of the divided space-time attention: 1 + 12 tokens per group) runs on attn_short.hip, one wave
    per (group, head): values against the oracle (causal + padding incl. a fully padded group, and unmasked), head dims 64 /
    96 / 128, and against the tiled flash kernels on the same inputs (MEANT_OPT attn_short = 0); the route counters say which ran"""
    from meant_amd import _lib
    ref, hip = _attn_pair(M, O, kind, H, d, dev)
    rs = np.random.RandomState(17 * S + d)
    x = t(rs.standard_normal((G, S, d)).astype("float32"))
    dy = t(rs.standard_normal((G, S, d)).astype("float32"))
    mask = torch.ones(G, S)
    if kind == "xpos" and S > 1:
        mask[0, S // 2:] = 0
        mask[1, 1:S - 1] = 0
        mask[G - 1, :] = 0
    xq, dyq = x.bfloat16().float(), dy.bfloat16().float()
    xr = xq.clone().requires_grad_()
    yr = ref(xr, mask) if kind == "xpos" else ref(xr)
    yr.backward(dyq)

    def run():
        for p_ in hip.parameters():
            p_.grad = None
        xh = x.to(dev).bfloat16().requires_grad_()
        yh = hip(xh, mask.to(dev)) if kind == "xpos" else hip(xh)
        yh.backward(dy.to(dev).bfloat16())
        return yh.detach().float().cpu(), xh.grad.float().cpu()

    prev = _lib.get_option("attn_short")
    _lib.set_option("attn_short", 1)
    _lib.route_reset()
    y1, g1 = run()
    assert _lib.route_count("attn_short") == 2 and _lib.route_count("attn_fwd") + _lib.route_count("attn_fwd_d96") + _lib.route_count("attn_fwd_d128") == 0
    tol = TOL[torch.bfloat16]
    assert_close(y1, yr, tol["out"] * 4, "y")
    assert_grad_close(g1, xr.grad, tol["gelem"], "dx")
    compare_param_grads(ref, hip, torch.bfloat16, kind)
    _lib.set_option("attn_short", 0)
    try:
        _lib.route_reset()
        y0, g0 = run()
        assert _lib.route_count("attn_short") == 0
    finally:
        _lib.set_option("attn_short", prev)
    assert_close(y1, y0, tol["out"] * 4, "short vs tiled: y")
    assert_grad_close(g1, g0, tol["gelem"], "short vs tiled: dx")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_temporal_golden(M, O, dev, golden, dtype):
    g = golden("temporal_h12_d1536_l12")
    ref, hip = pair(O.temporal(12, 1536), M.temporal(12, 1536), 4321, dev)
    x = t(g["x"]).to(dev).to(dtype).requires_grad_()
    y = hip(x)
    y.backward(t(g["dy"]).to(dev).to(dtype))
    tol = TOL[dtype]
    assert y.shape == (3, 1, 1536)
    assert_close(y, t(g["y"]), tol["out"] * (1 if dtype == torch.float32 else 4), "y")
    assert_grad_close(x.grad, t(g["dx"]), tol["gelem"], "dx")
    params = dict(hip.named_parameters())
    floor = norm_floor(g["grad_norms"], dtype)
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        a = params[str(nm)].grad.double().norm().item()
        assert abs(a - refn) <= tol["gnorm"] * max(refn, floor), (nm, a, refn)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meanpool_patchify_embedding_rowvec(M, O, dev, dtype):
    from meant_amd import ops
    rs = np.random.RandomState(11)
    a = t(rs.standard_normal((6, 37, 128)).astype("float32"))
    b = t(rs.standard_normal((6, 196, 256)).astype("float32"))
    ah, bh = a.to(dev).to(dtype).requires_grad_(), b.to(dev).to(dtype).requires_grad_()
    ref = torch.cat((a.to(dtype).float().mean(1), b.to(dtype).float().mean(1)), dim=1)
    out = ops.meanpool_cat(ah, bh)                       # default: the pooled features stay in the tier's dtype
    assert out.dtype == dtype
    assert_close(out, ref, 1e-5 if dtype == torch.float32 else 4e-3, "meanpool")
    out = ops._MeanPoolCat.apply(ah, bh, torch.float32)  # fp32 tail (ops.TAIL_FP32)
    assert out.dtype == torch.float32
    assert_close(out, ref, 1e-5, "meanpool")
    w = torch.randn_like(out)
    out.backward(w)
    assert_close(ah.grad, (w[:, :128].cpu() / 37)[:, None, :].expand(6, 37, 128), 1e-6 if dtype == torch.float32 else 1e-3, "d meanpool a")
    assert_close(bh.grad, (w[:, 128:].cpu() / 196)[:, None, :].expand(6, 196, 256), 1e-6 if dtype == torch.float32 else 1e-3, "d meanpool b")
    # patchify (exact data movement)
    img = t(rs.standard_normal((3, 4, 32, 48)).astype("float32"))
    got = ops.patchify(img.to(dev), 16, dtype)
    assert_close(got, O.patchify(img, 16).to(dtype).float(), 0.0, "patchify")
    # embedding gather + scatter-add
    table = torch.randn(50, 128)
    ids = torch.randint(0, 50, (4, 9))
    th = table.to(dev).requires_grad_()
    e = ops.embedding(ids.to(dev), th, dtype)
    assert_close(e, table[ids].to(dtype).float(), 0.0, "embedding")
    ge = torch.randn(4, 9, 128)
    e.backward(ge.to(dev).to(dtype))
    tr = table.clone().requires_grad_()
    torch.nn.functional.embedding(ids, tr).backward(ge.to(dtype).float())
    assert_close(th.grad, tr.grad, 1e-5 if dtype == torch.float32 else 1e-2, "d embedding")
    # temp-embedding add
    x = torch.randn(5, 3, 256)
    v = torch.randn(1, 3, 256)
    xh, vh = x.to(dev).to(dtype).requires_grad_(), v.to(dev).requires_grad_()
    y = ops.add_rowvec(xh, vh)
    assert_close(y, x.to(dtype).float() + v, 1e-6 if dtype == torch.float32 else 2e-2, "add_rowvec")
    gy = torch.randn(5, 3, 256)
    y.backward(gy.to(dev).to(dtype))
    assert_close(vh.grad, gy.to(dtype).float().sum(0, keepdim=True), 1e-5 if dtype == torch.float32 else 1e-2, "d temp_embedding")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("raw", ["f64", "f32", "u8"])
def test_patchify_raw_inputs(M, O, dev, dtype, raw):
    """input pipeline: float64 / float32 / uint8 pixels + the data set's global (x - mean) / std
    become patches in one device pass; oracle = normalise on the host in float64 as in_loop_train.py:591-593 does,
    then the oracle's patchify"""
    from meant_amd import ops
    rs = np.random.RandomState(5)
    if raw == "u8":
        img = rs.randint(0, 256, (3, 4, 32, 48)).astype("uint8")
    else:
        img = (rs.standard_normal((3, 4, 32, 48)) * 3 + 1).astype("float64" if raw == "f64" else "float32")
    mean, std = float(img.astype("float64").mean()), float(img.astype("float64").std())
    ref = O.patchify(t(((img.astype("float64") - mean) / std).astype("float32")), 16)
    got = ops.patchify(torch.from_numpy(img).to(dev), 16, dtype, mean, std)
    assert got.dtype == dtype and got.shape == ref.shape
    assert_close(got, ref, 1e-5 if dtype == torch.float32 else 2e-2, "patchify_raw")
    # C != 4 and no normalisation
    img3 = rs.standard_normal((2, 3, 16, 16))
    got3 = ops.patchify(torch.from_numpy(img3).to(dev), 8, dtype)
    assert_close(got3, O.patchify(t(img3.astype("float32")), 8), 1e-6 if dtype == torch.float32 else 2e-2, "patchify_raw c3")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("V,d,B,S", [(300, 768, 3, 2048), (5000, 264, 1, 5003), (64, 1024, 2, 4099)])
def test_embedding_backward_sorted_path(M, dev, dtype, V, d, B, S):
    """large token counts take the sorted scatter-add: heavy duplication (a padding-like hot id whose run crosses many of the
    kernel's 256-entry stretches: float atomics), ids that occur once or not at all (plain read-modify-write), token counts
    that are not a multiple of the stretch, row widths that use one / both / part of the lanes' chunks -- all must sum exactly"""
    from meant_amd import ops
    g = torch.Generator().manual_seed(3)
    table = torch.randn(V, d, generator=g)
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[0, :1500] = 7                                  # one id carries a large share of all tokens
    th = table.to(dev).requires_grad_()
    e = ops.embedding(ids.to(dev), th, dtype)
    ge = torch.randn(B, S, d, generator=g)
    e.backward(ge.to(dev).to(dtype))
    tr = table.clone().requires_grad_()
    torch.nn.functional.embedding(ids, tr).backward(ge.to(dtype).float())
    assert_grad_close(th.grad, tr.grad, 1e-5 if dtype == torch.float32 else 2e-3, "d embedding (sorted)")
    assert th.grad[8:].abs().sum() > 0 and torch.equal(th.grad.cpu() == 0, tr.grad == 0)
    # option "deterministic": runs are summed by the wave in whose stretch they start, no atomics -> bit-identical repeats
    from meant_amd import _lib
    prev = _lib.get_option("deterministic")
    _lib.set_option("deterministic", 1)
    try:
        grads = []
        for _ in range(2):
            t2 = table.to(dev).requires_grad_()
            ops.embedding(ids.to(dev), t2, dtype).backward(ge.to(dev).to(dtype))
            grads.append(t2.grad.clone())
    finally:
        _lib.set_option("deterministic", prev)
    assert torch.equal(grads[0], grads[1])
    assert_grad_close(grads[0], tr.grad, 1e-5 if dtype == torch.float32 else 2e-3, "d embedding (sorted, deterministic)")


def test_errors_are_loud(M, dev):
    from meant_amd import ops
    with pytest.raises(RuntimeError):
        ops.rmsnorm(torch.randn(4, 768), torch.ones(768))            # CPU tensor: no fallback
    with pytest.raises(M.MeantHipError):
        ops.rmsnorm(torch.randn(4, 100, device=dev), torch.ones(100, device=dev))   # d % 8 != 0
    with pytest.raises(TypeError):
        ops.rmsnorm(torch.randn(4, 768, device=dev).half(), torch.ones(768, device=dev))
