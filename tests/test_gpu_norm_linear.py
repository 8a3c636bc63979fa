"""RMSNorm folded into the Linear that consumes it (meant_linear_fwd_rowscale / meant_rmsnorm_bwd_chain / meant_linear_bwd_dx_norm):
Linear(RMSNorm(x)) of utils/rms_norm.py:40-57 + meant/meant.py:61-64,103-107 without the normalised tensor, against fp32 PyTorch
and against the separate kernels."""
import numpy as np
import pytest
import torch

from util import TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rms(t, gain, eps=1e-8):
    d = t.shape[-1]
    return gain * t / (t.norm(dim=-1, keepdim=True) / d ** 0.5 + eps)


def _mask_of(rows, d, p, seed, dev):
    from meant_amd import ops
    if p == 0:
        return torch.ones(rows, d)
    y = ops.rmsnorm(torch.ones(rows, d, device=dev), torch.ones(d, device=dev), 1e-8, p, seed)
    return ((y != 0).float() / (1.0 - p)).cpu()


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("d,G,S", [(768, 5, 64), (128, 4, 24), (256, 33, 8), (768, 2, 196)])
def test_norm_linear_gelu_norm_against_torch(dev, monkeypatch, d, G, S, p):
    """ops.norm_linear_gelu_norm: (dropout(RMSNorm(gelu(Linear(RMSNorm(x))))), x) with a residual gradient, forward and every
    gradient against an fp32 evaluation with the same dropout mask; and the separate-kernel path agrees with it to the same gates"""
    from meant_amd import ops
    seed = 4242
    gen = torch.Generator().manual_seed(d + S)
    x = torch.randn(G, S, d, generator=gen) * (1 + torch.rand(G, S, 1, generator=gen))      # rows of different norms
    W = torch.randn(d, d, generator=gen) / d ** 0.5
    b = torch.randn(d, generator=gen) * 0.1
    g0 = 1 + 0.1 * torch.randn(d, generator=gen)
    g3 = 1 + 0.1 * torch.randn(d, generator=gen)
    wy = torch.randn(G, S, d, generator=gen)
    wr = torch.randn(G, S, d, generator=gen)
    mask = _mask_of(G * S, d, p, seed, dev).view(G, S, d)
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", True)
    assert ops.norm_linear_ok(x.to(dev).bfloat16(), W)

    xr = x.bfloat16().float().clone().requires_grad_()
    Wr, br, g0r, g3r = (t.clone().requires_grad_() for t in (W, b, g0, g3))
    y_ref = _rms(torch.nn.functional.gelu(_rms(xr, g0r) @ Wr.t() + br), g3r) * mask
    ((y_ref * wy).sum() + (xr * wr).sum()).backward()

    def run(fused):
        xd = x.detach().clone().to(dev).bfloat16().requires_grad_()
        Wd, bd, g0d, g3d = (t.detach().clone().to(dev).requires_grad_() for t in (W, b, g0, g3))
        if fused:
            y, res = ops.norm_linear_gelu_norm(xd, g0d, 1e-8, Wd, bd, g3d, 1e-8, p, seed)
        else:
            n, res = ops.rmsnorm_fork(xd, g0d, 1e-8)
            y = ops.linear_gelu_rmsnorm(n, Wd, bd, g3d, 1e-8, p, seed)
        ((y.float() * wy.to(dev)).sum() + (res.float() * wr.to(dev)).sum()).backward()
        return y.detach().float().cpu(), [t.grad.detach().float().cpu() for t in (xd, Wd, bd, g0d, g3d)]

    tol = TOL[torch.bfloat16]
    refs = [t.grad for t in (xr, Wr, br, g0r, g3r)]
    for fused in (True, False):
        y, grads = run(fused)
        assert (y - y_ref.detach()).abs().max().item() <= 4e-2 * y_ref.abs().max().item(), fused      # bf16 storage of an O(1..4) tensor
        for name, a, r in zip(("x", "W", "b", "gain0", "gain3"), grads, refs):
            assert (a - r).abs().max().item() <= tol["gelem"] * r.abs().max().item(), (fused, name)
            assert abs(a.norm().item() - r.norm().item()) <= tol["gnorm"] * r.norm().item(), (fused, name)


@pytest.mark.parametrize("p", [0.0, 0.5])
@pytest.mark.parametrize("d,G,S", [(768, 6, 32), (128, 5, 24), (768, 3, 196)])
def test_norm_linear_gelu_norm_pooled_against_torch(dev, d, G, S, p):
    """the pooled form of the last encoder layer: (mean_s dropout(RMSNorm(gelu(Linear(RMSNorm(x))))), mean_s x)"""
    from meant_amd import ops
    seed = 777
    gen = torch.Generator().manual_seed(d * 3 + S)
    x = torch.randn(G, S, d, generator=gen) * (1 + torch.rand(G, S, 1, generator=gen))
    W = torch.randn(d, d, generator=gen) / d ** 0.5
    b = torch.randn(d, generator=gen) * 0.1
    g0 = 1 + 0.1 * torch.randn(d, generator=gen)
    g3 = 1 + 0.1 * torch.randn(d, generator=gen)
    wm = torch.randn(G, d, generator=gen)
    wx = torch.randn(G, d, generator=gen)
    mask = _mask_of(G * S, d, p, seed, dev).view(G, S, d)
    assert ops.pooled_norm_ok(x.to(dev).bfloat16(), S)

    xr = x.bfloat16().float().clone().requires_grad_()
    Wr, br, g0r, g3r = (t.clone().requires_grad_() for t in (W, b, g0, g3))
    hm_ref = (_rms(torch.nn.functional.gelu(_rms(xr, g0r) @ Wr.t() + br), g3r) * mask).mean(dim=1)
    ((hm_ref * wm).sum() + (xr.mean(dim=1) * wx).sum()).backward()

    xd = x.detach().clone().to(dev).bfloat16().requires_grad_()
    Wd, bd, g0d, g3d = (t.detach().clone().to(dev).requires_grad_() for t in (W, b, g0, g3))
    hm, xm = ops.norm_linear_gelu_norm_pooled(xd, g0d, 1e-8, Wd, bd, g3d, 1e-8, p, seed)
    assert hm.dtype == torch.float32 and xm.dtype == torch.float32
    ((hm * wm.to(dev)).sum() + (xm * wx.to(dev)).sum()).backward()
    tol = TOL[torch.bfloat16]
    assert (xm.cpu() - xr.detach().mean(dim=1)).abs().max().item() <= 1e-5
    assert (hm.cpu() - hm_ref.detach()).abs().max().item() <= tol["out"]
    for name, a, r in zip(("x", "W", "b", "gain0", "gain3"), (xd, Wd, bd, g0d, g3d), (xr, Wr, br, g0r, g3r)):
        ag, rg = a.grad.float().cpu(), r.grad
        assert (ag - rg).abs().max().item() <= tol["gelem"] * rg.abs().max().item(), name
        assert abs(ag.norm().item() - rg.norm().item()) <= tol["gnorm"] * rg.norm().item(), name


@pytest.mark.parametrize("M,N,K", [(16384, 768, 768), (16384 + 96, 768, 768), (520, 256, 128), (4096, 3072, 768)])
def test_extended_epilogue_through_the_c_abi(dev, M, N, K):
    """meant_linear_fwd_rowscale and meant_linear_bwd_dx_norm called directly (streaming 256 x 256 kernel, its ragged last
    tile, the 128 x 128 kernel): y = gelu(r (x) (x W^T) + b) with the pre-activation, and dx = dy W - c (x) x + dres +
    dres_pooled / S, against fp32 PyTorch"""
    from meant_amd import _lib
    from meant_amd._lib import lib, check, BF16, EPI_GELU
    gen = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=gen).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev).bfloat16()
    b = (torch.randn(N, generator=gen) * 0.1).to(dev)
    r = (0.5 + torch.rand(M, generator=gen)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    pre = torch.empty_like(y)
    _lib.route_reset()
    check(lib.meant_linear_fwd_rowscale(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), r.data_ptr(), None, 0, y.data_ptr(), N, pre.data_ptr(),
                                        M, N, K, EPI_GELU, BF16, st), "fwd_rowscale")
    pre_ref = r[:, None] * (x.float() @ w.float().t()) + b
    assert (pre.float() - pre_ref).abs().max().item() <= 2e-2 * pre_ref.abs().max().item()
    assert (y.float() - torch.nn.functional.gelu(pre_ref)).abs().max().item() <= 2e-2 * pre_ref.abs().max().item()
    if M >= 4096:
        assert _lib.route_count("nt256s") >= 1
    # input gradient: dy [M, N] wT [K, N] -> dx [M, K]
    S = 8
    dy = torch.randn(M, N, generator=gen).to(dev).bfloat16()
    wT = w.t().contiguous()
    coef = torch.randn(M, generator=gen).to(dev) * 0.2
    dres = torch.randn(M, K, generator=gen).to(dev).bfloat16()
    dpool = torch.randn(M // S, K, generator=gen).to(dev)
    for with_res, with_pool in ((True, False), (False, True), (False, False)):
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        check(lib.meant_linear_bwd_dx_norm(dy.data_ptr(), N, wT.data_ptr(), x.data_ptr(), K, coef.data_ptr(), dres.data_ptr() if with_res else None,
                                           K if with_res else 0, dpool.data_ptr() if with_pool else None, S, dx.data_ptr(), K, M, N, K, BF16, st),
              "bwd_dx_norm")
        ref = dy.float() @ w.float() - coef[:, None] * x.float()
        if with_res:
            ref = ref + dres.float()
        if with_pool:
            ref = ref + dpool.repeat_interleave(S, dim=0) / S
        assert (dx.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item(), (with_res, with_pool)


def test_folded_and_separate_paths_agree_on_the_models(dev, monkeypatch):
    """ops.FUSE_NORM_LINEAR on vs off on whole models, train mode (same seeds -> same masks), bf16: outputs and parameter
    gradients agree to the tier's gates"""
    import meant_amd as M
    from meant_amd import ops
    torch.manual_seed(9)
    m = M.meant(128, 128, 4, 32, 64, 16, 2, 3, torch.nn.Embedding(50, 128), num_heads=2, num_encoders=2, channels=4).to(dev).train()
    m.compute_dtype = torch.bfloat16
    rs = np.random.RandomState(8)
    ids = torch.from_numpy(rs.randint(0, 50, (3, 2, 24))).to(dev)
    img = torch.from_numpy(rs.standard_normal((3, 2, 4, 32, 64)).astype("float32")).to(dev)
    mask = torch.ones(3, 2, 24, device=dev)
    mask[1, :, 17:] = 0
    res = []
    for fused in (False, True):
        monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", fused)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(31)
        out = m(ids, img, mask)
        (out * torch.arange(1, out.numel() + 1, device=dev).view_as(out)).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    tol = TOL[torch.bfloat16]
    assert (res[0][0] - res[1][0]).abs().max().item() <= tol["out"]
    assert res[0][1].keys() == res[1][1].keys()
    big = max(g.norm().item() for g in res[0][1].values())
    for k, g0 in res[0][1].items():
        assert (g0 - res[1][1][k]).norm().item() <= tol["gnorm"] * max(g0.norm().item(), 2e-2 * big), k


def test_checkpointed_layers_repeat_their_stacks_fold_decision(dev, monkeypatch):
    """activation checkpointing with the AUTO fold rule set so that the text stack folds and the vision stack does not (ADVICE r3:
    the decision used to be a process-wide hint read inside each layer's forward, and a recomputed vision layer saw the hint the
    text stack had left).  The fold is now decided once per stack and handed to the layers as an argument: the checkpointed run
    must reproduce the plain run's outputs and gradients, and both must take one folded and one separate path per layer pair."""
    import meant_amd as M
    from meant_amd import ops
    torch.manual_seed(9)
    m = M.meant(128, 128, 4, 32, 64, 16, 2, 3, torch.nn.Embedding(50, 128), num_heads=2, num_encoders=2, channels=4).to(dev).train()
    m.compute_dtype = torch.bfloat16
    rs = np.random.RandomState(8)
    ids = torch.from_numpy(rs.randint(0, 50, (3, 2, 24))).to(dev)
    img = torch.from_numpy(rs.standard_normal((3, 2, 4, 32, 64)).astype("float32")).to(dev)
    mask = torch.ones(3, 2, 24, device=dev)
    text_bytes = 2 * (3 * 2 * 24) * 128 * 2                  # layers x tokens x d x 2 B
    vision_bytes = 2 * (3 * 2 * 8) * 128 * 2
    assert vision_bytes < text_bytes
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", None)
    monkeypatch.setattr(ops, "FOLD_AUTO_BYTES", text_bytes)  # text stack: fold; vision stack: separate norms
    res = []
    for ck, recomputed in ((False, 0), (True, 2), (1, 1)):   # 1: only the first layer of each stack is recomputed
        m.activation_checkpointing = ck
        m.zero_grad(set_to_none=True)
        torch.manual_seed(123)
        before = list(ops.fold_calls)
        out = m(ids, img, mask)
        (out * torch.arange(1, out.numel() + 1, device=dev).view_as(out)).sum().backward()
        took = [ops.fold_calls[0] - before[0], ops.fold_calls[1] - before[1]]
        assert took == [2 + recomputed, 2 + recomputed], (ck, took)     # per stack: 2 layers forward (+ the recomputed ones)
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for k, g0 in res[0][1].items():
            assert (g0 - other[1][k]).abs().max().item() <= 1e-5 * max(1.0, g0.abs().max().item()), k
