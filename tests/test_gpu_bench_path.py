"""What bench.py times, pinned against something other than itself: the TRAIN-mode path (the language stack's Dropout(0.5)
of meant/meant.py:107 fused into the pooled / gelu-on-load RMSNorm kernels) and the 128-sample shapes of BASELINE.json
configs[2]/[3], whose q|k|v buffer is larger than 2^31 bytes.

The dropout mask of every norm kernel is a counter-based function of (seed, row * d + column) (csrc/common.h keep_scale8),
so the mask of a pooled call can be read out of a plain `ops.rmsnorm` call with the same seed and the pooled kernels
checked against an fp32 PyTorch evaluation that uses that mask."""
import os
import sys

import numpy as np
import pytest
import torch

from util import TOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _mask_of(rows, d, p, seed, dev):
    """keep / (1 - p) factors of the norm kernels' dropout, read from the non-pooled kernel: RMSNorm(ones) = 1 / (1 + eps)"""
    from meant_amd import ops
    y = ops.rmsnorm(torch.ones(rows, d, device=dev), torch.ones(d, device=dev), 1e-8, p, seed)
    keep = (y != 0).float()
    assert abs(keep.mean().item() - (1 - p)) < 0.02
    return keep / (1.0 - p)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("d,S", [(768, 32), (128, 24), (256, 8), (1024, 4)])
def test_pooled_gelu_dropout_norm_against_torch_with_the_extracted_mask(dev, dtype, d, S):
    """ops.linear_gelu_rmsnorm_pooled (GEMM -> rmsnorm_fwd_pooled_kernel<GELU_IN> -> rmsnorm_bwd_packed_kernel<5>) and
    ops.rmsnorm_fork_pooled (pooled residual operand), forward and backward, train-mode dropout live, against fp32 PyTorch:
    mean_s(mask * g * gelu(pre) / (rms(gelu(pre)) + eps)), pre = x W^T + b (meant/meant.py:106-107 -> :231)"""
    from meant_amd import ops
    G, p, seed, eps = 6, 0.5, 1234567, 1e-8
    gen = torch.Generator().manual_seed(d + S)
    x = torch.randn(G, S, d, generator=gen)
    W = torch.randn(d, d, generator=gen) / d ** 0.5
    b = torch.randn(d, generator=gen) * 0.1
    g = 1 + 0.1 * torch.randn(d, generator=gen)
    g2 = 1 + 0.1 * torch.randn(d, generator=gen)
    wm = torch.randn(G, d, generator=gen)                  # weights of the pooled features in the scalar objective
    wx = torch.randn(G, d, generator=gen)
    wn = torch.randn(G, S, d, generator=gen) * 0.1
    assert ops.pooled_norm_ok(x.to(dev), S)
    mask = _mask_of(G * S, d, p, seed, dev).view(G, S, d).cpu()

    def rms(t, gain):
        return gain * t / (t.norm(dim=-1, keepdim=True) / d ** 0.5 + eps)

    # fp32 reference on the values the device path sees (inputs rounded to the tier's storage type)
    xr = x.to(dtype).float().clone().requires_grad_()
    Wr, br, gr, g2r = (t.clone().requires_grad_() for t in (W.to(dtype).float() if dtype == torch.bfloat16 else W, b, g, g2))
    n_ref = rms(xr, g2r)                                   # rmsnorm_fork_pooled: (RMSNorm(x), mean_s x)
    xm_ref = xr.mean(dim=1)
    pre = n_ref @ Wr.t() + br
    hm_ref = (rms(torch.nn.functional.gelu(pre), gr) * mask).mean(dim=1)
    ((hm_ref * wm).sum() + (xm_ref * wx).sum()).backward()

    xd = x.detach().clone().to(dev).to(dtype).requires_grad_()
    Wd, bd, gd, g2d = (t.clone().to(dev).requires_grad_() for t in (W, b, g, g2))
    n, xm = ops.rmsnorm_fork_pooled(xd, g2d, eps)
    hm = ops.linear_gelu_rmsnorm_pooled(n, Wd, bd, gd, eps, p, seed)
    assert hm.dtype == torch.float32 and xm.dtype == torch.float32 and hm.shape == (G, d)
    ((hm * wm.to(dev)).sum() + (xm * wx.to(dev)).sum()).backward()

    tol = TOL[dtype]
    assert (xm.cpu() - xm_ref.detach()).abs().max().item() <= (1e-5 if dtype == torch.float32 else 1e-3)
    assert (hm.cpu() - hm_ref.detach()).abs().max().item() <= tol["out"] * max(1.0, hm_ref.abs().max().item())
    for name, a, r in (("x", xd, xr), ("W", Wd, Wr), ("b", bd, br), ("gain", gd, gr), ("gain_fork", g2d, g2r)):
        scale = r.grad.abs().max().item()
        err = (a.grad.float().cpu() - r.grad).abs().max().item() / scale
        assert err <= tol["gelem"], f"d{name}: {err:.3e}"
        na, nr = a.grad.float().norm().item(), r.grad.norm().item()
        assert abs(na - nr) <= tol["gnorm"] * nr, f"|d{name}|: {na} vs {nr}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cls,dim", [("meant", 128), ("meant_tweet", 768), ("meant_vqa", 256)])
def test_train_mode_pooled_tail_equals_literal_module_list(dev, monkeypatch, dtype, cls, dim):
    """modules.POOL_LAST_LINEAR on vs off in .train(): both draw the same seeds from torch's generator and key the mask on
    (row, column), so the pooled + gelu-on-load + dropout kernels must reproduce the literal RMSNorm -> Dropout -> Linear ->
    add -> mean-pool sequence (meant/meant.py:101-120 -> :231) with the SAME masks: outputs and every parameter gradient"""
    import meant_amd as M
    import meant_amd.modules as mm
    torch.manual_seed(5)
    H = dim // 64
    emb = torch.nn.Embedding(50, dim)
    if cls == "meant":
        m = M.meant(dim, dim, 4, 32, 64, 16, 2, 3, emb, num_heads=H, num_encoders=2, channels=4)
    elif cls == "meant_vqa":
        m = M.meant_vqa(dim, dim, 4, 32, 64, 16, 1, 5, emb, num_heads=H, num_encoders=1, channels=4)
    else:
        m = M.meant_tweet(dim, 4, 2, 3, emb, num_heads=H, num_encoders=1)
    m = m.to(dev).train()
    m.compute_dtype = dtype
    rs = np.random.RandomState(8)
    lag = 1 if cls == "meant_vqa" else 2
    ids = torch.from_numpy(rs.randint(0, 50, (3, lag, 24))).to(dev)
    img = torch.from_numpy(rs.standard_normal((3, lag, 4, 32, 64)).astype("float32")).to(dev)
    mask = torch.ones(3, lag, 24, device=dev)
    mask[1, :, 17:] = 0
    if cls == "meant_vqa":
        ids, img, mask = ids[:, 0], img[:, 0], mask[:, 0]
    args = {"meant": (ids, img, mask), "meant_vqa": (ids, img, mask), "meant_tweet": (ids, mask)}[cls]
    from meant_amd import ops
    assert ops.pooled_norm_ok(torch.empty(3 * lag, 24, dim, device=dev), 24)       # the pooled kernels really are on the path
    res = []
    for pooled in (False, True):
        monkeypatch.setattr(mm, "POOL_LAST_LINEAR", pooled)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(77)
        out = m(*args)
        (out * torch.arange(1, out.numel() + 1, device=dev).view_as(out)).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    torch.manual_seed(78)                                    # and the masks do matter: another seed moves the output
    with torch.no_grad():
        other = m(*args)
    assert not torch.equal(other, res[1][0])
    exact = dtype == torch.float32
    assert (res[0][0] - res[1][0]).abs().max().item() <= (2e-6 if exact else TOL[dtype]["out"])
    assert res[0][1].keys() == res[1][1].keys()
    big = max(g.norm().item() for g in res[0][1].values())
    for k, g0 in res[0][1].items():
        g1 = res[1][1][k]
        if exact:
            assert (g0 - g1).abs().max().item() <= 2e-5 * max(1.0, g0.abs().max().item()), k
        else:                                                # bf16: the literal path rounds [tokens, d] tensors the pooled one never forms
            assert (g0 - g1).norm().item() <= TOL[dtype]["gnorm"] * max(g0.norm().item(), 2e-2 * big), k


# ---- the bench's own batch: 128 samples, full dimensions ---------------------------------------------------------------
@pytest.fixture(scope="module")
def bench_case(dev):
    sys.path.insert(0, ROOT)
    import bench
    model = bench.build_model(1, dev)
    (tweets, images, mask), target = bench.make_batch(128, 0, dev)
    return model, (tweets, images, mask, target)


def test_128_sample_forward_rows_equal_two_sample_runs(dev, bench_case):
    """rows [0:2] and [126:128] of the 128-sample forward (q|k|v buffer 3.6 GB: byte offsets past 2^31, element offsets at
    84 % of 2^31) equal the corresponding 2-sample runs -- in eval mode for both ends, and in TRAIN mode for the first two
    rows, whose token rows (and therefore dropout masks) coincide in the two runs"""
    model, (tweets, images, mask, target) = bench_case
    model.eval()
    with torch.no_grad():
        full = model(tweets, images, mask)
        head = model(tweets[:2], images[:2], mask[:2])
        tail = model(tweets[126:], images[126:], mask[126:])
    assert torch.isfinite(full).all() and ((full > 0) & (full < 1)).all()
    assert (full[:2] - head).abs().max().item() <= 1e-6
    assert (full[126:] - tail).abs().max().item() <= 1e-6
    assert full.std(dim=0).min().item() > 0                  # not a constant
    model.train()
    with torch.no_grad():
        torch.manual_seed(21); full_t = model(tweets, images, mask)
        torch.manual_seed(21); head_t = model(tweets[:2], images[:2], mask[:2])
    model.eval()
    assert (full_t[:2] - head_t).abs().max().item() <= 1e-6
    assert not torch.equal(full_t, full)


def test_128_sample_gradients_equal_the_sum_of_two_64_sample_halves(dev, bench_case):
    """linearity of backward over the batch at the bench's size: what the data-parallel all-reduce assumes, and a check that
    no kernel of the backward pass mis-addresses rows beyond 2^31 bytes"""
    from meant_amd.train import cross_entropy_on_probs
    model, (tweets, images, mask, target) = bench_case
    model.eval()

    def grads(sl):
        model.zero_grad(set_to_none=True)
        out = model(tweets[sl], images[sl], mask[sl])
        n = out.shape[0]
        (cross_entropy_on_probs(out, target[sl]) * n).backward()           # sum over the samples
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    g_all, g_a, g_b = grads(slice(0, 128)), grads(slice(0, 64)), grads(slice(64, 128))
    big = max(v.norm().item() for v in g_all.values())
    worst, where = 0.0, ""
    for k, v in g_all.items():
        assert torch.isfinite(v).all(), k
        err = (v - (g_a[k] + g_b[k])).norm().item() / max(v.norm().item(), 1e-3 * big)
        if err > worst:
            worst, where = err, k
    assert worst <= 2e-2, (worst, where)
    # the second half contributes: the sum is not the first half alone
    k = "languageEncoders.0.encode.1.weight"
    assert (g_all[k] - g_a[k]).norm().item() > 0.1 * g_all[k].norm().item()
