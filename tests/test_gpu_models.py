"""GPU parity of the model classes (HIP path through the C ABI) against the golden vectors the
reference produced and against the CPU oracle on the same seeded inputs; both precision tiers."""
import numpy as np
import pytest
import torch

from tests.util import DTYPES, IDS, TOL, t, assert_close, assert_grad_close, pair, compare_param_grads, norm_floor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _run_golden(g, hip, inputs, dtype, dev, slices=True):
    hip.compute_dtype = dtype
    hip.zero_grad(set_to_none=True)
    out = hip(*[x.to(dev) for x in inputs])
    assert out.dtype == torch.float32
    loss = torch.nn.functional.cross_entropy(out, t(g["target"]).to(dev))
    loss.backward()
    tol = TOL[dtype]
    assert_close(out, t(g["out"]), tol["out"], "out")
    assert abs(loss.item() - float(g["loss"])) <= tol["out"]
    params = dict(hip.named_parameters())
    worst = 0.0
    floor = norm_floor(g["grad_norms"], dtype)
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        a = params[str(name)].grad.double().norm().item()
        rel = abs(a - ref) / max(ref, floor)
        worst = max(worst, rel)
        assert rel <= tol["gnorm"], (str(name), a, ref)
    if slices:
        for key in g.files:
            if key.startswith("grad__"):
                p = params[key[6:]]
                got = p.grad if p.grad.numel() <= 4096 else p.grad[:4]
                assert_grad_close(got, t(g[key]), tol["gelem"], key)
    return worst


def _mk(cls_name, args, kw, emb, dev):
    import meant_amd
    from oracle import meant_oracle as O
    a = list(args)
    ref = getattr(O, cls_name)(*(a + ([torch.nn.Embedding(*emb)] if emb else [])), **kw)
    hip = getattr(meant_amd, cls_name)(*(a + ([torch.nn.Embedding(*emb)] if emb else [])), **kw)
    return pair(ref, hip, 1234, dev)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_tiny_golden(golden, dev, dtype):
    g = golden("meant_tiny")
    _, hip = _mk("meant", (128, 128, 4, 32, 32, 16, 3, 2), dict(num_heads=2, num_encoders=1, channels=4), (100, 128), dev)
    _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_images"]), t(g["in_mask"])), dtype, dev)


@pytest.mark.parametrize("fold", [False, True], ids=["separate_norms", "norm_folded"])
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_tiny_two_layers_golden(golden, dev, dtype, fold, monkeypatch):
    """fold: the RMSNorm-into-Linear fold (ops.FUSE_NORM_LINEAR) is what `--encoders 12` runs at 128 samples per GPU; the
    reference-generated fixture pins it as well as the separate kernels (the fp32 tier has no folded path: same run twice)"""
    from meant_amd import ops
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", fold)
    g = golden("meant_tiny_e2")
    _, hip = _mk("meant", (128, 192, 4, 32, 48, 16, 2, 3), dict(num_heads=2, num_encoders=2, channels=4), (50, 128), dev)
    before = list(ops.fold_calls)
    _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_images"]), t(g["in_mask"])), dtype, dev)
    if dtype == torch.bfloat16:
        assert (ops.fold_calls[1] > before[1]) == fold       # the folded kernels really ran (or really did not)


@pytest.mark.parametrize("det", [0, 1], ids=["atomics", "deterministic"])
def test_meant_tiny_golden_deterministic_option(golden, dev, det):
    """MEANT_DETERMINISTIC=1 (ordered dW / dbias / embedding / gain reductions) against the same reference-generated fixture"""
    from meant_amd import _lib
    old = _lib.get_option("deterministic")
    _lib.set_option("deterministic", det)
    try:
        g = golden("meant_tiny")
        for dtype in DTYPES:
            _, hip = _mk("meant", (128, 128, 4, 32, 32, 16, 3, 2), dict(num_heads=2, num_encoders=1, channels=4), (100, 128), dev)
            _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_images"]), t(g["in_mask"])), dtype, dev)
    finally:
        _lib.set_option("deterministic", old)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_tweet_c1_golden(golden, dev, dtype):
    g = golden("meant_tweet_c1")
    _, hip = _mk("meant_tweet", (128, 4, 1, 2), dict(num_heads=2, num_encoders=1), (1000, 128), dev)
    _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_mask"])), dtype, dev)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_vision_tiny_golden(golden, dev, dtype):
    g = golden("meant_vision_tiny")
    _, hip = _mk("meant_vision", (128, 4, 32, 32, 16, 3, 2), dict(num_heads=2, num_encoders=1, channels=4), None, dev)
    _run_golden(g, hip, (t(g["in_images"]),), dtype, dev)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_vqa_tiny_golden(golden, dev, dtype):
    g = golden("meant_vqa_tiny")
    _, hip = _mk("meant_vqa", (128, 128, 4, 32, 32, 16, 1, 7), dict(num_heads=2, num_encoders=1, channels=4), (100, 128), dev)
    _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_images"]), t(g["in_mask"])), dtype, dev)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_vision_c2_golden(golden, dev, dtype):
    """BASELINE config 2: meant_vision lag=1, 224x224 p=16, d=768, 12 heads."""
    g = golden("meant_vision_c2")
    r = np.random.RandomState(102)
    img = t(r.standard_normal((2, 1, 4, 224, 224)).astype("float32"))
    _, hip = _mk("meant_vision", (768, 4, 224, 224, 16, 1, 2), dict(num_heads=12, num_encoders=1, channels=4), None, dev)
    _run_golden(g, hip, (img,), dtype, dev)


@pytest.mark.parametrize("fold", [False, True], ids=["separate_norms", "norm_folded"])
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_full_c3_golden(golden, dev, dtype, fold, monkeypatch):
    """BASELINE config 3 at full dims: lag 12, d 768, 12 heads, S 512, 224x224 (B=2, V=2000); with the RMSNorm-into-Linear fold
    off and on (bf16 tier; the fp32 tier has no folded path and is run once)."""
    from meant_amd import ops
    if fold and dtype != torch.bfloat16:
        pytest.skip("the fold exists in the bf16 tier only")
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", fold)
    g = golden("meant_full_c3")
    r = np.random.RandomState(99)
    ids = t(r.randint(0, 2000, (2, 12, 512)).astype("int64"))
    img = t(r.standard_normal((2, 12, 4, 224, 224)).astype("float32"))
    mask = torch.ones(2, 12, 512)
    mask[1, :, 400:] = 0
    _, hip = _mk("meant", (768, 768, 4, 224, 224, 16, 12, 2), dict(num_heads=12, num_encoders=1), (2000, 768), dev)
    worst = _run_golden(g, hip, (ids, img, mask), dtype, dev)
    with torch.no_grad():
        err = (hip(ids.to(dev), img.to(dev), mask.to(dev)).cpu() - t(g["out"])).abs().max().item()
    print(f"full C3 [{dtype}]: max |out - golden| {err:.3e}, worst grad-norm rel err {worst:.3e}")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_model_vs_oracle_ragged(dev, dtype):
    """oracle comparison on shapes no fixture covers: ragged token / patch counts, 3 lag steps, batch 5"""
    ref, hip = _mk("meant", (128, 128, 4, 48, 80, 16, 3, 4), dict(num_heads=2, num_encoders=1, channels=4), (77, 128), dev)
    r = np.random.RandomState(7)
    ids = t(r.randint(0, 77, (5, 3, 41)).astype("int64"))
    img = t(r.standard_normal((5, 3, 4, 48, 80)).astype("float32"))
    mask = torch.ones(5, 3, 41)
    mask[0, :, 30:] = 0
    mask[3, 1, 1:] = 0
    tgt = torch.tensor([0, 3, 1, 2, 2])
    out_r = ref(ids, img, mask)
    torch.nn.functional.cross_entropy(out_r, tgt).backward()
    hip.compute_dtype = dtype
    out = hip(ids.to(dev), img.to(dev), mask.to(dev))
    torch.nn.functional.cross_entropy(out, tgt.to(dev)).backward()
    assert_close(out, out_r, TOL[dtype]["out"], "out")
    compare_param_grads(ref, hip, dtype, "ragged")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_meant_vqa_full_width_vs_oracle(dev, dtype):
    """BASELINE config 5 geometry: meant_vqa d=768, 12 heads, seq=512, 224x224, 3129 answer classes (B=2)"""
    ref, hip = _mk("meant_vqa", (768, 768, 4, 224, 224, 16, 1, 3129), dict(num_heads=12, num_encoders=1), (3000, 768), dev)
    r = np.random.RandomState(21)
    ids = t(r.randint(0, 3000, (2, 512)).astype("int64"))
    img = t(r.standard_normal((2, 4, 224, 224)).astype("float32"))
    mask = torch.ones(2, 512)
    mask[1, 300:] = 0
    tgt = torch.tensor([17, 3000])
    out_r = ref(ids, img, mask)
    torch.nn.functional.cross_entropy(out_r, tgt).backward()
    hip.compute_dtype = dtype
    out = hip(ids.to(dev), img.to(dev), mask.to(dev))
    assert out.shape == (2, 3129)
    torch.nn.functional.cross_entropy(out, tgt.to(dev)).backward()
    assert_close(out, out_r, TOL[dtype]["out"], "out")
    ref_p = dict(ref.named_parameters())
    floor = norm_floor([p.grad.double().norm().item() for p in ref_p.values() if p.grad is not None], dtype)
    for k, p in hip.named_parameters():
        if ref_p[k].grad is None:
            continue
        a, b = p.grad.double().norm().item(), ref_p[k].grad.double().norm().item()
        assert abs(a - b) <= TOL[dtype]["gnorm"] * max(b, floor), (k, a, b)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_default_eight_heads_head_dim_96(dev, dtype):
    """the reference's own runs never pass num_heads (default 8 -> head dim 96, rotary dim 48 for the patches): the bf16
    tier serves it on the 96-wide attention kernels"""
    ref, hip = _mk("meant", (768, 768, 4, 64, 64, 16, 2, 2), dict(num_encoders=1), (500, 768), dev)     # num_heads default 8
    r = np.random.RandomState(22)
    ids = t(r.randint(0, 500, (2, 2, 40)).astype("int64"))
    img = t(r.standard_normal((2, 2, 4, 64, 64)).astype("float32"))
    mask = torch.ones(2, 2, 40)
    mask[0, :, 33:] = 0
    tgt = torch.tensor([1, 0])
    out_r = ref(ids, img, mask)
    torch.nn.functional.cross_entropy(out_r, tgt).backward()
    hip.compute_dtype = dtype
    out = hip(ids.to(dev), img.to(dev), mask.to(dev))
    torch.nn.functional.cross_entropy(out, tgt.to(dev)).backward()
    assert_close(out, out_r, TOL[dtype]["out"], "out")
    compare_param_grads(ref, hip, dtype, "heads8")


def test_autocast_selects_bf16_and_state_dict_roundtrip(dev):
    import meant_amd
    m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    ids = torch.randint(0, 100, (2, 3, 16), device=dev)
    img = torch.randn(2, 3, 4, 32, 32, device=dev)
    mask = torch.ones(2, 3, 16, device=dev)
    o32 = m(ids, img, mask)
    with torch.autocast("cuda", dtype=torch.float16):          # what in_loop_train.py:215 does
        o16 = m(ids, img.half(), attention_mask=mask)
    assert o16.dtype == torch.float32 and (o16 - o32).abs().max().item() < 1e-2
    assert (o16 - o32).abs().max().item() > 0                  # really took the bf16 path
    sd = m.state_dict()
    m2 = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    m2.load_state_dict(sd)
    assert torch.equal(m2(ids, img, mask), o32)
    import io, pickle
    buf = io.BytesIO()
    torch.save(m, buf)                                          # whole-module pickle, in_loop_train.py:331
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert torch.equal(m3(ids, img, mask), o32)


def test_small_class_heads_run_in_fp32_in_the_bf16_tier(dev):
    """modules._head: a head of at most HEAD_F32_MAX_CLASSES outputs gets fp32 features and returns probabilities that were never
    rounded to bf16 (the loss differentiates them: meant/meant.py:204 -> in_loop_train.py:222); wider heads stay on the bf16 GEMM"""
    import meant_amd
    from meant_amd import modules
    seen = {}

    def hook(name):
        def f(mod, inp, out):
            seen[name] = (inp[0].dtype, out.dtype)
        return f
    m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    m.compute_dtype = torch.bfloat16
    m.mlpHead[1].register_forward_hook(hook("small"))
    ids = torch.randint(0, 100, (4, 3, 16), device=dev)
    img = torch.randn(4, 3, 4, 32, 32, device=dev)
    out = m(ids, img, torch.ones(4, 3, 16, device=dev))
    assert seen["small"] == (torch.float32, torch.float32)
    assert out.dtype == torch.float32 and not torch.equal(out, out.bfloat16().float())        # not bf16-quantised
    out.sum().backward()
    assert m.mlpHead[1].weight.grad is not None and torch.isfinite(m.mlpHead[1].weight.grad).all()
    wide = modules.HEAD_F32_MAX_CLASSES + 8
    m2 = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, wide, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    m2.compute_dtype = torch.bfloat16
    m2.mlpHead[1].register_forward_hook(hook("wide"))
    m2(ids, img, torch.ones(4, 3, 16, device=dev))
    assert seen["wide"] == (torch.bfloat16, torch.bfloat16)


@pytest.mark.gpu
def test_model_on_device_batch_loader_float64(dev):
    """the reference's data path end to end (in_loop_train.py:579-639 -> :202-217): float64 graphs on the host,
    global normalisation, batches through the double-buffered loader, forward under autocast; must equal the forward
    on tensors prepared by hand"""
    import numpy as np
    import meant_amd as M
    from meant_amd.data import DeviceBatchLoader, global_mean_std
    rs = np.random.RandomState(11)
    n, L, S = 6, 3, 16
    graphs = rs.standard_normal((n, L, 4, 32, 32)) * 2 + 0.5
    tweets = rs.randint(0, 100, (n, L, S))
    masks = np.ones((n, L, S), dtype=np.float32)
    masks[1, :, 12:] = 0
    labels = rs.randint(0, 2, (n,))
    torch.manual_seed(0)
    model = M.meant(128, 128, 4, 32, 32, 16, L, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=1, channels=4).to(dev).eval()
    mean, std = global_mean_std(graphs)
    model.patchEmbed[0].set_normalization(mean, std)
    got = None
    for pin in (0, 1):                                   # staged through pinned buffers / DMA out of the page-locked arrays
        loader = DeviceBatchLoader(graphs, tweets, None, masks, labels, batch_size=2, device=dev, pin_source_bytes=pin)
        outs = []
        for g, tw, _, am, y in loader:
            assert g.dtype == torch.float64 and g.is_cuda
            outs.append(model(tw.long(), g, am).float().cpu())
        loader.close()
        if got is not None:
            assert torch.equal(got, torch.cat(outs))
        got = torch.cat(outs)
    model.patchEmbed[0].set_normalization(0.0, 1.0)
    gn = torch.from_numpy(((graphs - mean) / std).astype("float32")).to(dev)
    ref = model(torch.from_numpy(tweets).to(dev), gn, torch.from_numpy(masks).to(dev)).float().cpu()
    assert got.shape == (n, 2)
    assert (got - ref).abs().max().item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_mlm_pretrainer_golden(dev, golden, dtype):
    """pretrain_mlm.py:74-88 + CrossEntropyLoss(:160) against the reference's own class (fixture), both
    through forward() + the fused loss on the returned logits view and through loss() (padded logits end to end)"""
    import meant_amd as M
    from oracle import meant_oracle as O
    g = golden("mlm_pretrainer_tiny")
    ids, mask, labels = (torch.from_numpy(g[k]).to(dev) for k in ("ids", "mask", "labels"))
    tol_out, tol_loss, tol_g = (2e-4, 2e-5, 2e-3) if dtype == torch.float32 else (6e-2, 2e-2, 6e-2)
    for mode in ("forward", "loss"):
        torch.manual_seed(0)
        emb, head = O.mlm_parts()
        m = M.meant_language_pretrainer(2, 128, emb, head, text_dim=128, num_heads=2)
        O.fill_weights_(m, 2468)
        m = m.to(dev).eval()
        m.compute_dtype = dtype
        if mode == "forward":
            out = m(ids, attention_mask=mask)
            assert out.shape == (3, 24, 120)
            assert (out.float().cpu() - torch.from_numpy(g["logits"])).abs().max().item() < tol_out * max(1.0, float(np.abs(g["logits"]).max()))
            loss = M.ops.softmax_cross_entropy(out, labels)
        else:
            loss = m.loss(ids, mask, labels)
        assert abs(loss.item() - float(g["loss"])) < tol_loss * max(1.0, float(g["loss"])), (mode, loss.item(), float(g["loss"]))
        loss.backward()
        params = dict(m.named_parameters())
        floor = 1e-3 * float(np.max(g["grad_norms"]))
        for nm, refn in zip(g["grad_names"], g["grad_norms"]):
            got = params[str(nm)].grad.double().norm().item()
            assert abs(got - refn) <= tol_g * max(refn, floor), (mode, str(nm), got, refn)
        for k in g.files:
            if k.startswith("grad__"):
                p_ = params[k[6:]]
                ref = torch.from_numpy(g[k])
                got = (p_.grad if p_.grad.numel() <= 4096 else p_.grad[:4]).float().cpu()
                assert (got - ref).abs().max().item() <= tol_g * max(ref.abs().max().item(), floor), (mode, k)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_softmax_cross_entropy_large_vocab(dev, dtype):
    """the fused loss at the real vocabulary width (V = 64001, ragged: not a multiple of 8), ignored rows, padding
    columns: loss and d logits against torch on the same (rounded) logits"""
    import meant_amd as M
    T, V = 37, 64001
    gen = torch.Generator().manual_seed(3)
    logits = (torch.randn(T, V, generator=gen) * 3).to(dtype)
    target = torch.randint(0, V, (T,), generator=gen)
    target[::5] = -100
    ref_in = logits.float().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, target)
    ref.backward()
    x = logits.to(dev).requires_grad_()
    loss = M.ops.softmax_cross_entropy(x, target.to(dev))
    loss.backward()
    assert abs(loss.item() - ref.item()) < (1e-5 if dtype == torch.float32 else 2e-3) * ref.item()
    tol = 1e-6 if dtype == torch.float32 else 2e-2 * ref_in.grad.abs().max().item()
    assert (x.grad.float().cpu() - ref_in.grad).abs().max().item() <= max(tol, 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_mim_pretrainer_golden(dev, golden, dtype):
    """pretrain_mim.py:77-99 + L1Loss (:162) against the reference's own class (fixture); the HF ViT
    decoder's 1x1 convolution runs as a GEMM on the HIP path"""
    import meant_amd as M
    from oracle import meant_oracle as O
    g = golden("mim_pretrainer_tiny")
    torch.manual_seed(0)
    m = M.meant_vision_pretrainer(1, O.mim_decoder(), 128, patch_res=16, channels=4, height=32, width=32, image_dim=128, num_heads=2)
    O.fill_weights_(m, 1357)
    m = m.to(dev).eval()
    m.compute_dtype = dtype
    out = m(torch.from_numpy(g["images"]).to(dev))
    assert out.shape == (3, 3, 32, 32)
    scale = float(np.abs(g["out"]).max())
    assert (out.float().cpu() - torch.from_numpy(g["out"])).abs().max().item() < (2e-5 if dtype == torch.float32 else 3e-2) * scale
    loss = torch.nn.functional.l1_loss(out.float(), torch.from_numpy(g["target"]).to(dev)[:, 0:3])
    assert abs(loss.item() - float(g["loss"])) < (1e-5 if dtype == torch.float32 else 1e-2) * float(g["loss"])
    loss.backward()
    params = dict(m.named_parameters())
    tol_g = 2e-3 if dtype == torch.float32 else 6e-2
    floor = 1e-3 * float(np.max(g["grad_norms"]))
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(nm)].grad.double().norm().item()
        assert abs(got - refn) <= tol_g * max(refn, floor), (str(nm), got, refn)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("fixture", ["timesformer_tiny", "timesformer_tiny_mask", "timesformer_tiny_shift", "timesformer_tiny_posemb"])
def test_timesformer_golden(dev, golden, dtype, fixture):
    """divided space-time attention (time then space, cls token, frame + axial rotary, GEGLU; with and
    without the frame mask) against the fork's TimeSformer (fixtures from src/meant/timesformer_pytorch.py).  The whole
    attention half runs on the HIP path: row gathers, flash core, cls-query kernel (route counter)."""
    import meant_amd as M
    from meant_amd import _lib
    from oracle import meant_oracle as O
    g = golden(fixture)
    torch.manual_seed(0)
    from tests.test_oracle_golden import TS_CFG
    cfg = dict(TS_CFG[fixture])
    seed = cfg.pop("seed")
    m = M.TimeSformer(image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64, **cfg)
    O.fill_weights_(m, seed)
    m = m.to(dev).eval()
    m.compute_dtype = dtype
    mask = torch.from_numpy(g["mask"]).to(dev) if "mask" in g.files else None
    _lib.route_reset()
    x = m.meant_forward(torch.from_numpy(g["video"]).to(dev), mask=mask)
    assert _lib.route_count("attn_cls") == 4                 # 2 layers x (time, space): the cls query's attention is a HIP kernel
    logits = m.to_out(x[:, 0])
    assert x.shape == (2, 1 + cfg["num_frames"] * 4, cfg["dim"]) and logits.shape == (2, cfg["num_classes"])
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    assert (x.float().cpu() - torch.from_numpy(g["tokens"])).abs().max().item() < tol * float(np.abs(g["tokens"]).max())
    assert (logits.float().cpu() - torch.from_numpy(g["logits"])).abs().max().item() < tol * max(1.0, float(np.abs(g["logits"]).max()))
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.from_numpy(g["target"]).to(dev)) + 0.01 * x.float().pow(2).mean()
    assert abs(loss.item() - float(g["loss"])) < (1e-4 if dtype == torch.float32 else 2e-2) * float(g["loss"])
    loss.backward()
    params = dict(m.named_parameters())
    tol_g = 2e-3 if dtype == torch.float32 else 6e-2
    floor = 1e-3 * float(np.max(g["grad_norms"]))
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(nm)].grad.double().norm().item()
        assert abs(got - refn) <= tol_g * max(refn, floor), (str(nm), got, refn)
    for k in g.files:
        if k.startswith("grad__"):
            p_ = params[k[6:]]
            ref = torch.from_numpy(g[k])
            got = (p_.grad if p_.grad.numel() <= 4096 else p_.grad[:4]).float().cpu()
            assert (got - ref).abs().max().item() <= tol_g * max(ref.abs().max().item(), floor), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_gather_rows_rot_equals_gather_then_rotary(dev, dtype):
    """the one-pass regroup + rotary of the divided attention == meant_gather_rows followed by meant_rotary_qk in place, bit for bit
    (zero rows for index -1, partial rotary dim, v untouched)"""
    from meant_amd import _lib
    from meant_amd.ops import _p, _dt, _stream, check, gather_rows
    lib = _lib.lib
    rs = np.random.RandomState(3)
    rows, S, H, Dh, R = 500, 13, 3, 64, 48
    src = torch.from_numpy(rs.standard_normal((rows, 3 * H * Dh)).astype("float32")).to(dev).to(dtype)
    n = 40 * S
    idx = torch.from_numpy(rs.randint(0, rows, size=n).astype("int32")).to(dev)
    idx[::7] = -1
    tabs = [torch.from_numpy(rs.standard_normal((S, R)).astype("float32")).to(dev) for _ in range(4)]
    a = gather_rows(src, idx)
    check(lib.meant_rotary_qk(_p(a), n, S, H, Dh, R, *[_p(t_) for t_ in tabs], 0, _dt(a), _stream()), "rotary_qk")
    b = torch.empty_like(a)
    check(lib.meant_gather_rows_rot(_p(src), _p(idx), _p(b), n, S, H, Dh, R, *[_p(t_) for t_ in tabs], _dt(b), _stream()), "gather_rows_rot")
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(b[::7], torch.zeros_like(b[::7]))


@pytest.mark.gpu
def test_dropout_kernel_statistics_and_backward(dev):
    """meant_dropout (the TimeSformer's attention / feed-forward dropouts, src/meant/timesformer_pytorch.py:70,101): keep rate
    1 - p, survivors scaled by 1 / (1 - p), the same mask in backward, a different one for another seed"""
    from meant_amd import ops
    for dtype in (torch.float32, torch.bfloat16):
        x = (torch.rand(512, 768, device=dev) + 0.5).to(dtype).requires_grad_()
        y = ops.dropout(x, 0.25, 99)
        kept = y != 0
        assert 0.74 < kept.float().mean().item() < 0.76
        assert ((y[kept].float() - x.detach()[kept].float() / 0.75).abs() <= 2e-2 * x.detach()[kept].float()).all()
        y.backward(torch.ones_like(y))
        assert torch.equal(x.grad != 0, kept) and (x.grad[kept].float() - 1 / 0.75).abs().max().item() < 1e-2
        assert not torch.equal(ops.dropout(x, 0.25, 100) != 0, kept)
        assert ops.dropout(x, 0.0, 1) is x


@pytest.mark.gpu
def test_activation_checkpointing_same_gradients(dev):
    """model.activation_checkpointing = True (the E = 12 memory switch): identical outputs and gradients, also in train
    mode where the recomputation has to redraw the forward's dropout masks"""
    import meant_amd as M
    torch.manual_seed(0)
    m = M.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2, channels=4).to(dev).train()
    m.compute_dtype = torch.bfloat16
    rs = np.random.RandomState(4)
    ids = torch.from_numpy(rs.randint(0, 100, (2, 3, 16))).to(dev)
    img = torch.from_numpy(rs.standard_normal((2, 3, 4, 32, 32)).astype("float32")).to(dev)
    mask = torch.ones(2, 3, 16, device=dev)
    res = []
    for ck in (False, True, 1):                              # 1: only the first layer of each stack is recomputed
        m.activation_checkpointing = ck
        m.zero_grad(set_to_none=True)
        torch.manual_seed(123)                               # same dropout seeds in both runs
        out = m(ids, img, mask)
        out.sum().backward()
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        for k, g0 in res[0][1].items():
            g1 = other[1][k]
            assert (g0 - g1).abs().max().item() <= 1e-5 * max(1.0, g0.abs().max().item()), k


@pytest.mark.parametrize("cls", ["meant", "meant_vqa", "meant_tweet", "meant_vision"])
def test_pooled_last_linear_equals_literal_module_list(dev, monkeypatch, cls):
    """modules.POOL_LAST_LINEAR evaluates mean_s(h W^T + b + x) of the last encoder layer (meant/meant.py:74,120 -> :231) as
    mean_s(h) W^T + b + mean_s(x): outputs and every parameter gradient must equal the literal Linear -> add -> mean-pool
    sequence (fp32 tier: to rounding), for two stacks, one stack, ragged token counts (S = 24 vs n = 6) and two layers"""
    import meant_amd as M
    import meant_amd.modules as mm
    torch.manual_seed(3)
    emb = torch.nn.Embedding(50, 128)
    if cls == "meant":
        m = M.meant(128, 192, 4, 32, 48, 16, 2, 3, emb, num_heads=2, num_encoders=2, channels=4)
    elif cls == "meant_vqa":
        m = M.meant_vqa(128, 128, 4, 32, 48, 16, 1, 5, emb, num_heads=2, num_encoders=1, channels=4)
    elif cls == "meant_tweet":
        m = M.meant_tweet(128, 4, 2, 3, emb, num_heads=2, num_encoders=1)
    else:
        m = M.meant_vision(128, 4, 32, 48, 16, 2, 3, num_heads=2, num_encoders=2, channels=4)
    m = m.to(dev).eval()
    rs = np.random.RandomState(8)
    lag = 1 if cls == "meant_vqa" else 2
    ids = torch.from_numpy(rs.randint(0, 50, (3, lag, 24))).to(dev)
    img = torch.from_numpy(rs.standard_normal((3, lag, 4, 32, 48)).astype("float32")).to(dev)
    mask = torch.ones(3, lag, 24, device=dev)
    mask[1, :, 17:] = 0
    if cls == "meant_vqa":
        ids, img, mask = ids[:, 0], img[:, 0], mask[:, 0]
    args = {"meant": (ids, img, mask), "meant_vqa": (ids, img, mask), "meant_tweet": (ids, mask), "meant_vision": (img,)}[cls]
    res = []
    for pooled in (False, True):
        monkeypatch.setattr(mm, "POOL_LAST_LINEAR", pooled)
        m.zero_grad(set_to_none=True)
        out = m(*args)
        (out * torch.arange(1, out.numel() + 1, device=dev).view_as(out)).sum().backward()
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert (res[0][0] - res[1][0]).abs().max().item() <= 2e-6
    assert res[0][1].keys() == res[1][1].keys()
    for k, g0 in res[0][1].items():
        g1 = res[1][1][k]
        assert (g0 - g1).abs().max().item() <= 2e-5 * max(1.0, g0.abs().max().item()), k


def test_fp32_tail_switch(golden, dev, monkeypatch):
    """ops.TAIL_FP32 = True keeps the temporal encoder and the head in fp32 inside the bf16 tier (the default lets them follow
    the tier): same gates, and the pooled features really are fp32"""
    from meant_amd import ops
    monkeypatch.setattr(ops, "TAIL_FP32", True)
    a = torch.randn(2, 5, 128, device=dev, dtype=torch.bfloat16)
    assert ops.meanpool_cat(a).dtype == torch.float32
    g = golden("meant_tiny")
    _, hip = _mk("meant", (128, 128, 4, 32, 32, 16, 3, 2), dict(num_heads=2, num_encoders=1, channels=4), (100, 128), dev)
    _run_golden(g, hip, (t(g["in_tweets"]), t(g["in_images"]), t(g["in_mask"])), torch.bfloat16, dev)
    monkeypatch.setattr(ops, "TAIL_FP32", False)
    assert ops.meanpool_cat(a).dtype == torch.bfloat16
