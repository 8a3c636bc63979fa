"""Pin the CPU oracle (oracle/meant_oracle.py) to golden vectors produced by the
reference itself (oracle/gen_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import meant_oracle as O

TOL = 2e-5   # fp32 CPU <-> fp32 CPU, different op order


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_rmsnorm_kat_and_grads(golden):
    g = golden("rmsnorm_768")
    n = O.RMSNorm(4)
    np.testing.assert_allclose(n(_t(g["kat_in"])).detach().numpy(), g["kat_out"], atol=1e-6)
    np.testing.assert_allclose(g["kat_out"], [0.365148, 0.730297, 1.095445, 1.460593], atol=1e-6)
    n = O.RMSNorm(768)
    with torch.no_grad():
        n.scale.copy_(_t(g["scale"]))
    x = _t(g["x"]).requires_grad_()
    y = n(x)
    y.backward(_t(g["dy"]))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], atol=TOL)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=TOL)
    np.testing.assert_allclose(n.scale.grad.numpy(), g["dscale"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_rmsnorm_partial_and_bias_forms(golden, tag):
    """the partial (statistics over the first int(d p) elements) and bias (learned offset) forms of utils/rms_norm.py:44-57 against the
    reference class's own outputs and gradients: p = 0.5 + bias, p = 0.3, p = 1.0 + bias, p = -1 + bias"""
    g = golden("rmsnorm_partial_bias")
    d, p, bias = int(g[f"{tag}_cfg"][0]), float(g[f"{tag}_cfg"][1]), bool(g[f"{tag}_cfg"][2])
    n = O.RMSNorm(d, p=p, bias=bias)
    with torch.no_grad():
        n.scale.copy_(_t(g[f"{tag}_scale"]))
        if bias:
            n.offset.copy_(_t(g[f"{tag}_offset"]))
    x = _t(g[f"{tag}_x"]).requires_grad_()
    y = n(x)
    y.backward(_t(g[f"{tag}_dy"]))
    np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}_y"], atol=TOL)
    np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_dx"], atol=TOL)
    np.testing.assert_allclose(n.scale.grad.numpy(), g[f"{tag}_dscale"], rtol=1e-4, atol=1e-4)
    if bias:
        np.testing.assert_allclose(n.offset.grad.numpy(), g[f"{tag}_doffset"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("S", [16, 64, 512])
def test_xpos_rotation(golden, S):
    g = golden("rotary_xpos48")
    tab = O.RotaryTable(48, "lang", use_xpos=True)
    np.testing.assert_allclose(tab.freqs.numpy(), g["freqs"], rtol=1e-6)
    np.testing.assert_allclose(tab.scale.numpy(), g["scale"], rtol=1e-6)
    cos, sin = tab.cos_sin(S)
    z = tab.xpos_scale(S)
    rq = O.rotate_pairs(_t(g[f"q{S}"]), cos, sin, z)
    rk = O.rotate_pairs(_t(g[f"k{S}"]), cos, sin, z ** -1)
    np.testing.assert_allclose(rq.numpy(), g[f"rq{S}"], atol=TOL)
    np.testing.assert_allclose(rk.numpy(), g[f"rk{S}"], atol=TOL)
    # lanes >= 48 untouched
    np.testing.assert_array_equal(rq.numpy()[..., 48:], g[f"q{S}"][..., 48:])


def test_xpos_survey_kat():
    tab = O.RotaryTable(48, "lang", use_xpos=True)
    assert abs(tab.scale[0].item() - 0.2857143) < 1e-6 and abs(tab.scale[23].item() - 0.9702381) < 1e-6
    assert abs(tab.freqs[1].item() - 0.6812921) < 1e-6
    cos, sin = tab.cos_sin(4)
    z = tab.xpos_scale(4)
    one = torch.ones(1, 1, 4, 64)
    q = O.rotate_pairs(one, cos, sin, z)[0, 0, 3, :4]
    k = O.rotate_pairs(one, cos, sin, z ** -1)[0, 0, 3, :4]
    np.testing.assert_allclose(q.numpy(), [-1.128348, -0.846962, -1.343007, 0.433708], atol=2e-6)
    np.testing.assert_allclose(k.numpy(), [-1.133883, -0.850787, -1.348598, 0.435372], atol=2e-6)


@pytest.mark.parametrize("dim,N", [(32, 196), (48, 196), (32, 4)])
def test_pixel_rotation(golden, dim, N):
    g = golden("rotary_pixel")
    tab = O.RotaryTable(dim, "pixel")
    np.testing.assert_allclose(tab.freqs.numpy(), g[f"freqs_{dim}"], rtol=1e-6)
    cos, sin = tab.cos_sin(N)
    r = O.rotate_pairs(_t(g[f"t_{dim}_{N}"]), cos, sin)
    np.testing.assert_allclose(r.numpy(), g[f"r_{dim}_{N}"], atol=TOL)


def _check_module(g, mod, args):
    O.fill_weights_(mod, 4321)
    mod.eval()
    x = _t(g["x"]).requires_grad_()
    y = mod(x, *args)
    y.backward(_t(g["dy"]))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], atol=TOL)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=5e-5)
    params = dict(mod.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        p = params[str(name)]
        assert abs(p.grad.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-3), name
        ga = g["grad__" + str(name)]
        got = p.grad if p.grad.numel() <= 4096 else p.grad[:4]
        np.testing.assert_allclose(got.numpy(), ga, atol=1e-4, rtol=1e-4)


def test_attention_module(golden):
    g = golden("attention_h2_d128_n196")
    _check_module(g, O.attention(2, 128, O.RotaryTable(32, "pixel")), ())


@pytest.mark.parametrize("name", ["xposattention_h2_d128_s80", "xposattention_h2_d128_s512"])
def test_xpos_attention_module(golden, name):
    g = golden(name)
    _check_module(g, O.xPosAttention(2, 128, O.RotaryTable(48, "lang", use_xpos=True)), (_t(g["mask"]),))


def test_temporal_module(golden):
    g = golden("temporal_h12_d1536_l12")
    _check_module(g, O.temporal(12, 1536), ())


def _check_model(g, model, inputs, grad_rtol=2e-4):
    O.fill_weights_(model, 1234)
    model.eval()
    out = model(*inputs)
    loss = O.cross_entropy_on_probs(out, _t(g["target"]))
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=TOL)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    params = dict(model.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(name)].grad.double().norm().item()
        assert abs(got - ref) <= grad_rtol * max(ref, 1e-6) + 1e-7, (name, got, ref)
    for key in g.files:
        if key.startswith("grad__"):
            p = params[key[6:]]
            got = p.grad if p.grad.numel() <= 4096 else p.grad[:4]
            np.testing.assert_allclose(got.numpy(), g[key], atol=1e-5, rtol=1e-3)


def test_meant_tiny(golden):
    g = golden("meant_tiny")
    m = O.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=1, channels=4)
    _check_model(g, m, (_t(g["in_tweets"]), _t(g["in_images"]), _t(g["in_mask"])))
    # the numbers recorded in SURVEY.md 8(c) from the reference
    np.testing.assert_allclose(g["out"], [[0.419537, 0.637061], [0.379703, 0.602564]], atol=1e-6)
    assert abs(float(g["loss"]) - 0.6978624) < 1e-6


def test_meant_tiny_two_layers(golden):
    g = golden("meant_tiny_e2")
    m = O.meant(128, 192, 4, 32, 48, 16, 2, 3, torch.nn.Embedding(50, 128), num_heads=2, num_encoders=2, channels=4)
    _check_model(g, m, (_t(g["in_tweets"]), _t(g["in_images"]), _t(g["in_mask"])))


def test_meant_tweet_c1(golden):
    g = golden("meant_tweet_c1")
    m = O.meant_tweet(128, 4, 1, 2, torch.nn.Embedding(1000, 128), num_heads=2, num_encoders=1)
    _check_model(g, m, (_t(g["in_tweets"]), _t(g["in_mask"])))


def test_meant_vision_tiny(golden):
    g = golden("meant_vision_tiny")
    m = O.meant_vision(128, 4, 32, 32, 16, 3, 2, num_heads=2, num_encoders=1, channels=4)
    _check_model(g, m, (_t(g["in_images"]),))


def test_meant_vqa_tiny(golden):
    g = golden("meant_vqa_tiny")
    m = O.meant_vqa(128, 128, 4, 32, 32, 16, 1, 7, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=1, channels=4)
    _check_model(g, m, (_t(g["in_tweets"]), _t(g["in_images"]), _t(g["in_mask"])))


def test_meant_vision_c2_full_width(golden):
    g = golden("meant_vision_c2")
    r = np.random.RandomState(102)
    img = _t(r.standard_normal((2, 1, 4, 224, 224)).astype("float32"))
    m = O.meant_vision(768, 4, 224, 224, 16, 1, 2, num_heads=12, num_encoders=1, channels=4)
    _check_model(g, m, (img,), grad_rtol=1e-3)


def test_meant_full_c3(golden):
    """Full dims (lag 12, d 768, 12 heads, S 512, 224x224): outputs, loss, every grad norm."""
    g = golden("meant_full_c3")
    r = np.random.RandomState(99)
    ids = _t(r.randint(0, 2000, (2, 12, 512)).astype("int64"))
    img = _t(r.standard_normal((2, 12, 4, 224, 224)).astype("float32"))
    mask = torch.ones(2, 12, 512)
    mask[1, :, 400:] = 0
    m = O.meant(768, 768, 4, 224, 224, 16, 12, 2, torch.nn.Embedding(2000, 768), num_heads=12, num_encoders=1)
    _check_model(g, m, (ids, img, mask), grad_rtol=1e-3)
    np.testing.assert_allclose(g["out"], [[0.713732, 0.527026], [0.701181, 0.410244]], atol=1e-6)


def test_mlm_pretrainer_tiny(golden):
    """the oracle's restatement of pretrain_mlm.py:74-88 against the reference's own class (HF Roberta
    embeddings + lm_head around two languageEncoders), logits / CE over V with -100 ignored / gradients"""
    g = golden("mlm_pretrainer_tiny")
    torch.manual_seed(0)
    emb, head = O.mlm_parts()
    m = O.meant_language_pretrainer(2, 128, emb, head, text_dim=128, num_heads=2).eval()
    O.fill_weights_(m, 2468)
    out = m(torch.from_numpy(g["ids"]), attention_mask=torch.from_numpy(g["mask"]))
    loss = torch.nn.functional.cross_entropy(out.view(-1, 120), torch.from_numpy(g["labels"]).view(-1))
    loss.backward()
    assert (out.detach() - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    params = dict(m.named_parameters())
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(nm)].grad.double().norm().item()
        assert abs(got - refn) <= 1e-4 * max(refn, 1e-6), (nm, got, refn)


def test_mim_pretrainer_tiny(golden):
    """the oracle's restatement of pretrain_mim.py:77-99 against the reference's own class (patch embed,
    one visionEncoder, HF ViT masked-image-modelling decoder), output / L1 loss on the first 3 channels / gradients"""
    g = golden("mim_pretrainer_tiny")
    torch.manual_seed(0)
    m = O.meant_vision_pretrainer(1, O.mim_decoder(), 128, patch_res=16, channels=4, height=32, width=32, image_dim=128, num_heads=2).eval()
    O.fill_weights_(m, 1357)
    out = m(torch.from_numpy(g["images"]))
    loss = torch.nn.functional.l1_loss(out, torch.from_numpy(g["target"])[:, 0:3])
    loss.backward()
    assert (out.detach() - torch.from_numpy(g["out"])).abs().max().item() < 5e-6 * float(np.abs(g["out"]).max())   # outputs are O(30)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    params = dict(m.named_parameters())
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(nm)].grad.double().norm().item()
        assert abs(got - refn) <= 1e-4 * max(refn, 1e-6), (nm, got, refn)


TS_CFG = {"timesformer_tiny": dict(dim=128, num_frames=3, num_classes=5, seed=8642),
          "timesformer_tiny_mask": dict(dim=128, num_frames=3, num_classes=5, seed=8642),
          "timesformer_tiny_shift": dict(dim=192, num_frames=4, num_classes=3, seed=8643, shift_tokens=True),
          "timesformer_tiny_posemb": dict(dim=128, num_frames=3, num_classes=5, seed=8644, rotary_emb=False)}


@pytest.mark.parametrize("fixture", list(TS_CFG))
def test_timesformer_tiny(golden, fixture):
    """the oracle's divided space-time attention (cls token, frame + axial rotary, GEGLU; with and
    without the frame mask of :241-253) against the fork's TimeSformer (src/meant/timesformer_pytorch.py), tokens /
    logits / loss / gradients"""
    g = golden(fixture)
    torch.manual_seed(0)
    cfg = dict(TS_CFG[fixture])
    seed = cfg.pop("seed")
    m = O.TimeSformer(image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64, **cfg).eval()
    O.fill_weights_(m, seed)
    mask = torch.from_numpy(g["mask"]) if "mask" in g.files else None
    x = m.meant_forward(torch.from_numpy(g["video"]), mask=mask)
    logits = m.to_out(x[:, 0])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["target"])) + 0.01 * x.pow(2).mean()
    loss.backward()
    assert (x.detach() - torch.from_numpy(g["tokens"])).abs().max().item() < 5e-6 * float(np.abs(g["tokens"]).max())
    assert (logits.detach() - torch.from_numpy(g["logits"])).abs().max().item() < 1e-5 * max(1.0, float(np.abs(g["logits"]).max()))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    params = dict(m.named_parameters())
    for nm, refn in zip(g["grad_names"], g["grad_norms"]):
        got = params[str(nm)].grad.double().norm().item()
        assert abs(got - refn) <= 1e-4 * max(refn, 1e-6), (nm, got, refn)
