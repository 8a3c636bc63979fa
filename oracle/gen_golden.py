#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  This script contains no reference source; it tells Python where
the reference's files are (/root/reference), imports its hot-path modules with the
loader shim of SURVEY.md appendix A, feeds them deterministic numpy-generated weights
and inputs (oracle.meant_oracle.fill_weights_ -- the recipe, not the arithmetic, is
shared with the oracle) and stores inputs / outputs / gradients as small fixtures.
The reference never travels: only the .npz data and this script are committed.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

It refuses to run when /root/reference is absent (e.g. on the GPU box).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden.py: /root/reference not present; fixtures are generated in the build container only")
    _load("rotary_embedding_torch", f"{REF}/meant/rotary_embedding_torch.py")     # vendored copy stands in for pip pkg
    rms = _load("_ref_rms_norm", f"{REF}/utils/rms_norm.py")
    u = types.ModuleType("utils"); u.RMSNorm = rms.RMSNorm; sys.modules["utils"] = u
    fa = types.ModuleType("flash_attn"); fa.flash_attn_func = fa.flash_attn_qkvpacked_func = None
    sys.modules["flash_attn"] = fa
    pkg = types.ModuleType("meant"); pkg.__path__ = [f"{REF}/meant"]; sys.modules["meant"] = pkg
    for n in ["attention", "xPosAttention", "temporal", "flash_attention", "xPosAttention_flash"]:
        _load(f"meant.{n}", f"{REF}/meant/{n}.py")
    ns = types.SimpleNamespace()
    ns.rms = rms
    ns.rot = sys.modules["rotary_embedding_torch"]
    ns.attention = sys.modules["meant.attention"]
    ns.xpos = sys.modules["meant.xPosAttention"]
    ns.temporal = sys.modules["meant.temporal"]
    ns.meant = _load("meant.meant", f"{REF}/meant/meant.py")
    ns.vision = _load("meant.meant_vision", f"{REF}/meant/meant_vision.py")
    ns.vqa = _load("meant.meant_vqa", f"{REF}/meant/meant_vqa.py")
    ns.tweet = _load("meant.meant_tweet", f"{REF}/meant/meant_tweet.py")
    ns.tweet.languageEncoder = ns.meant.languageEncoder        # meant_tweet.py:81 NameError work-around
    return ns


def _np(t):
    return t.detach().cpu().numpy()


def grads_of(model):
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is not None:
            names.append(k)
            norms.append(float(p.grad.double().norm()))
    return names, np.asarray(norms, dtype=np.float64)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)/1024:.1f} KiB)")


def model_case(name, model, inputs, target, full_grads=(), store_inputs=True):
    """inputs: dict of numpy arrays in forward-argument order."""
    from oracle.meant_oracle import fill_weights_
    model.eval()
    fill_weights_(model, 1234)
    args = [torch.from_numpy(v) for v in inputs.values()]
    out = model(*args)
    loss = torch.nn.functional.cross_entropy(out, torch.from_numpy(target))
    loss.backward()
    names, norms = grads_of(model)
    arrs = dict(out=_np(out), loss=np.float64(loss.item()), grad_names=np.array(names), grad_norms=norms,
                target=target)
    if store_inputs:
        for k, v in inputs.items():
            arrs["in_" + k] = v
    params = dict(model.named_parameters())
    for k in full_grads:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save(name, **arrs)


def gen_rmsnorm_partial(R):
    """the partial / bias forms of the reference's RMSNorm class (utils/rms_norm.py:44-57): not used by any model, pinned all the same"""
    rs = np.random.RandomState(11)
    arrs = {}
    for tag, d, p, bias in (("a", 256, 0.5, True), ("b", 768, 0.3, False), ("c", 128, 1.0, True), ("d", 64, -1.0, True)):
        x = rs.standard_normal((2, 3, d)).astype("float32")
        g = (1 + 0.1 * rs.standard_normal(d)).astype("float32")
        off = (0.2 * rs.standard_normal(d)).astype("float32")
        dy = rs.standard_normal((2, 3, d)).astype("float32")
        n = R.rms.RMSNorm(d, p=p, bias=bias)
        with torch.no_grad():
            n.scale.copy_(torch.from_numpy(g))
            if bias:
                n.offset.copy_(torch.from_numpy(off))
        xt = torch.from_numpy(x).requires_grad_()
        y = n(xt)
        y.backward(torch.from_numpy(dy))
        arrs.update({f"{tag}_x": x, f"{tag}_scale": g, f"{tag}_offset": off, f"{tag}_dy": dy, f"{tag}_y": _np(y), f"{tag}_dx": _np(xt.grad),
                     f"{tag}_dscale": _np(n.scale.grad), f"{tag}_doffset": _np(n.offset.grad) if bias else np.zeros(d, "float32"),
                     f"{tag}_cfg": np.array([d, p, float(bias)], dtype="float64")})
    save("rmsnorm_partial_bias", **arrs)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = load_reference()

    # ---------------- leaf KATs ----------------
    print("leaf KATs")
    rs = np.random.RandomState(7)
    x = rs.standard_normal((3, 17, 768)).astype("float32")
    g = (1 + 0.1 * rs.standard_normal(768)).astype("float32")
    dy = rs.standard_normal((3, 17, 768)).astype("float32")
    n = R.rms.RMSNorm(768)
    with torch.no_grad():
        n.scale.copy_(torch.from_numpy(g))
    xt = torch.from_numpy(x).requires_grad_()
    y = n(xt); y.backward(torch.from_numpy(dy))
    save("rmsnorm_768", x=x, scale=g, dy=dy, y=_np(y), dx=_np(xt.grad), dscale=_np(n.scale.grad),
         kat_in=np.array([1, 2, 3, 4], dtype="float32"),
         kat_out=_np(R.rms.RMSNorm(4)(torch.tensor([1., 2, 3, 4]))))

    gen_rmsnorm_partial(R)

    # rotary: xPos (dim 48 of Dh 64, S=512 and S=16) and pixel (dim 32 and 48; N=196)
    xp = R.rot.RotaryEmbedding(dim=48, use_xpos=True)
    arrs = {}
    for S in (16, 64, 512):
        q = rs.standard_normal((1, 2, S, 64)).astype("float32")
        k = rs.standard_normal((1, 2, S, 64)).astype("float32")
        xp.cache.clear()
        rq, rk = xp.rotate_queries_and_keys(torch.from_numpy(q), torch.from_numpy(k))
        arrs.update({f"q{S}": q, f"k{S}": k, f"rq{S}": _np(rq), f"rk{S}": _np(rk)})
    arrs["freqs"] = _np(xp.freqs); arrs["scale"] = _np(xp.scale)
    save("rotary_xpos48", **arrs)
    arrs = {}
    for dim, N, Dh in ((32, 196, 64), (48, 196, 96), (32, 4, 64)):
        px = R.rot.RotaryEmbedding(dim=dim, freqs_for="pixel")
        t = rs.standard_normal((1, 2, N, Dh)).astype("float32")
        arrs.update({f"t_{dim}_{N}": t, f"r_{dim}_{N}": _np(px.rotate_queries_or_keys(torch.from_numpy(t))),
                     f"freqs_{dim}": _np(px.freqs)})
    save("rotary_pixel", **arrs)

    # attention modules, fwd + bwd, small but with the real head geometry (Dh=64)
    from oracle.meant_oracle import fill_weights_
    def attn_case(name, mod, x, dy, mask=None):
        mod.eval(); fill_weights_(mod, 4321)
        xt = torch.from_numpy(x).requires_grad_()
        y = mod(xt, torch.from_numpy(mask)) if mask is not None else mod(xt)
        y.backward(torch.from_numpy(dy))
        names, norms = grads_of(mod)
        arrs = dict(x=x, dy=dy, y=_np(y), dx=_np(xt.grad), grad_names=np.array(names), grad_norms=norms)
        for k, p in mod.named_parameters():
            if p.grad is not None:                      # big matrices: first 4 rows (+ the norm above)
                arrs["grad__" + k] = _np(p.grad if p.grad.numel() <= 4096 else p.grad[:4])
        if mask is not None:
            arrs["mask"] = mask
        save(name, **arrs)

    H, d = 2, 128
    x = rs.standard_normal((2, 196, d)).astype("float32"); dy = rs.standard_normal((2, 196, d)).astype("float32")
    attn_case("attention_h2_d128_n196", R.attention.attention(H, d, R.rot.RotaryEmbedding(dim=d // H // 2, freqs_for="pixel")), x, dy)
    x = rs.standard_normal((3, 80, d)).astype("float32"); dy = rs.standard_normal((3, 80, d)).astype("float32")
    mask = np.ones((3, 80), dtype="float32"); mask[1, 50:] = 0; mask[2, 1:] = 0
    attn_case("xposattention_h2_d128_s80", R.xpos.xPosAttention(H, d, R.rot.RotaryEmbedding(dim=48, use_xpos=True)), x, dy, mask)
    x = rs.standard_normal((2, 512, d)).astype("float32"); dy = rs.standard_normal((2, 512, d)).astype("float32")
    mask = np.ones((2, 512), dtype="float32"); mask[1, 400:] = 0
    attn_case("xposattention_h2_d128_s512", R.xpos.xPosAttention(H, d, R.rot.RotaryEmbedding(dim=48, use_xpos=True)), x, dy, mask)
    x = rs.standard_normal((3, 12, 1536)).astype("float32"); dy = rs.standard_normal((3, 1, 1536)).astype("float32")
    attn_case("temporal_h12_d1536_l12", R.temporal.temporal(12, 1536), x, dy)

    # ---------------- model KATs ----------------
    print("model KATs (tiny)")
    emb = lambda V, dm: torch.nn.Embedding(V, dm)
    r = np.random.RandomState(99)
    ids = r.randint(0, 100, (2, 3, 16)).astype("int64")
    img = r.standard_normal((2, 3, 4, 32, 32)).astype("float32")
    mask = np.ones((2, 3, 16), dtype="float32"); mask[1, :, 12:] = 0
    m = R.meant.meant(128, 128, 4, 32, 32, 16, 3, 2, emb(100, 128), num_heads=2, num_encoders=1, channels=4)
    model_case("meant_tiny", m, dict(tweets=ids, images=img, mask=mask), np.array([0, 1]),
               full_grads=["patchEmbed.1.weight", "languageEncoders.0.encode.2.v.weight",
                           "visionEncoders.0.encode.2.q.weight", "temporal_encoding.0.temp_embedding",
                           "mlpHead.0.scale", "languageEncoders.0.encode2.3.scale",
                           "visionEncoders.0.encode.1.bias"])
    # two encoder layers, 3 classes
    r = np.random.RandomState(100)
    ids = r.randint(0, 50, (3, 2, 24)).astype("int64")
    img = r.standard_normal((3, 2, 4, 32, 48)).astype("float32")
    mask = np.ones((3, 2, 24), dtype="float32"); mask[0, :, 20:] = 0; mask[2, 1, 5:] = 0
    m = R.meant.meant(128, 192, 4, 32, 48, 16, 2, 3, emb(50, 128), num_heads=2, num_encoders=2, channels=4)
    model_case("meant_tiny_e2", m, dict(tweets=ids, images=img, mask=mask), np.array([2, 0, 1]),
               full_grads=["languageEncoders.1.encode.2.q.weight", "visionEncoders.1.encode2.4.weight"])

    # C1: meant_tweet lag=1 d=128 H=2 S=64 B=4
    r = np.random.RandomState(99)
    ids = r.randint(0, 1000, (4, 1, 64)).astype("int64")
    mask = np.ones((4, 1, 64), dtype="float32"); mask[1, :, 56:] = 0; mask[3, :, 56:] = 0
    m = R.tweet.meant_tweet(128, 4, 1, 2, emb(1000, 128), num_heads=2, num_encoders=1)
    model_case("meant_tweet_c1", m, dict(tweets=ids, mask=mask), np.array([0, 1, 1, 0]),
               full_grads=["languageEncoders.0.encode.2.k.weight", "mlpHead.0.weight",
                           "temporal_encoding.0.temp_encode.1.multi_mad.weight"])

    # meant_vision tiny (lag 3) and C2-shaped (lag 1, d=768, H=12, B=2)
    r = np.random.RandomState(101)
    img = r.standard_normal((2, 3, 4, 32, 32)).astype("float32")
    m = R.vision.meant_vision(128, 4, 32, 32, 16, 3, 2, num_heads=2, num_encoders=1, channels=4)
    model_case("meant_vision_tiny", m, dict(images=img), np.array([1, 0]),
               full_grads=["patchEmbed.1.weight", "temporal_encoding.0.temp_encode.1.q.weight"])
    r = np.random.RandomState(102)
    img = r.standard_normal((2, 1, 4, 224, 224)).astype("float32")
    m = R.vision.meant_vision(768, 4, 224, 224, 16, 1, 2, num_heads=12, num_encoders=1, channels=4)
    model_case("meant_vision_c2", m, dict(images=img), np.array([1, 0]), store_inputs=False)

    # meant_vqa tiny
    r = np.random.RandomState(103)
    ids = r.randint(0, 100, (3, 20)).astype("int64")
    img = r.standard_normal((3, 4, 32, 32)).astype("float32")
    mask = np.ones((3, 20), dtype="float32"); mask[2, 11:] = 0
    m = R.vqa.meant_vqa(128, 128, 4, 32, 32, 16, 1, 7, emb(100, 128), num_heads=2, num_encoders=1, channels=4)
    model_case("meant_vqa_tiny", m, dict(tweets=ids, images=img, mask=mask), np.array([3, 6, 0]),
               full_grads=["mlpHead.1.weight"])

    # C3 full dims (B=2, V=2000 to keep the fixture generation light): outputs / loss / grad norms only
    print("model KAT (full dims, takes ~10 s)")
    r = np.random.RandomState(99)
    ids = r.randint(0, 2000, (2, 12, 512)).astype("int64")
    img = r.standard_normal((2, 12, 4, 224, 224)).astype("float32")
    mask = np.ones((2, 12, 512), dtype="float32"); mask[1, :, 400:] = 0
    m = R.meant.meant(768, 768, 4, 224, 224, 16, 12, 2, emb(2000, 768), num_heads=12, num_encoders=1)
    model_case("meant_full_c3", m, dict(tweets=ids, images=img, mask=mask), np.array([0, 1]), store_inputs=False)


def gen_mlm(R):
    """MLM pretrainer KAT: the reference's own class from pretrain_mlm.py, imported with its plotting /
    logging dependencies stubbed, 2 encoder layers, d=128, 2 heads (Dh=64), V=120, S=24"""
    import types
    from oracle.meant_oracle import fill_weights_, mlm_parts
    sys.modules.pop("flash_attn", None)      # the stub has served the reference's imports; transformers probes the real name
    tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.modules["meant"].languageEncoder = R.meant.languageEncoder          # `from meant import languageEncoder` (:44)
    sys.modules["utils"].mlm_dataset = object                               # `from utils import mlm_dataset` (:45)
    ref = _load("_ref_pretrain_mlm", f"{REF}/pretrain_mlm.py")
    torch.manual_seed(0)
    emb, head = mlm_parts()
    m = ref.meant_language_pretrainer(2, 128, emb, head, text_dim=128, num_heads=2).eval()
    fill_weights_(m, 2468)
    r = np.random.RandomState(104)
    ids = r.randint(2, 120, (3, 24)).astype("int64")
    mask = np.ones((3, 24), dtype="float32"); mask[1, 18:] = 0; mask[2, 9:] = 0
    labels = np.full((3, 24), -100, dtype="int64")
    pick = r.rand(3, 24) < 0.25
    labels[pick] = r.randint(2, 120, int(pick.sum()))
    out = m(torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))
    loss = torch.nn.CrossEntropyLoss()(out.view(-1, 120), torch.from_numpy(labels).view(-1))     # pretrain_mlm.py:160,178
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(ids=ids, mask=mask, labels=labels, logits=_np(out), loss=np.array(loss.item(), dtype="float64"),
                grad_names=np.array(names), grad_norms=norms)
    for k in ["mlm_head.dense.weight", "mlm_head.bias", "mlm_head.layer_norm.weight", "languageEncoders.1.encode.2.q.weight",
              "embedding.0.word_embeddings.weight", "embedding.0.LayerNorm.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("mlm_pretrainer_tiny", **arrs)


def gen_mim(R):
    """MIM pretrainer KAT: the reference's own class from pretrain_mim.py, d=128, 2 heads, 32x32 images
    with 4 channels, HF ViTForMaskedImageModeling decoder (1x1 conv + PixelShuffle) with stride 16"""
    import types
    from oracle.meant_oracle import fill_weights_, mim_decoder
    sys.modules.pop("flash_attn", None)
    tb = types.ModuleType("torch.utils.tensorboard"); tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    sys.modules["meant"].visionEncoder = R.meant.visionEncoder
    sys.modules["meant"].languageEncoder = R.meant.languageEncoder
    sys.modules["utils"].mlm_dataset = object
    sys.modules["utils"].mim_dataset = object
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))                  # data-set I/O only (pretrain_mim.py:48)
    ref = _load("_ref_pretrain_mim", f"{REF}/pretrain_mim.py")
    torch.manual_seed(0)
    m = ref.meant_vision_pretrainer(1, mim_decoder(), 128, patch_res=16, channels=4, height=32, width=32, image_dim=128, num_heads=2).eval()
    fill_weights_(m, 1357)
    r = np.random.RandomState(105)
    img = r.standard_normal((3, 4, 32, 32)).astype("float32")
    tgt = r.standard_normal((3, 4, 32, 32)).astype("float32")
    out = m(torch.from_numpy(img))
    loss = torch.nn.L1Loss()(out, torch.from_numpy(tgt)[:, 0:3])                 # pretrain_mim.py:162,204
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(images=img, target=tgt, out=_np(out), loss=np.array(loss.item(), dtype="float64"), grad_names=np.array(names), grad_norms=norms)
    for k in ["decoder.0.bias", "visionEncoders.0.encode.2.q.weight", "patchEmbed.1.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("mim_pretrainer_tiny", **arrs)


def gen_timesformer(R):
    """divided space-time attention KAT: the fork's TimeSformer (src/meant/timesformer_pytorch.py),
    dim=128, 3 frames, 32x32 images of 4 channels with 16x16 patches, depth 2, 2 heads of 64"""
    import types
    from oracle.meant_oracle import fill_weights_
    for name in ("src", "src.utils", "src.meant"):
        if name not in sys.modules:
            pkg = types.ModuleType(name); pkg.__path__ = []; sys.modules[name] = pkg
    _load("src.utils.rotary", f"{REF}/src/utils/rotary.py")
    ref = _load("src.meant.timesformer_pytorch", f"{REF}/src/meant/timesformer_pytorch.py")
    torch.manual_seed(0)
    m = ref.TimeSformer(dim=128, num_frames=3, num_classes=5, image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64).eval()
    fill_weights_(m, 8642)
    r = np.random.RandomState(106)
    video = r.standard_normal((2, 3, 4, 32, 32)).astype("float32")
    x = m.meant_forward(torch.from_numpy(video))
    logits = m.to_out(x[:, 0])
    tgt = np.array([1, 4])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(tgt)) + 0.01 * x.pow(2).mean()
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(video=video, target=tgt, tokens=_np(x), logits=_np(logits), loss=np.array(loss.item(), dtype="float64"),
                grad_names=np.array(names), grad_norms=norms)
    for k in ["cls_token", "layers.0.0.fn.to_qkv.weight", "layers.1.1.fn.to_out.0.bias", "layers.0.2.fn.net.0.bias", "to_patch_embedding.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("timesformer_tiny", **arrs)


def gen_timesformer_shift(R):
    """the same with shift_tokens=True (PreTokenShift, src/meant/timesformer_pytorch.py:28-53,196-199), dim 192 so that the
    thirds are 64 features wide, 4 frames"""
    import types
    from oracle.meant_oracle import fill_weights_
    for name in ("src", "src.utils", "src.meant"):
        if name not in sys.modules:
            pkg = types.ModuleType(name); pkg.__path__ = []; sys.modules[name] = pkg
    _load("src.utils.rotary", f"{REF}/src/utils/rotary.py")
    ref = _load("src.meant.timesformer_pytorch", f"{REF}/src/meant/timesformer_pytorch.py")
    torch.manual_seed(0)
    m = ref.TimeSformer(dim=192, num_frames=4, num_classes=3, image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64,
                        shift_tokens=True).eval()
    fill_weights_(m, 8643)
    r = np.random.RandomState(108)
    video = r.standard_normal((2, 4, 4, 32, 32)).astype("float32")
    x = m.meant_forward(torch.from_numpy(video))
    logits = m.to_out(x[:, 0])
    tgt = np.array([2, 0])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(tgt)) + 0.01 * x.pow(2).mean()
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(video=video, target=tgt, tokens=_np(x), logits=_np(logits), loss=np.array(loss.item(), dtype="float64"),
                grad_names=np.array(names), grad_norms=norms)
    for k in ["cls_token", "layers.0.0.fn.fn.to_qkv.weight", "layers.1.2.fn.fn.net.0.bias", "to_patch_embedding.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("timesformer_tiny_shift", **arrs)


def gen_timesformer_posemb(R):
    """rotary_emb=False: learned positional embedding added to the tokens (src/meant/timesformer_pytorch.py:186,220-221)"""
    import types
    from oracle.meant_oracle import fill_weights_
    for name in ("src", "src.utils", "src.meant"):
        if name not in sys.modules:
            pkg = types.ModuleType(name); pkg.__path__ = []; sys.modules[name] = pkg
    _load("src.utils.rotary", f"{REF}/src/utils/rotary.py")
    ref = _load("src.meant.timesformer_pytorch", f"{REF}/src/meant/timesformer_pytorch.py")
    torch.manual_seed(0)
    m = ref.TimeSformer(dim=128, num_frames=3, num_classes=5, image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64,
                        rotary_emb=False).eval()
    fill_weights_(m, 8644)
    r = np.random.RandomState(109)
    video = r.standard_normal((2, 3, 4, 32, 32)).astype("float32")
    x = m.meant_forward(torch.from_numpy(video))
    logits = m.to_out(x[:, 0])
    tgt = np.array([3, 1])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(tgt)) + 0.01 * x.pow(2).mean()
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(video=video, target=tgt, tokens=_np(x), logits=_np(logits), loss=np.array(loss.item(), dtype="float64"),
                grad_names=np.array(names), grad_norms=norms)
    for k in ["cls_token", "pos_emb.weight", "layers.0.0.fn.to_qkv.weight", "to_patch_embedding.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("timesformer_tiny_posemb", **arrs)


def gen_timesformer_mask(R):
    """the same with a frame mask (src/meant/timesformer_pytorch.py:241-253: `mask` [b, f] bool hides whole frames from the
    time attention's keys and from the cls query; the space attention only masks the cls query): video 0 loses its last
    frame, video 1 its last two"""
    import types
    from oracle.meant_oracle import fill_weights_
    for name in ("src", "src.utils", "src.meant"):
        if name not in sys.modules:
            pkg = types.ModuleType(name); pkg.__path__ = []; sys.modules[name] = pkg
    _load("src.utils.rotary", f"{REF}/src/utils/rotary.py")
    ref = _load("src.meant.timesformer_pytorch", f"{REF}/src/meant/timesformer_pytorch.py")
    torch.manual_seed(0)
    m = ref.TimeSformer(dim=128, num_frames=3, num_classes=5, image_size=32, patch_size=16, channels=4, depth=2, heads=2, dim_head=64).eval()
    fill_weights_(m, 8642)
    r = np.random.RandomState(107)
    video = r.standard_normal((2, 3, 4, 32, 32)).astype("float32")
    mask = np.array([[True, True, False], [True, False, False]])
    x = m.meant_forward(torch.from_numpy(video), mask=torch.from_numpy(mask))
    logits = m.to_out(x[:, 0])
    tgt = np.array([2, 0])
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(tgt)) + 0.01 * x.pow(2).mean()
    loss.backward()
    names, norms = grads_of(m)
    params = dict(m.named_parameters())
    arrs = dict(video=video, mask=mask, target=tgt, tokens=_np(x), logits=_np(logits), loss=np.array(loss.item(), dtype="float64"),
                grad_names=np.array(names), grad_norms=norms)
    for k in ["cls_token", "layers.0.0.fn.to_qkv.weight", "layers.1.1.fn.to_out.0.bias", "layers.0.2.fn.net.0.bias", "to_patch_embedding.bias"]:
        g = params[k].grad
        arrs["grad__" + k] = _np(g if g.numel() <= 4096 else g[:4])
    save("timesformer_tiny_mask", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        torch.set_num_threads(8)
        {"rmsnorm_partial": gen_rmsnorm_partial, "mlm": gen_mlm, "mim": gen_mim, "timesformer": gen_timesformer, "timesformer_mask": gen_timesformer_mask, "timesformer_shift": gen_timesformer_shift, "timesformer_posemb": gen_timesformer_posemb}[sys.argv[2]](load_reference())
    else:
        main()
        R_ = load_reference()
        gen_mlm(R_)
        gen_mim(R_)
        gen_timesformer(R_)
        gen_timesformer_mask(R_)
        gen_timesformer_shift(R_)
        gen_timesformer_posemb(R_)
