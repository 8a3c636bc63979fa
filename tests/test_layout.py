"""CPU-side checks: the C-ABI library builds, loads and exports exactly what include/meant_hip.h declares;
the product package never touches the oracle; state_dict keys match the reference's (via the fixtures)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "meant_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(meant_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_header_symbols():
    import __graft_entry__ as g
    g.build()
    import meant_amd
    from meant_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(meant_amd.lib, s), f"{s} declared in include/meant_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (meant_[a-z0-9_]+)", out))
    assert exported == set(syms), exported ^ set(syms)
    assert meant_amd.lib.meant_version() >= 100          # pure host call, no GPU needed


def test_attention_forward_workspace_leaves_out_the_backward_scratch():
    """the single-pass backward's partial-dQ scratch (G*H*64 KiB for 256 < S <= 512) and the backward's row statistics
    are not part of what a forward call allocates (pure host arithmetic, no GPU needed)"""
    import meant_amd
    lib = meant_amd.lib
    BF16, F32 = 1, 0
    G, S, H, Dh = 1536, 512, 12, 64                      # the text shape of the benchmark
    full, fwd = lib.meant_attn_ws(G, S, H, Dh, BF16), lib.meant_attn_fwd_ws(G, S, H, Dh, BF16)
    assert full >= G * H * 64 * 1024                     # the scratch is in the backward's size
    assert fwd < 16 << 20 and fwd < full                 # mask bias + tile flags only
    assert lib.meant_attn_fwd_ws(G, 196, H, Dh, BF16) <= lib.meant_attn_ws(G, 196, H, Dh, BF16)
    assert lib.meant_attn_fwd_ws(8, 64, 4, 80, BF16) == lib.meant_attn_ws(8, 64, 4, 80, BF16)     # the fp32 detour needs it all
    assert lib.meant_attn_fwd_ws(8, 64, 4, 64, F32) == lib.meant_attn_ws(8, 64, 4, 64, F32)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "meant_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions the oracle"
    for f in os.listdir(os.path.join(ROOT, "dropin", "meant")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "dropin", "meant", f)).read()


def test_no_cpu_fallback():
    import meant_amd
    from meant_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback|There is no CPU fallback"):
        ops.rmsnorm(torch.randn(2, 768), torch.ones(768))
    m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(10, 128), num_heads=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 16, dtype=torch.long), torch.zeros(1, 3, 4, 32, 32), torch.ones(1, 3, 16))


@pytest.mark.parametrize("fixture,cls,args,emb,kw", [
    ("meant_tiny", "meant", (128, 128, 4, 32, 32, 16, 3, 2), (100, 128), dict(num_heads=2, num_encoders=1, channels=4)),
    ("meant_tiny_e2", "meant", (128, 192, 4, 32, 48, 16, 2, 3), (50, 128), dict(num_heads=2, num_encoders=2, channels=4)),
    ("meant_tweet_c1", "meant_tweet", (128, 4, 1, 2), (1000, 128), dict(num_heads=2, num_encoders=1)),
    ("meant_vision_tiny", "meant_vision", (128, 4, 32, 32, 16, 3, 2), None, dict(num_heads=2, num_encoders=1, channels=4)),
    ("meant_vqa_tiny", "meant_vqa", (128, 128, 4, 32, 32, 16, 1, 7), (100, 128), dict(num_heads=2, num_encoders=1, channels=4)),
])
def test_parameter_names_match_reference(golden, fixture, cls, args, emb, kw):
    """grad_names in each fixture are the reference model's own named_parameters() (those that got a
    gradient); the native class must expose the same names, shapes being checked by load_state_dict."""
    import meant_amd
    from oracle import meant_oracle as O
    g = golden(fixture)
    a = list(args) + ([torch.nn.Embedding(*emb)] if emb else [])
    hip = getattr(meant_amd, cls)(*a, **kw)
    names = {k for k, p in hip.named_parameters() if p.requires_grad}
    ref_names = {str(n) for n in g["grad_names"]}
    assert ref_names <= names, ref_names - names
    b = list(args) + ([torch.nn.Embedding(*emb)] if emb else [])
    ref = getattr(O, cls)(*b, **kw)
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys()) or set(hip.state_dict()) == set(ref.state_dict())
    hip.load_state_dict(ref.state_dict())


def test_dropin_packages_resolve():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        for mod in [m for m in list(sys.modules) if m == "meant" or m.startswith("meant.") or m == "utils" or m.startswith("utils.")]:
            del sys.modules[mod]
        import meant as pkg
        from meant import meant, meant_vision, meant_tweet, meant_vqa, temporal, attention, xPosAttention, languageEncoder, visionEncoder  # noqa
        from utils import RMSNorm  # noqa
        mm = sys.modules["meant.meant"]                      # where reference-saved pickles look the class up
        assert mm.meant is pkg.meant and isinstance(pkg.meant, type)
        assert type(languageEncoder(128, 2).encode[2]).__name__ == "xPosAttention"     # meant/meant.py:112 dispatch
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))
        for mod in [m for m in list(sys.modules) if m == "meant" or m.startswith("meant.") or m == "utils" or m.startswith("utils.")]:
            del sys.modules[mod]


def test_norm_linear_fold_rule(monkeypatch):
    """the RMSNorm-into-Linear fold is decided by the size of the stack in flight unless forced (ops.fold_wanted): it is even
    to slightly behind on time and saves one [tokens, d] tensor per layer, so only stacks where that adds up take it"""
    from meant_amd import ops
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", None)
    monkeypatch.setattr(ops, "_stack_bytes", 0)
    assert not ops.fold_wanted()
    ops.set_stack_hint(1, 128 * 12 * 512, 768)               # the headline step's text stack (lag 12)
    assert not ops.fold_wanted()
    ops.set_stack_hint(12, 64 * 512, 768)                    # the MLM pretrainer: deep, few tokens
    assert not ops.fold_wanted()
    ops.set_stack_hint(12, 32 * 12 * 512, 768)               # 12 layers at 32 samples per GPU: fits as it is
    assert not ops.fold_wanted()
    ops.set_stack_hint(12, 128 * 12 * 512, 768)              # 12 layers at 128 per GPU: text stack
    assert ops.fold_wanted()
    ops.set_stack_hint(12, 128 * 12 * 196, 768)              # ... and its vision stack
    assert ops.fold_wanted()
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", False)      # MEANT_FUSE_NORM_LINEAR=0
    assert not ops.fold_wanted()
    ops.set_stack_hint(1, 64, 64)
    monkeypatch.setattr(ops, "FUSE_NORM_LINEAR", True)       # MEANT_FUSE_NORM_LINEAR=1
    assert ops.fold_wanted()


def test_scheduling_switches_host_logic():
    """host-side rules of the step's scheduling (no GPU): which embedding gradients are summed in sorted-id order (and therefore have
    their ids sorted in forward), what a stack's own-stream pooling leaves alone, the language stack's stream under a multi-process launch"""
    import subprocess
    import sys
    import torch
    from meant_amd import modules, ops
    assert ops._emb_sorted_bwd_ok(786432, 768) and ops._emb_sorted_bwd_ok(4096, 768)
    assert not ops._emb_sorted_bwd_ok(4095, 768)              # few tokens: the float-atomics kernel (unless option deterministic)
    assert not ops._emb_sorted_bwd_ok(786432, 2048) and not ops._emb_sorted_bwd_ok(786432, 772)
    toks = torch.zeros(2, 5, 8)
    assert modules._pool_own_stream(torch.float32, toks) is toks                    # a token tensor: pooled later, the literal way
    part = (torch.zeros(2, 5, 8), torch.zeros(2, 5, 8), torch.zeros(8, 8), None)
    assert modules._pool_own_stream(torch.float32, part) is part                    # (h, x, W, b) tokens: pool_linear_cat's job
    code = "import meant_amd.modules as m; print(int(m.LANG_PRIORITY))"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("MEANT_LANG_PRIORITY", None)
    for ws, want in (("1", "1"), ("8", "0")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(env, WORLD_SIZE=ws), capture_output=True, text=True, cwd=ROOT)
        assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == want, (ws, out.stdout, out.stderr[-300:])
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, WORLD_SIZE="8", MEANT_LANG_PRIORITY="1"), capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip().splitlines()[-1] == "1"


def test_rotary_tables_match_golden(golden):
    """host-side table builder (CPU part of the rotary path) against the reference's rotated vectors"""
    import meant_amd
    g = golden("rotary_xpos48")
    rot = meant_amd.RotaryEmbedding(dim=48, use_xpos=True)
    np.testing.assert_allclose(rot.freqs.detach().numpy(), g["freqs"], rtol=1e-6)
    np.testing.assert_allclose(rot.scale.numpy(), g["scale"], rtol=1e-6)
    for S in (16, 512):
        qa, qb, ka, kb = rot.tables(S, "cpu")
        q = torch.from_numpy(g[f"q{S}"])
        h = q[..., :48]
        sw = torch.stack((-h[..., 1::2], h[..., 0::2]), -1).reshape(h.shape)
        np.testing.assert_allclose((h * qa + sw * qb).numpy(), g[f"rq{S}"][..., :48], atol=2e-5)
        k = torch.from_numpy(g[f"k{S}"])[..., :48]
        sw = torch.stack((-k[..., 1::2], k[..., 0::2]), -1).reshape(k.shape)
        np.testing.assert_allclose((k * ka + sw * kb).numpy(), g[f"rk{S}"][..., :48], atol=2e-5)


def test_cosine_warm_restarts_matches_torch_over_two_periods():
    """meant_amd.train.CosineWarmRestarts against torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(T_0 = 7, eta_min) stepped
    once per epoch (in_loop_train.py:547-567): same learning rate at every epoch through two restarts, and after a
    state_dict() -> load_state_dict() into a fresh scheduler"""
    from meant_amd.train import CosineWarmRestarts

    class Opt:
        lr = 5e-5
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=5e-5)
    ref = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ref_opt, T_0=7, T_mult=1, eta_min=1e-6)
    mine = CosineWarmRestarts(Opt(), T_0=7, eta_min=1e-6)
    for epoch in range(1, 16):
        ref_opt.step(); ref.step(); mine.step()
        assert abs(mine.opt.lr - ref_opt.param_groups[0]["lr"]) < 1e-12, epoch
        if epoch in (7, 14):
            assert abs(mine.opt.lr - 5e-5) < 1e-15          # the warm restart
        if epoch == 9:
            again = CosineWarmRestarts(Opt(), T_0=3, eta_min=0.0)
            again.load_state_dict(mine.state_dict())
            assert again.epoch == 9 and again.opt.lr == mine.opt.lr and again.lr_at(12) == mine.lr_at(12)
