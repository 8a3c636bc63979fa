"""Drop-in proof at the level of the reference's training loop: the literal call sequence of in_loop_train.py:213-239 (fp16
autocast, fp16 pixels, nn.CrossEntropyLoss on the probabilities, optimizer.zero_grad(), GradScaler.scale(loss).backward(),
clip_grad_norm_(1.0) on the still-scaled gradients, scaler.step(torch.optim.AdamW), scaler.update()) executed on the
`dropin/meant` package, compared with the CPU oracle doing the same arithmetic in fp32; and the encoder transplant of
in_loop_train.py:503-504."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def dropin_meant():
    """`from meant import meant` resolved by <repo>/dropin, as a maintainer of the reference would set PYTHONPATH"""
    path = os.path.join(ROOT, "dropin")
    stale = lambda: [m for m in list(sys.modules) if m in ("meant", "utils") or m.startswith("meant.") or m.startswith("utils.")]
    for mod in stale():
        del sys.modules[mod]
    sys.path.insert(0, path)
    try:
        import meant as pkg
        yield pkg
    finally:
        sys.path.remove(path)
        for mod in stale():
            del sys.modules[mod]


ARGS = dict(text_dim=128, image_dim=128, price_dim=4, height=32, width=32, patch_res=16, lag=3, num_classes=2)
LR, SCALE = 1e-3, 65536.0            # GradScaler's initial scale (torch.cuda.amp.GradScaler(), in_loop_train.py:202)


def _data(steps, B=4):
    rs = np.random.RandomState(3)
    out = []
    for _ in range(steps):
        tweets = torch.from_numpy(rs.randint(0, 100, (B, 3, 16)).astype("int64"))
        graphs = torch.from_numpy(rs.standard_normal((B, 3, 4, 32, 32)).astype("float32"))
        masks = torch.ones(B, 3, 16)
        masks[1, :, 11:] = 0
        target = torch.from_numpy(rs.randint(0, 2, (B,)).astype("int64"))
        out.append((graphs, tweets, masks, target))
    return out


def test_reference_train_loop_sequence_on_the_dropin(dev, dropin_meant):
    from oracle import meant_oracle as O
    steps = 3
    ref = O.meant(*ARGS.values(), torch.nn.Embedding(100, 128), num_heads=2, num_encoders=1, channels=4)
    O.fill_weights_(ref, 1234)
    ref.eval()                                   # dropout off on both sides: the oracle cannot share the device's masks
    model = dropin_meant.meant(embedding=torch.nn.Embedding(100, 128), flash=False, num_heads=2, num_encoders=1, **ARGS)
    model.load_state_dict(ref.state_dict())
    model = model.to(dev).eval()
    start = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    device, torch_dtype = dev, torch.float16
    loss_fct = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.AdamW(model.parameters(), lr=LR)
    scaler = torch.amp.GradScaler("cuda")
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=LR)
    losses, ref_losses = [], []
    for graphs, tweets, attention_masks, target in _data(steps):
        # ---- in_loop_train.py:215-239, verbatim up to `self.` ----
        with torch.autocast(device_type="cuda", dtype=torch_dtype):
            out = model.forward(tweets.long().to(device), graphs.to(torch_dtype).to(device), attention_mask=attention_masks.cuda())
            assert not torch.isnan(out).any()
            loss = loss_fct(out, target.to(device).long())
        optimizer.zero_grad()
        scaler.scale(loss).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        scaler.step(optimizer)
        scaler.update()
        losses.append(loss.item())
        # ---- the same arithmetic in fp32 on the CPU, without the scaler object: gradients of SCALE * loss, clipped while
        # still scaled (the reference never calls unscale_), divided by SCALE as scaler.step does, AdamW
        rout = ref(tweets.long(), graphs.to(torch_dtype).float(), attention_masks)
        rloss = loss_fct(rout, target.long())
        ref_opt.zero_grad()
        (rloss * SCALE).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=1.0)
        for p in ref.parameters():
            if p.grad is not None:
                p.grad.div_(SCALE)
        ref_opt.step()
        ref_losses.append(rloss.item())
    assert scaler.get_scale() == SCALE                       # no step was skipped: the scaled bf16 gradients stayed finite
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-2, (losses, ref_losses)
    # parameters after 3 steps: the UPDATES (p - p0) against the oracle's.  After the clip + unscale the gradients are ~1e-5
    # of their size, so Adam's eps matters and an update is lr * g / (|g| + eps)-shaped: compare direction and size
    hip_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref_sd = ref.state_dict()
    checked = 0
    for k, p0 in start.items():
        if not p0.is_floating_point() or k.endswith("freqs") or k.endswith("xPos.scale"):
            continue
        da, db = (hip_sd[k] - p0).flatten().double(), (ref_sd[k].detach() - p0).flatten().double()
        if db.norm().item() < 1e-9:
            assert da.norm().item() < 1e-6, k
            continue
        assert (hip_sd[k] - ref_sd[k].detach()).abs().max().item() <= 2.5 * LR * steps, k       # never further apart than the steps allow
        if k.endswith(".v.bias"):                            # key bias: structurally zero gradient (softmax shift invariance), rounding noise on both sides
            continue
        cos = (da @ db).item() / (da.norm().item() * db.norm().item())
        ratio = da.norm().item() / db.norm().item()
        assert cos >= 0.85 and 0.8 <= ratio <= 1.25, (k, cos, ratio)
        checked += 1
    assert checked >= 40
    with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch_dtype):
        graphs, tweets, attention_masks, _ = _data(1)[0]
        o_hip = model(tweets.to(dev), graphs.to(torch_dtype).to(dev), attention_mask=attention_masks.to(dev))
    o_ref = ref(tweets, graphs.to(torch_dtype).float(), attention_masks)
    assert (o_hip.cpu() - o_ref.detach()).abs().max().item() <= 1e-2


def test_reference_train_loop_in_train_mode_runs_and_learns(dev, dropin_meant):
    """the same sequence with the model in .train() (Dropout(0.5) live, fused into the norm kernels): finite, nothing skipped
    by the scaler, the loss on a fixed batch goes down"""
    torch.manual_seed(0)
    model = dropin_meant.meant(embedding=torch.nn.Embedding(100, 128), flash=False, num_heads=2, num_encoders=1, **ARGS).to(dev).train()
    optimizer = torch.optim.AdamW(model.parameters(), lr=3e-3)
    scaler = torch.amp.GradScaler("cuda")
    loss_fct = torch.nn.CrossEntropyLoss()
    graphs, tweets, masks, target = _data(1, B=8)[0]
    hist = []
    for _ in range(12):
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            out = model.forward(tweets.long().to(dev), graphs.to(torch.float16).to(dev), attention_mask=masks.cuda())
            loss = loss_fct(out, target.to(dev).long())
        optimizer.zero_grad()
        scaler.scale(loss).backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        scaler.step(optimizer)
        scaler.update()
        hist.append(loss.item())
    assert all(np.isfinite(hist)) and scaler.get_scale() == SCALE
    assert min(hist[-3:]) < hist[0] - 0.02, hist


def test_encoder_transplant(dev, dropin_meant):
    """in_loop_train.py:503-504: `model.languageEncoders = language_encoders.languageEncoders` (pretrained stacks moved into
    a fresh model): the transplanted parameters are the ones the kernels read (weight cache keyed on the new parameters),
    state_dict follows, and a round trip through state_dict reproduces the output bit for bit"""
    mk = lambda seed: (torch.manual_seed(seed), dropin_meant.meant(embedding=torch.nn.Embedding(100, 128), flash=False, num_heads=2,
                                                                  num_encoders=2, **ARGS).to(dev).eval())[1]
    model, donor = mk(1), mk(2)
    rs = np.random.RandomState(5)
    ids = torch.from_numpy(rs.randint(0, 100, (2, 3, 16))).to(dev)
    img = torch.from_numpy(rs.standard_normal((2, 3, 4, 32, 32)).astype("float32")).to(dev)
    mask = torch.ones(2, 3, 16, device=dev)
    for dtype in (torch.float32, torch.bfloat16):
        model.compute_dtype = donor.compute_dtype = dtype
        before = model(ids, img, mask)                      # fills the weight cache with the OLD parameters' copies
    model.languageEncoders = donor.languageEncoders
    model.visionEncoders = donor.visionEncoders
    sd = model.state_dict()
    for k, v in donor.state_dict().items():
        if k.startswith(("languageEncoders.", "visionEncoders.")):
            assert sd[k].data_ptr() == v.data_ptr(), k
    fresh = mk(3)
    fresh.load_state_dict(sd)
    for dtype in (torch.float32, torch.bfloat16):
        model.compute_dtype = fresh.compute_dtype = dtype
        after = model(ids, img, mask)
        assert torch.equal(after, fresh(ids, img, mask))
        assert not torch.equal(after, before)
