"""GPU parity of the train-step tail (row a18): CE-on-probabilities, global-norm clip + AdamW, end-to-end steps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_ce_on_probs_matches_torch(dev):
    from meant_amd.train import cross_entropy_on_probs
    g = torch.Generator().manual_seed(0)
    for B, C in [(5, 2), (128, 2), (300, 7), (4, 3129)]:
        probs = torch.rand(B, C, generator=g)
        tgt = torch.randint(0, C, (B,), generator=g)
        pr = probs.clone().requires_grad_()
        ref = torch.nn.functional.cross_entropy(pr, tgt)
        ref.backward()
        ph = probs.to(dev).requires_grad_()
        out = cross_entropy_on_probs(ph, tgt.to(dev))
        out.backward()
        assert abs(out.item() - ref.item()) < 1e-5
        assert (ph.grad.cpu() - pr.grad).abs().max().item() < 1e-6


def test_ce_on_probs_out_of_range_target_is_flagged(dev):
    """torch raises for a class index outside [0, C); the kernel poisons the loss and that row's gradient with NaN instead of
    reading out of bounds (in_loop_train.py:232 feeds 2-class heads; a 3-class label is the typical mistake)"""
    from meant_amd.train import cross_entropy_on_probs
    probs = torch.rand(6, 2, device=dev, requires_grad=True)
    for bad in (2, -100, 10 ** 9):
        tgt = torch.tensor([0, 1, bad, 1, 0, 1], device=dev)
        out = cross_entropy_on_probs(probs, tgt)
        assert torch.isnan(out)
        probs.grad = None
        out.backward()
        assert torch.isnan(probs.grad[2]).all() and torch.isfinite(probs.grad[[0, 1, 3, 4, 5]]).all()


def test_optimizer_and_scheduler_resume_continue_identically(dev):
    """checkpoint_train.py:217-219,333-336 / in_loop_train.py:547-567 of the reference save and restore `model.state_dict()`,
    `optimizer.state_dict()` and the scheduler: a FRESH model + TrainStep + CosineWarmRestarts loaded from the three state dicts
    after step 3 (epoch 8, i.e. one epoch past the warm restart at 7) must take the next steps exactly as the original run does.
    Train mode (the dropout masks are drawn from the torch seed set before every step), option `deterministic` (every reduction in
    one order, including the clip's global-norm sum: with float atomics there the clip coefficient differs in its last bit from run
    to run, a few fp32 weights then round to a different bf16 value in the next step, and AdamW turns the changed rounding noise of
    mathematically-zero gradients -- the key biases of every attention -- into fractions of an lr step: 1e-5 after two steps).
    So "exactly" means bit for bit.  Control: the same resume WITHOUT the optimizer's state lands somewhere else."""
    import meant_amd
    from meant_amd import _lib
    from meant_amd.train import TrainStep, CosineWarmRestarts
    old = _lib.get_option("deterministic")
    _lib.set_option("deterministic", 1)
    try:
        def make():
            torch.manual_seed(0)
            m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2).to(dev).train()
            m.compute_dtype = torch.bfloat16
            ts = TrainStep(m, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
            return m, ts, CosineWarmRestarts(ts.opt, T_0=7, eta_min=1e-5)
        g = torch.Generator().manual_seed(5)
        ids = torch.randint(0, 100, (8, 3, 16), generator=g).to(dev)
        img = torch.randn(8, 3, 4, 32, 32, generator=g).to(dev)
        mask = torch.ones(8, 3, 16, device=dev)
        tgt = torch.tensor([0, 1, 0, 1, 1, 0, 1, 0], device=dev)

        def steps(ts, sched, first, n):
            for i in range(first, first + n):
                torch.manual_seed(100 + i)
                ts(ids, img, mask, target=tgt)
                sched.step()
        m, ts, sched = make()
        for _ in range(5):                                   # epochs 1..5 without training: the restart at 7 is then inside the test
            sched.step()
        steps(ts, sched, 0, 3)                               # epochs 6, 7 (lr back at its base value), 8
        assert sched.epoch == 8 and abs(sched.lr_at(7) - 1e-3) < 1e-12
        saved = {"model": {k: v.clone() for k, v in m.state_dict().items()}, "opt": ts.opt.state_dict(), "sched": sched.state_dict()}
        steps(ts, sched, 3, 2)
        want = {k: v.clone() for k, v in m.state_dict().items()}
        m2, ts2, sched2 = make()
        with torch.no_grad():
            for p in m2.parameters():                        # a different starting point: everything must come from the state dicts
                p.add_(0.123)
        m2.load_state_dict(saved["model"])
        ts2.opt.load_state_dict(saved["opt"])
        sched2.load_state_dict(saved["sched"])
        assert ts2.opt.step_count == 3 and ts2.opt.lr == sched2.lr_at(8)
        steps(ts2, sched2, 3, 2)
        worst = max((v.float() - want[k].float()).abs().max().item() for k, v in m2.state_dict().items())
        assert worst == 0.0, worst
        m3, ts3, sched3 = make()                             # control: model and scheduler restored, optimizer moments lost
        m3.load_state_dict(saved["model"])
        sched3.load_state_dict(saved["sched"])
        steps(ts3, sched3, 3, 2)
        lost = max((v.float() - want[k].float()).abs().max().item() for k, v in m3.state_dict().items())
        assert lost > 1e-4, lost
    finally:
        _lib.set_option("deterministic", old)


@pytest.mark.parametrize("max_norm", [None, 1.0, 1e-3])
def test_fused_adamw_matches_torch(dev, max_norm):
    """3 steps of clip_grad_norm_ + torch.optim.AdamW on the CPU vs the fused flat-bucket kernels"""
    from meant_amd.parallel import GradReducer
    from meant_amd.train import FusedAdamW
    g = torch.Generator().manual_seed(1)
    shapes = [(768, 768), (768,), (3, 5, 7), (1,), (2304, 768)]
    ref_params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    hip_params = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_params]
    ref_opt = torch.optim.AdamW(ref_params, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    red = GradReducer(hip_params, bucket_mb=1.0)          # several buckets
    assert red.num_buckets >= 2
    opt = FusedAdamW(red, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, max_grad_norm=max_norm)
    for step in range(3):
        grads = [torch.randn(*s, generator=g) * (10.0 if step == 1 else 0.1) for s in shapes]
        for p, gr in zip(ref_params, grads):
            p.grad = gr.clone()
        if max_norm is not None:
            torch.nn.utils.clip_grad_norm_(ref_params, max_norm)
        ref_opt.step()
        red.prepare()
        for p, gr in zip(hip_params, grads):
            p.grad.copy_(gr.to(dev))
        red.wait()
        opt.step()
        for k, (a, b) in enumerate(zip(hip_params, ref_params)):
            assert (a.detach().cpu() - b.detach()).abs().max().item() < 2e-6, (step, k)
    gn = opt.grad_norm().item()
    assert abs(gn - torch.sqrt(sum((gr ** 2).sum() for gr in grads)).item()) < 1e-3 * gn


def test_train_step_learns_and_refreshes_weight_copies(dev):
    """a few full steps on a tiny MEANT: the loss goes down, parameters move, and the cached bf16 weight copies
    follow the raw-pointer updates of the fused optimizer"""
    import meant_amd
    from meant_amd.train import TrainStep, CosineWarmRestarts
    torch.manual_seed(0)
    m = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    m.compute_dtype = torch.bfloat16
    ids = torch.randint(0, 100, (8, 3, 16), device=dev)
    img = torch.randn(8, 3, 4, 32, 32, device=dev)
    mask = torch.ones(8, 3, 16, device=dev)
    tgt = torch.tensor([0, 1, 0, 1, 1, 0, 1, 0], device=dev)
    ts = TrainStep(m, lr=2e-3, weight_decay=0.0, max_grad_norm=1.0)
    sched = CosineWarmRestarts(ts.opt, T_0=7)
    w0 = m.mlpHead[1].weight.detach().clone()
    losses = []
    for _ in range(30):
        loss, out = ts(ids, img, mask, target=tgt)
        losses.append(loss.item())
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0] - 0.02, losses
    assert (m.mlpHead[1].weight.detach() - w0).abs().max().item() > 0
    sched.step()
    assert abs(ts.opt.lr - 2e-3 * (1 + np.cos(np.pi / 7)) / 2) < 1e-9
    # state_dict still has the reference's keys and loads into a fresh model that reproduces the output
    m2 = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).to(dev).eval()
    m2.compute_dtype = torch.bfloat16
    m2.load_state_dict(m.state_dict())
    assert torch.equal(m2(ids, img, mask), m(ids, img, mask))


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MEANT_REPO"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import meant_amd
from meant_amd.parallel import GradReducer
from meant_amd.train import cross_entropy_on_probs
torch.manual_seed(0)
model = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2).cuda().eval()
model.compute_dtype = torch.bfloat16
tw = torch.randint(0, 100, (4, 3, 16), device="cuda"); im = torch.randn(4, 3, 4, 32, 32, device="cuda")
mask = torch.ones(4, 3, 16, device="cuda"); tgt = torch.randint(0, 2, (4,), device="cuda")
def grads(always):
    os.environ["MEANT_REDUCE_ALWAYS"] = "1" if always else "0"
    for p in model.parameters():
        p.grad = None
    red = GradReducer(model.parameters(), bucket_mb=0.05)
    assert red.active == always and red.fused_avg
    red.prepare()
    cross_entropy_on_probs(model(tw, im, mask), tgt).backward()
    if always:
        assert any(b.handle is not None for b in red.buckets), "no collective was launched from the hooks"
    red.wait()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters() if p.grad is not None], red.num_buckets
g1, nb = grads(True)
g0, _ = grads(False)
assert nb > 1
for a, b in zip(g1, g0):
    # (not bit-equal: the dW kernels accumulate with floating-point atomics, two backward passes differ in the last bits)
    assert (a - b).abs().max().item() <= 1e-3 * max(b.abs().max().item(), 1e-6), "a one-rank average must be the identity"
x = torch.arange(8, device="cuda", dtype=torch.float32)
dist.all_reduce(x, op=dist.ReduceOp.AVG); assert torch.equal(x.cpu(), torch.arange(8, dtype=torch.float32))
# the embedding table's gradient delivered and reduced in row slices, over the real RCCL call sequence (one collective per slice from
# the launch stream, ncclAvg, joined in wait())
from meant_amd import ops
os.environ["MEANT_REDUCE_ALWAYS"] = "1"
emb, lin = torch.nn.Embedding(6001, 128).cuda(), torch.nn.Linear(128, 128).cuda()
ids = torch.randint(0, 6001, (8192,), device="cuda")
def sliced(slices):
    for p in list(emb.parameters()) + list(lin.parameters()):
        p.grad = None
    red = GradReducer(list(emb.parameters()) + list(lin.parameters()), bucket_mb=1.0, direct_grads=True, row_slices=slices)
    red.row_slice_min_bytes = 0
    red.prepare()
    ops.linear(ops.embedding(ids, emb.weight, torch.bfloat16), lin.weight, lin.bias).float().sum().backward()
    n = len(red._owner[id(emb.weight)].slice_handles)
    red.wait()
    torch.cuda.synchronize()
    g = emb.weight.grad.clone()
    red.close()
    return g, n
ga, na = sliced(4)
gb, nb1 = sliced(1)
assert na == 4 and nb1 == 0, (na, nb1)
assert (ga - gb).abs().max().item() <= 1e-4 * gb.abs().max().item()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", nb)
"""


def test_grad_reducer_over_rccl_one_rank(dev):
    """the real RCCL call sequence (async all-reduce with ncclAvg from autograd hooks, two-stream ordering, wait) on a
    one-rank group: the only part of the multi-GPU path a one-GPU box can run for real; in a child process so that the
    process group does not leak into the other tests"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MEANT_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_TWO_RANKS_ON_ONE_GPU = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MEANT_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share GPU 0: RCCL needs one device per rank
torch.cuda.set_device(0)
import numpy as np
import meant_amd
from meant_amd.parallel import GradReducer, shard_batch
from meant_amd.train import cross_entropy_on_probs
torch.manual_seed(0)
model = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2).cuda().eval()
model.compute_dtype = torch.bfloat16
d = torch.load(os.environ["MEANT_BATCH"])
lo, hi = shard_batch(d["ids"].shape[0], rank, world)
red = GradReducer(model.parameters(), bucket_mb=0.25, direct_grads=True)
assert red.active and red.num_buckets > 1 and not red.fused_avg
for _ in range(2):                                                # two steps: prepare() re-arms the hooks and the sinks
    red.prepare()
    out = model(d["ids"][lo:hi].cuda(), d["img"][lo:hi].cuda(), d["mask"][lo:hi].cuda())
    cross_entropy_on_probs(out, d["tgt"][lo:hi].cuda()).backward()
    assert any(b.handle is not None for b in red.buckets), "no collective was launched during backward"
    red.wait()
torch.cuda.synchronize()
if rank == 0:
    torch.save({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}, os.environ["MEANT_OUT"])
dist.barrier()
dist.destroy_process_group()
print("TWO_RANKS_OK", rank)
"""


def test_two_ranks_share_one_gpu_reduced_grads_equal_single_process(dev, tmp_path):
    """bench.py's N > 1 path rehearsed on one GPU: two processes, one rank each, both on cuda:0 over gloo, the HIP modules,
    the batch sharded by shard_batch, gradients written straight into the reducer's buckets (direct_grads) and all-reduced
    from the autograd hooks while backward runs.  The averaged gradients must equal those of ONE process on the whole
    batch (mean loss over equal shards == mean over the batch), which is what the 8-GPU launch relies on."""
    import os
    import subprocess
    import sys
    import meant_amd
    from meant_amd.train import cross_entropy_on_probs
    rs = np.random.RandomState(17)
    batch = {"ids": torch.from_numpy(rs.randint(0, 100, (4, 3, 16))), "img": torch.from_numpy(rs.standard_normal((4, 3, 4, 32, 32)).astype("float32")),
             "mask": torch.ones(4, 3, 16), "tgt": torch.tensor([0, 1, 1, 0])}
    batch["mask"][1, :, 9:] = 0
    bpath, opath = str(tmp_path / "batch.pt"), str(tmp_path / "grads.pt")
    torch.save(batch, bpath)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0",
                   MEANT_REPO=root, MEANT_BATCH=bpath, MEANT_OUT=opath, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANKS_ON_ONE_GPU], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "TWO_RANKS_OK" in so, so[-2000:] + se[-4000:]
    got = torch.load(opath)
    torch.manual_seed(0)
    model = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2).to(dev).eval()
    model.compute_dtype = torch.bfloat16
    out = model(batch["ids"].to(dev), batch["img"].to(dev), batch["mask"].to(dev))
    cross_entropy_on_probs(out, batch["tgt"].to(dev)).backward()
    assert set(got) == {k for k, p in model.named_parameters() if p.grad is not None}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        ref = p.grad.detach().float().cpu()
        # bf16 tier: the two half-batches round differently from the whole batch only in the last bits of the fp32 sums
        assert (got[k] - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1e-4), k


# ---- bench.py as the driver launches it ----------------------------------------------------------------------------------
def _run_bench(argv, env_extra, nproc=1, port="29561", timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(root, "bench.py"), "--gpus", str(nproc)] + argv
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + argv
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_prints_one_contract_line(dev):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` exactly as the driver launches it, both
    ranks on GPU 0 over gloo (a one-GPU box cannot give RCCL two devices): one JSON line from rank 0 with the contract's
    fields, the whole-job batch, the gradient collective live"""
    res = _run_bench(["--batch-per-gpu", "2", "--steps", "2", "--warmup", "1"],
                     {"MEANT_DIST_BACKEND": "gloo", "MEANT_ALL_RANKS_ON_GPU0": "1"}, nproc=2)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["grad_allreduce"] is True and res["config"]["global_batch"] == 4 and res["config"]["parallelism"] == "dp2"
    assert res["unit"] == "samples/s" and res["scaling"] == "weak" and res["higher_is_better"] is True and res["dtype"] == "bf16"
    assert np.isfinite(res["value"]) and res["value"] > 0
    assert abs(res["value"] - 4 / res["ms_per_step"] * 1e3) <= 0.01 * res["value"]
    assert "roofline" in res and "cpu_baseline" not in res          # the CPU leg runs at N = 1 only


@pytest.mark.parametrize("model,batch", [("meant_vqa", 4), ("meant_vision", 8)])
def test_bench_other_configs_print_the_contract_line(dev, model, batch):
    """BASELINE.json configs[4] (meant_vqa) and configs[1] (meant_vision) through bench.py --model: same contract"""
    res = _run_bench(["--model", model, "--batch-per-gpu", str(batch), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert res["n_gpus"] == 1 and model in res["metric"] and model in res["config"]["workload"]
    assert res["config"]["batch_per_gpu"] == batch and np.isfinite(res["value"]) and res["value"] > 0
    assert res["ms_per_step_median"] > 0 and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(res["roofline"])


def test_bench_fp32_tier_field_and_one_rank_rccl_line(dev):
    """`--fp32-batch n` adds the fp32 tier's own throughput to the line (the tier north_star's 1e-3 tolerance is stated
    for); `MEANT_REDUCE_ALWAYS=1` runs the step with a one-rank RCCL group -- the reducer's launch stream and RCCL's stream beside
    the step's own, the rehearsal of a multi-GPU launch one GPU allows -- and must come out as the same contract line"""
    res = _run_bench(["--batch-per-gpu", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--fp32-batch", "1"], {})
    f = res["fp32_tier"]
    assert f["batch_per_gpu"] == 1 and f["ms_per_step"] > 0 and np.isfinite(f["samples_per_s"]) and f["samples_per_s"] > 0
    assert 0 < f["whole_step_f32_matrix_frac"] < 1
    red = _run_bench(["--batch-per-gpu", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {"MEANT_REDUCE_ALWAYS": "1"})
    assert red["n_gpus"] == 1 and red["config"]["grad_allreduce"] is True and np.isfinite(red["value"]) and red["value"] > 0
    assert res["config"]["grad_allreduce"] is False


# ---- gradient sinks and shared parameters ------------------------------------------------------------------------
_TIED_TWO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MEANT_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from meant_amd import ops
from meant_amd.parallel import GradReducer
V, d = int(os.environ["MEANT_V"]), 128
torch.manual_seed(0)
emb = torch.nn.Embedding(V, d).cuda()
lin = torch.nn.Linear(d, d).cuda()
g = torch.Generator().manual_seed(5)
ids = torch.randint(0, V, (2 * 4096,), generator=g).view(2, 4096)[rank].cuda()
tgt = torch.randint(0, V, (2 * 4096,), generator=g).view(2, 4096)[rank].cuda()
params = list(emb.parameters()) + list(lin.parameters())
red = GradReducer(params, bucket_mb=64.0, direct_grads=True)
for _ in range(2):
    red.prepare()
    x = ops.embedding(ids, emb.weight, torch.bfloat16)                     # the table read by the gather ...
    h = ops.linear(x, lin.weight, lin.bias)
    loss = ops.vocab_linear_cross_entropy(h, emb.weight, None, tgt)        # ... and, tied, by the vocabulary decoder
    loss.backward()
    red.wait()
torch.cuda.synchronize()
if rank == 0:
    torch.save({"emb": emb.weight.grad.detach().cpu().clone(), "w": lin.weight.grad.detach().cpu().clone()}, os.environ["MEANT_OUT"])
dist.barrier()
dist.destroy_process_group()
print("TIED_OK", rank)
"""


@pytest.mark.parametrize("V", [1000, 1024])
def test_tied_embedding_and_vocab_decoder_reduce_both_contributions(dev, tmp_path, V):
    """a word embedding tied to the vocabulary decoder (pretrain_mlm.py:318-319) under GradReducer(direct_grads=True) on two
    ranks: the table's gradient has an early contribution (decoder) and a late one (gather); neither may start the
    bucket's all-reduce alone.  V = 1024 is the unpadded case in which the decoder could use a sink as well.  The reduced
    gradient must equal one process on both shards."""
    import os
    import subprocess
    import sys
    from meant_amd import ops
    opath = str(tmp_path / "tied.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0",
                   MEANT_REPO=root, MEANT_OUT=opath, MEANT_V=str(V), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TIED_TWO_RANKS], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "TIED_OK" in so, so[-2000:] + se[-4000:]
    got = torch.load(opath)
    d = 128
    torch.manual_seed(0)
    emb = torch.nn.Embedding(V, d).to(dev)
    lin = torch.nn.Linear(d, d).to(dev)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (2 * 4096,), generator=g).to(dev)
    tgt = torch.randint(0, V, (2 * 4096,), generator=g).to(dev)
    x = ops.embedding(ids, emb.weight, torch.bfloat16)
    loss = ops.vocab_linear_cross_entropy(ops.linear(x, lin.weight, lin.bias), emb.weight, None, tgt)
    loss.backward()                                                        # mean over both shards == average of the shard means
    for k, ref in (("emb", emb.weight.grad), ("w", lin.weight.grad)):
        ref = ref.float().cpu()
        assert (got[k] - ref).abs().max().item() <= 5e-3 * ref.abs().max().item(), k


def test_vocab_linear_without_padding_under_direct_grads(dev):
    """V a multiple of 256, bias None, the weight registered with GradReducer(direct_grads=True): dW goes straight into the
    bucket (the backward used to index a None); with a bias both go there; gradients equal the autograd path's"""
    from meant_amd import ops
    from meant_amd.parallel import GradReducer
    torch.manual_seed(2)
    V, d, T = 512, 128, 1024
    x = torch.randn(T, d, device=dev).bfloat16()
    tgt = torch.randint(0, V, (T,), device=dev)
    for with_bias in (False, True):
        w = torch.nn.Parameter(torch.randn(V, d, device=dev) * 0.05)
        b = torch.nn.Parameter(torch.randn(V, device=dev) * 0.1) if with_bias else None
        xr = x.clone().requires_grad_()
        ops.vocab_linear_cross_entropy(xr, w, b, tgt).backward()
        ref_w, ref_b, ref_x = w.grad.clone(), (b.grad.clone() if with_bias else None), xr.grad.clone()
        w.grad = None
        if with_bias:
            b.grad = None
        red = GradReducer([w] + ([b] if with_bias else []), direct_grads=True)
        for _ in range(2):
            red.prepare()
            xs = x.clone().requires_grad_()
            ops.vocab_linear_cross_entropy(xs, w, b, tgt).backward()
            red.wait()
            assert (w.grad - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item() + 1e-7
            assert torch.equal(xs.grad, ref_x)
            if with_bias:
                assert (b.grad - ref_b).abs().max().item() <= 1e-5 * ref_b.abs().max().item() + 1e-7
        red.close()
        assert id(w) not in ops.grad_sinks


def test_grad_reducer_no_sync_accumulates_and_second_reducer_closes_the_first(dev):
    """gradient accumulation over micro-batches with direct_grads: backward passes under no_sync() add into the buckets
    without counting; the reducer a second one replaces stops receiving reports"""
    from meant_amd import ops
    from meant_amd.parallel import GradReducer
    torch.manual_seed(4)
    lin = torch.nn.Linear(128, 128).to(dev)
    xs = [torch.randn(256, 128, device=dev).bfloat16() for _ in range(3)]
    ref = None
    for x in xs:
        ops.linear(x, lin.weight, lin.bias).float().sum().backward()
    ref_w, ref_b = lin.weight.grad.clone(), lin.bias.grad.clone()
    red = GradReducer(lin.parameters(), direct_grads=True)
    red.prepare()
    with red.no_sync():
        for x in xs[:2]:
            ops.linear(x, lin.weight, lin.bias).float().sum().backward()
    ops.linear(xs[2], lin.weight, lin.bias).float().sum().backward()
    red.wait()
    assert (lin.weight.grad - ref_w).abs().max().item() <= 1e-4 * ref_w.abs().max().item()
    assert (lin.bias.grad - ref_b).abs().max().item() <= 1e-4 * ref_b.abs().max().item()
    with pytest.raises(RuntimeError, match="twice"):                       # outside no_sync a second backward is refused, as before
        ops.linear(xs[0], lin.weight, lin.bias).float().sum().backward()
    red2 = GradReducer(lin.parameters(), direct_grads=True)
    assert red._closed and ops.grad_sinks[id(lin.weight)].reducer() is red2
    red2.prepare()
    ops.linear(xs[0], lin.weight, lin.bias).float().sum().backward()
    red2.wait()
    widx = [i for i, q in enumerate(red2.buckets[0].params) if q is lin.weight][0]
    assert lin.weight.grad.data_ptr() == red2.buckets[0].flat.data_ptr() + red2.buckets[0].offsets[widx] * 4
    del red2
    import gc
    gc.collect()
    assert id(lin.weight) not in ops.grad_sinks                            # the table does not keep a dead reducer's entries


def test_train_step_micro_batches_give_the_same_update(dev):
    """TrainStep(micro_batches=2): two half-batch forward/backward passes accumulated in the reducer's buckets (no_sync) move the
    parameters exactly as one pass over the whole batch does (eval mode: no dropout masks to differ)"""
    import meant_amd
    from meant_amd.train import TrainStep
    rs = np.random.RandomState(23)
    ids = torch.from_numpy(rs.randint(0, 100, (4, 3, 16))).to(dev)
    img = torch.from_numpy(rs.standard_normal((4, 3, 4, 32, 32)).astype("float32")).to(dev)
    mask = torch.ones(4, 3, 16, device=dev)
    tgt = torch.tensor([0, 1, 1, 0], device=dev)
    after = []
    for k in (1, 2):
        torch.manual_seed(6)
        model = meant_amd.meant(128, 128, 4, 32, 32, 16, 3, 2, torch.nn.Embedding(100, 128), num_heads=2, num_encoders=2).to(dev).eval()
        model.compute_dtype = torch.bfloat16
        ts = TrainStep(model, lr=1e-2, max_grad_norm=None, micro_batches=k)
        loss, out = ts(ids, img, mask, target=tgt)
        assert out.shape == (4, 2) and torch.isfinite(loss)
        torch.cuda.synchronize()
        after.append(({n: p.detach().clone() for n, p in model.named_parameters()}, loss.item(),
                      {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        ts.reducer.close()
    assert abs(after[0][1] - after[1][1]) < 2e-3
    big = max(g.norm().item() for g in after[0][2].values())
    for n, g in after[0][2].items():                       # the accumulated gradients themselves
        assert (g - after[1][2][n]).norm().item() <= 2e-2 * max(g.norm().item(), 2e-2 * big), n


_EMB_SLICES_TWO_RANKS = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MEANT_REPO"])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from meant_amd import ops
from meant_amd.parallel import GradReducer
V, d = 6001, 128
torch.manual_seed(0)
emb = torch.nn.Embedding(V, d).cuda()
lin = torch.nn.Linear(d, d).cuda()
g = torch.Generator().manual_seed(7)
ids = torch.randint(0, V, (2, 8192), generator=g)[rank].cuda()
wts = torch.randn(2, 8192, d, generator=g)[rank].cuda()
red = GradReducer(list(emb.parameters()) + list(lin.parameters()), bucket_mb=1.0, direct_grads=True, row_slices=int(os.environ["MEANT_SLICES"]))
red.row_slice_min_bytes = 0
assert len(red._owner[id(emb.weight)].params) == 1
for _ in range(2):
    red.prepare()
    x = ops.embedding(ids, emb.weight, torch.bfloat16)
    (ops.linear(x, lin.weight, lin.bias).float() * wts).sum().backward()
    nh = len(red._owner[id(emb.weight)].slice_handles)
    red.wait()
assert nh == (int(os.environ["MEANT_SLICES"]) if int(os.environ["MEANT_SLICES"]) > 1 else 0), nh
torch.cuda.synchronize()
if rank == 0:
    torch.save({"emb": emb.weight.grad.detach().cpu().clone(), "w": lin.weight.grad.detach().cpu().clone()}, os.environ["MEANT_OUT"])
dist.barrier()
dist.destroy_process_group()
print("SLICES_OK", rank)
"""


@pytest.mark.parametrize("slices", [1, 4])
def test_embedding_gradient_reduced_in_row_slices(dev, tmp_path, slices):
    """the embedding table's gradient produced and all-reduced in row slices (meant_embedding_bwd_sorted_range + one collective per
    slice, each started while the next slice is being summed): two ranks on one GPU over gloo; the averaged gradient equals one
    process on both shards, for 4 slices as for the single-collective path"""
    import os
    import subprocess
    import sys
    from meant_amd import ops
    opath = str(tmp_path / "slices.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0",
                   MEANT_REPO=root, MEANT_OUT=opath, MEANT_SLICES=str(slices), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _EMB_SLICES_TWO_RANKS], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "SLICES_OK" in so, so[-2000:] + se[-4000:]
    got = torch.load(opath)
    V, d = 6001, 128
    torch.manual_seed(0)
    emb = torch.nn.Embedding(V, d).to(dev)
    lin = torch.nn.Linear(d, d).to(dev)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, V, (2, 8192), generator=g).to(dev)
    wts = torch.randn(2, 8192, d, generator=g).to(dev)
    x = ops.embedding(ids, emb.weight, torch.bfloat16)
    ((ops.linear(x, lin.weight, lin.bias).float() * wts).sum() * 0.5).backward()        # the average over the two ranks' sums
    for k, ref in (("emb", emb.weight.grad), ("w", lin.weight.grad)):
        ref = ref.float().cpu()
        assert (got[k] - ref).abs().max().item() <= 5e-3 * ref.abs().max().item(), k
