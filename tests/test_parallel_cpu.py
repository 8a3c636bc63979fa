"""world_size-2 gloo test of the data-parallel gradient path (meant_amd/parallel.py): sharding the batch
over ranks + bucketed all-reduce must reproduce the single-process gradients of the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_model():
    # the reducer is model-agnostic; on CPU we drive it with the CPU oracle's own modules (test infrastructure)
    from oracle import meant_oracle as O
    m = O.meant_tweet(128, 4, 2, 2, torch.nn.Embedding(50, 128), num_heads=2, num_encoders=1)
    O.fill_weights_(m, 1234)
    return m.eval()


def _data():
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 50, (4, 2, 16), generator=g)
    mask = torch.ones(4, 2, 16)
    mask[1, :, 10:] = 0
    tgt = torch.tensor([0, 1, 1, 0])
    return ids, mask, tgt


def _worker(rank, world, port, bucket_mb, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from meant_amd.parallel import GradReducer, shard_batch
    m = _make_model()
    red = GradReducer(m.parameters(), bucket_mb=bucket_mb)
    ids, mask, tgt = _data()
    lo, hi = shard_batch(4, rank, world)
    for _ in range(2):                      # two steps: prepare() must re-arm and re-zero correctly
        red.prepare()
        loss = torch.nn.functional.cross_entropy(m(ids[lo:hi], mask[lo:hi]), tgt[lo:hi])
        loss.backward()
        red.wait()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}, out)
        torch.save(red.num_buckets, out + ".nb")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb", [64.0, 0.05])
def test_two_rank_allreduce_matches_single_process(tmp_path, bucket_mb):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), bucket_mb, out), nprocs=2, join=True)
    got = torch.load(out)
    nb = torch.load(out + ".nb")
    assert nb >= (1 if bucket_mb > 1 else 3)
    m = _make_model()
    ids, mask, tgt = _data()
    # mean over ranks of per-shard mean losses == mean loss over the whole batch (equal shard sizes)
    torch.nn.functional.cross_entropy(m(ids, mask), tgt).backward()
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert torch.allclose(got[k], p.grad, atol=1e-6, rtol=1e-4), k


def test_shard_batch():
    from meant_amd.parallel import shard_batch
    assert shard_batch(1024, 3, 8) == (384, 512)
    with pytest.raises(AssertionError):
        shard_batch(10, 0, 4)


def test_device_batch_loader_cpu_matches_slicing():
    """the input pipeline's host logic: batches equal plain slicing of the host arrays in the shared
    permutation, `None` members stay None, ranks partition every global batch, mean/std are the global ones"""
    import numpy as np
    from meant_amd.data import DeviceBatchLoader, global_mean_std, shard_indices
    rs = np.random.RandomState(0)
    g = rs.standard_normal((20, 2, 4, 8, 8))
    t = rs.randint(0, 100, (20, 2, 16))
    m = np.ones((20, 2, 16), dtype=np.float32)
    y = rs.randint(0, 2, (20,))
    ld = DeviceBatchLoader(g, t, None, m, y, batch_size=4, device="cpu", shuffle=True, seed=3)
    idx = shard_indices(20, 0, 1, 4, True, 3, 0)
    assert len(ld) == 5
    for b, (gg, tt, mm, am, yy) in enumerate(ld):
        sel = idx[b * 4:(b + 1) * 4]
        assert mm is None and gg.dtype == torch.float64
        assert np.array_equal(gg.numpy(), g[sel]) and np.array_equal(tt.numpy(), t[sel]) and np.array_equal(yy.numpy(), y[sel])
    i0, i1 = shard_indices(20, 0, 2, 4, True, 3, 0), shard_indices(20, 1, 2, 4, True, 3, 0)
    assert len(set(i0) & set(i1)) == 0 and len(i0) == len(i1) == 8
    mean, std = global_mean_std(g)
    assert abs(mean - g.mean()) < 1e-12 and abs(std - g.std()) < 1e-12


def _worker_accumulate(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from meant_amd.parallel import GradReducer, shard_batch
    m = _make_model()
    red = GradReducer(m.parameters(), bucket_mb=0.05)
    ids, mask, tgt = _data()
    lo, hi = shard_batch(4, rank, world)
    mid = (lo + hi) // 2
    for _ in range(2):
        red.prepare()
        with red.no_sync():                 # first micro-batch: accumulates locally, counts nothing, starts no collective
            (torch.nn.functional.cross_entropy(m(ids[lo:mid], mask[lo:mid]), tgt[lo:mid]) * 0.5).backward()
            assert all(b.handle is None for b in red.buckets)
        (torch.nn.functional.cross_entropy(m(ids[mid:hi], mask[mid:hi]), tgt[mid:hi]) * 0.5).backward()
        assert any(b.handle is not None for b in red.buckets)
        red.wait()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}, out)
    red.close()
    n_before = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}
    torch.nn.functional.cross_entropy(m(ids[lo:hi], mask[lo:hi]), tgt[lo:hi]).backward()      # hooks are gone: plain accumulation, no collective
    assert all(b.handle is None or b.handle.is_completed() for b in red.buckets)
    assert any(not torch.equal(n_before[k], p.grad) for k, p in m.named_parameters() if p.requires_grad)
    dist.barrier()
    dist.destroy_process_group()


def test_no_sync_micro_batches_reduce_once_and_close_detaches(tmp_path):
    """GradReducer.no_sync(): two micro-batches per rank accumulate in the buckets and are reduced by ONE round of collectives,
    giving the gradients of the whole batch; close() removes the hooks"""
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker_accumulate, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _make_model()
    ids, mask, tgt = _data()
    torch.nn.functional.cross_entropy(m(ids, mask), tgt).backward()
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert torch.allclose(got[k], p.grad, atol=1e-6, rtol=1e-4), k
