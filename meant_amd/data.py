"""Input pipeline for the MEANT path (reference: in_loop_train.py:66-78, 579-639).

What the reference does: the whole data set lives in host numpy arrays -- price-chart "graphs" as FLOAT64
(in_loop_train.py:48,589: 12 x 4 x 224 x 224 = 2.4 M values = 19.3 MB per sample), token ids, {0,1} attention masks, labels --
optionally normalised in place by the global mean / std (:591-593); a `DataLoader(pin_memory=True)` slices
batches and the train loop calls `.to(device)` on every tensor, synchronously, before each step (:202-212).

What this does instead (one process per GPU, the GPU is the consumer):
  * batches are cut from the host arrays straight into two sets of PINNED staging buffers and sent with
    non-blocking copies on a dedicated HIP stream, so the H2D transfer of batch i+1 overlaps the step on batch i;
  * pixels travel in their STORAGE type (float64 / float32 / uint8 -- nothing is converted or normalised on the
    host) and are turned into normalised bf16 patches by one kernel on the device (`meant_patchify_raw`, reached
    through the model's own patch embedding: `model.patchEmbed[0].set_normalization(mean, std)`);
  * arrays that live in RAM can be page-locked in place (`pin_source_bytes`): their rows then travel by DMA straight
    out of the data set, one copy per sample, and the host touches no pixel at all;
  * otherwise the gather into pinned memory runs on a small thread pool (torch releases the GIL inside index_select) and one
    batch AHEAD of the consumer on a producer thread, so it overlaps both the GPU step and the Python that launches it;
  * the global mean / std are one streaming pass over the array (`global_mean_std`), not two in-place passes that
    rewrite 19.3 MB per sample.
The loader is an iterator of `(graphs, tweets, macds, attention_masks, labels)` like the reference's
`customDataset` (in_loop_train.py:62-76), so the reference's loop body runs on it unchanged.  On a CPU device it
degrades to plain slicing (used by the CPU tests).
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, Optional, Sequence, Tuple

import numpy as np
import torch


def global_mean_std(a: np.ndarray, chunk: int = 64) -> Tuple[float, float]:
    """mean and (population) std of a whole array in one chunked pass, float64 accumulation -- what
    `graphs -= np.mean(graphs); graphs /= np.std(graphs)` (in_loop_train.py:591-593) normalises by."""
    n, s, ss = 0, 0.0, 0.0
    flat = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(1, -1)
    for i in range(0, flat.shape[0], chunk):
        blk = np.asarray(flat[i:i + chunk], dtype=np.float64)
        n += blk.size
        s += float(blk.sum())
        ss += float(np.square(blk).sum())
    mean = s / max(n, 1)
    var = max(ss / max(n, 1) - mean * mean, 0.0)
    return mean, float(np.sqrt(var))


def shard_indices(n: int, rank: int, world: int, batch_size: int, shuffle: bool, seed: int, epoch: int) -> np.ndarray:
    """sample indices of this rank for one epoch: a permutation shared by all ranks (same seed + epoch), truncated to
    whole global batches, rank r taking rows [r*batch, (r+1)*batch) of every global batch."""
    order = np.random.RandomState(seed + epoch).permutation(n) if shuffle else np.arange(n)
    g = batch_size * world
    nb = n // g
    order = order[:nb * g].reshape(nb, world, batch_size)
    return order[:, rank, :].reshape(-1)


class DeviceBatchLoader:
    """Double-buffered host -> device batch stream.

        loader = DeviceBatchLoader(graphs, tweets, macds, attention_masks, labels, batch_size=128, device="cuda")
        for graphs, tweets, macds, attention_masks, target in loader:       # device tensors, storage dtypes
            out = model(tweets.long(), graphs, attention_masks)

    `arrays` are host numpy arrays (or memmaps) with a common leading dimension; `None` entries stay `None`."""

    def __init__(self, *arrays: Optional[np.ndarray], batch_size: int, device="cuda", shuffle: bool = False, seed: int = 0,
                 rank: int = 0, world: int = 1, drop_last: bool = True, pin_source_bytes: int = 0):
        assert any(a is not None for a in arrays), "no data"
        self.arrays = arrays
        self.n = next(a.shape[0] for a in arrays if a is not None)
        for a in arrays:
            assert a is None or a.shape[0] == self.n, "arrays must share the sample axis"
        self.batch_size, self.shuffle, self.seed, self.rank, self.world = batch_size, shuffle, seed, rank, world
        self.device = torch.device(device)
        self.epoch = 0
        self.on_gpu = self.device.type == "cuda"
        assert drop_last, "partial batches are dropped (the kernels are tuned for the fixed batch shape)"
        self._stage = None
        self._stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # Arrays that live in RAM (not memmaps) and are at least `pin_source_bytes` big can be page-locked IN PLACE
        # (hipHostRegister): their rows then go to the device by DMA straight from the data set, one copy per sample,
        # and the host never touches the pixels at all.  0 disables (default): registering is a one-time cost
        # proportional to the array and needs the memory to stay resident.
        self._registered = [False] * len(arrays)
        if self.on_gpu and pin_source_bytes > 0:
            rt = torch.cuda.cudart()
            for i, a in enumerate(arrays):
                if a is None or isinstance(a, np.memmap) or not a.flags["C_CONTIGUOUS"] or a.nbytes < pin_source_bytes:
                    continue
                try:
                    rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
                    self._registered[i] = (int(rc) == 0)
                except Exception:
                    self._registered[i] = False
        self._workers = max(1, min(int(os.environ.get("MEANT_LOADER_THREADS", "8")), (os.cpu_count() or 2) // 2))
        self._pool = ThreadPoolExecutor(max_workers=self._workers)

    def close(self):
        """undo the in-place page-locking of the source arrays (also done on garbage collection)"""
        if any(self._registered):
            if self._stream is not None:
                self._stream.synchronize()               # rows still being DMA'd out of the page-locked arrays must have left them
            rt = torch.cuda.cudart()
            for i, a in enumerate(self.arrays):
                if self._registered[i]:
                    try:
                        rt.cudaHostUnregister(a.ctypes.data)
                    except Exception:
                        pass
                    self._registered[i] = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return self.n // (self.batch_size * self.world)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    # -- staging --------------------------------------------------------------------------------
    def _alloc(self):
        def pinned(a):
            if a is None:
                return None
            t = torch.empty((self.batch_size, *a.shape[1:]), dtype=torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype)
            return t.pin_memory() if self.on_gpu else t
        # two host staging sets and two device sets: while set k is being consumed, set k^1 is filled and sent
        self._stage = [[pinned(a) for a in self.arrays] for _ in range(2)]
        self._dev = [[None if t is None else torch.empty_like(t, device=self.device) for t in s] for s in self._stage]
        self._ready = [None, None]

    def _fill_and_send(self, k: int, idx: np.ndarray):
        # multi-threaded gather straight into the pinned buffers (no intermediate copy, no conversion): each worker
        # takes a contiguous slice of the batch.  Page-locked source arrays skip this: see below.
        nw = self._workers
        bounds = [(i * len(idx)) // nw for i in range(nw + 1)]
        staged = [a is not None and not self._registered[i] for i, a in enumerate(self.arrays)]

        def part(w):
            lo, hi = bounds[w], bounds[w + 1]
            if hi > lo:
                sel = torch.from_numpy(np.ascontiguousarray(idx[lo:hi]))
                for a, host, st in zip(self.arrays, self._stage[k], staged):
                    if st:
                        torch.index_select(torch.from_numpy(a) if not isinstance(a, torch.Tensor) else a, 0, sel, out=host[lo:hi])
        if any(staged):
            list(self._pool.map(part, range(nw)))
        if not self.on_gpu:
            self._dev[k] = [None if h is None else h.clone() for h in self._stage[k]]
            return
        with torch.cuda.stream(self._stream):
            for i, (a, host, dev) in enumerate(zip(self.arrays, self._stage[k], self._dev[k])):
                if host is None:
                    continue
                if self._registered[i]:                       # DMA row by row out of the page-locked data set
                    src = torch.from_numpy(a)
                    for j, r in enumerate(idx):
                        dev[j].copy_(src[int(r)], non_blocking=True)
                else:
                    dev.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._ready[k] = ev

    # -- iteration ------------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Sequence[Optional[torch.Tensor]]]:
        if self._stage is None:
            self._alloc()
        idx = shard_indices(self.n, self.rank, self.world, self.batch_size, self.shuffle, self.seed, self.epoch)
        nb = len(idx) // self.batch_size
        consumed = [None, None]                          # event: the compute stream is done with device set k
        pending = None                                   # producer thread preparing the next batch

        def produce(k, b):
            if self.on_gpu:
                torch.cuda.set_device(self.device)
                if consumed[k] is not None:
                    self._ready[k].synchronize()             # the pinned set is free once its last copy has left it
                    self._stream.wait_event(consumed[k])     # the device set once the model is done reading it
            self._fill_and_send(k, idx[b * self.batch_size:(b + 1) * self.batch_size])

        if nb:
            produce(0, 0)
        try:
            for b in range(nb):
                k = b & 1
                if pending is not None:
                    pending.join()
                    pending = None
                if self.on_gpu:
                    torch.cuda.current_stream(self.device).wait_event(self._ready[k])
                if b + 1 < nb:                            # batch b+1 is gathered and sent while the caller runs step b
                    pending = threading.Thread(target=produce, args=(k ^ 1, b + 1), daemon=True)
                    pending.start()
                yield tuple(self._dev[k])
                if self.on_gpu:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    consumed[k] = ev
            self.epoch += 1
        finally:
            # also when the caller leaves the loop early (break, exception, the generator being collected): the producer
            # thread must not outlive the iteration -- it copies into staging buffers that die with this object -- and its
            # copies must have left the copy stream before those buffers can be freed
            if pending is not None:
                pending.join()
            if self.on_gpu:
                self._stream.synchronize()
