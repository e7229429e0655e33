"""Dataset container, collate and a loader that keeps up with the training step.

Interfaces of ``e3_layers/data/dataset.py:22-121`` (``CondensedDataset``: ONE Batch holding every sample, key map,
``dataset[i]`` -> preprocessed ``Data``, ``dataset[ids]`` -> sub-dataset, loading from a file / list of files / directory
/ ``directory:regexp``) and ``e3_layers/data/dataloader.py:13-118`` (``Collater``, ``DataLoader``, ``getDataIters``:
per-rank path sharding, train / validation split, auto-resetting iterators).

What is different, and why: the reference collates every batch from ``batch_size`` ``Data`` objects in Python
(``Batch.from_data_list``: ~6 ms per 256 molecules here -- longer than the 5.7 ms training step it feeds).
``PrefetchLoader`` assembles a batch with ONE vectorised gather per tensor from the condensed store
(``Batch.index_select``: element ranges of the chosen graphs by segment arithmetic, ``edge_index`` re-based in the same
pass), bit-identical to collating the same samples one by one, on a worker thread that runs ahead of the step: pinned
host buffers, asynchronous host-to-device copies on a copy stream, the consumer's stream waits on an event (no host
sync).  ``bench.py --loader`` measures the step fed this way next to the HBM-resident figure.

On-disk format: ``.npz`` shards (one condensed Batch each: its tensors plus ``__attrs__``, JSON) -- numpy is what this
image has; ``.h5`` files with the reference's layout (one dataset per key, attributes = ``attrs``) load when ``h5py`` is
importable and fail loudly when it is not.
"""
from __future__ import annotations

import atexit
import json
import math
import os
import queue
import re
import threading
import weakref
from inspect import signature
from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from .data import Batch, Data

_SKIP = ("_node_segment", "_edge_segment")


# ---------------------------------------------------------------------------------------------------------------------
# files
# ---------------------------------------------------------------------------------------------------------------------
def save_npz(batch: Batch, path: str) -> None:
    """One condensed Batch -> one ``.npz`` shard."""
    arrays = {k: v.detach().cpu().numpy() for k, v in batch.data.items() if k not in _SKIP and not k.startswith("_e3k_")}
    attrs = {k: list(v) for k, v in batch.attrs.items() if k in arrays}
    np.savez(path, __attrs__=np.frombuffer(json.dumps(attrs).encode(), dtype=np.uint8), **arrays)


def _load_file(file: str):
    """(tensors, attrs) of one file; int32 -> int64 and float64 -> float32 as the reference does (``dataset.py:55-60``)."""
    data, attrs = {}, {}
    if file.endswith(".npz"):
        with np.load(file) as z:
            for key in z.files:
                if key == "__attrs__":
                    attrs = {k: tuple(v) for k, v in json.loads(bytes(z[key]).decode()).items()}
                    continue
                data[key] = torch.from_numpy(z[key])
    elif file.endswith((".h5", ".hdf5")):
        try:
            import h5py
        except ImportError as exc:      # (absent from this image; the reference requires it: dataset.py:9)
            raise RuntimeError(f"{file}: reading HDF5 datasets needs h5py, which is not installed; convert to .npz shards "
                               "with e3_layers_amd.data.loader.save_npz") from exc
        with h5py.File(file, "r") as f:
            for key in f.keys():
                data[key] = torch.tensor(f[key][:])
            for key in f.attrs.keys():
                attrs[key] = tuple(f.attrs[key])
    else:
        raise ValueError(f"{file}: unknown dataset file type (.npz shards or .h5)")
    for key, item in data.items():
        if item.dtype == torch.int32:
            data[key] = item.long()
        elif item.dtype == torch.float64:
            data[key] = item.float()
    return data, attrs


def load_path(path):
    """``path``: a file, a list / tuple of paths, a directory (walked), or ``directory:regexp`` -- ``dataset.py:49-104``.
    Returns (list of per-file tensor dicts, merged attrs)."""
    if isinstance(path, (list, tuple)):
        data, attrs = [], {}
        for item in path:
            d, a = load_path(item)
            data += d
            attrs.update(a)
        return data, attrs
    regexp = None
    if ":" in path and not os.path.exists(path):
        path, pattern = path.split(":", 1)
        regexp = re.compile(pattern)
    if os.path.isdir(path):
        data, attrs = [], {}
        for root, _, files in sorted(os.walk(path)):
            for name in sorted(files):
                file = os.path.join(root, name)
                if regexp is not None and regexp.match(file) is None:
                    continue
                if not file.endswith((".npz", ".h5", ".hdf5")):
                    continue
                d, a = _load_file(file)
                data.append(d)
                attrs.update(a)
        return data, attrs
    d, a = _load_file(path)
    return [d], a


# ---------------------------------------------------------------------------------------------------------------------
# CondensedDataset
# ---------------------------------------------------------------------------------------------------------------------
def _map_keys(d: dict, key_map: dict) -> dict:
    return {key_map.get(k, k): v for k, v in d.items()}


class CondensedDataset(Batch):
    """A ``Batch`` holding every sample, with key mapping and per-sample preprocessing (``dataset.py:22-121``)."""

    def __init__(self, path=None, data=None, attrs=None, key_map=None, type_names=None, preprocess=(), **kwargs):
        data, attrs, key_map = dict(data or {}), dict(attrs or {}), dict(key_map or {})
        if path is not None:
            parts, attrs = load_path(path)
            if not parts:
                raise FileNotFoundError(f"no dataset file found in {path}")
            if len(parts) == 1:
                data = parts[0]
            else:       # several condensed files: graphs of file k follow those of file k-1
                whole = [Batch({k: v for k, v in attrs.items() if k in p}, **p) for p in parts]
                data = Batch.from_data_list([b.data for b in whole], dict(attrs)).data
        data = {k: v for k, v in data.items() if k not in _SKIP}
        super().__init__(_map_keys({k: (v[0], v[1]) for k, v in attrs.items()}, key_map), **_map_keys(data, key_map))
        self.type_names = list(type_names) if type_names is not None else None
        self.preprocess = list(preprocess)
        self.kwargs = kwargs

    def _apply_preprocess(self, sample: Data) -> Data:
        for func in self.preprocess:
            if len(signature(func).parameters) == 1:
                sample = func(sample)
            else:
                sample.data, sample.attrs = func(sample.data, sample.attrs)
        return sample

    def __getitem__(self, idx):
        if isinstance(idx, str):
            return self.data[idx]
        if isinstance(idx, (int, np.integer)):
            return self._apply_preprocess(self.get(idx).clone())
        return self.index_select(idx)

    def index_select(self, idx):
        sub = super().index_select(idx)
        out = CondensedDataset(data=sub.data, attrs=sub.attrs, type_names=self.type_names, preprocess=self.preprocess,
                               **self.kwargs)
        return out


def samples_of(batch: Batch) -> List[Data]:
    """The graphs of a batch as individual host samples (what a dataset hands a collate function)."""
    host = batch if batch["_n_nodes"].device.type == "cpu" else batch.clone().to("cpu")
    return [host.get(i) for i in range(len(host))]


# ---------------------------------------------------------------------------------------------------------------------
# the reference's loader shape: Collater + torch DataLoader (worker PROCESSES collate Data objects)
# ---------------------------------------------------------------------------------------------------------------------
class Collater:
    @classmethod
    def for_dataset(cls, dataset):
        return cls()

    def collate(self, batch: List[Data]) -> Batch:
        return Batch.from_data_list(batch, attrs=dict(batch[0].attrs))

    __call__ = collate


class DataLoader(torch.utils.data.DataLoader):
    """``e3_layers/data/dataloader.py:28-45``: a torch DataLoader whose collate is ``Batch.from_data_list``.  Worker
    processes never touch the GPU; start them before the first HIP call of the parent (fork) or use ``spawn``."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, **kwargs):
        super().__init__(dataset, batch_size, shuffle, collate_fn=Collater.for_dataset(dataset), **kwargs)


# ---------------------------------------------------------------------------------------------------------------------
# the loader that keeps up
# ---------------------------------------------------------------------------------------------------------------------
class PrefetchLoader:
    """Batches of ``batch_size`` graphs drawn from a condensed store, assembled ahead of the consumer.

    ``source``: a ``Batch`` / ``CondensedDataset`` on the host, or a list of ``Data`` samples (condensed once here).
    A worker thread draws the ids of the next batch (``shuffle``: a fresh permutation per epoch from ``seed``; the same
    ``torch.Generator`` semantics as the reference's ``loader_rng``, ``dataloader.py:83-86``), gathers the batch with
    ``index_select`` (vectorised; bit-identical to ``Batch.from_data_list`` over the same samples in the same order),
    pins it and -- when ``device`` is a GPU -- enqueues its host-to-device copies on a dedicated copy stream; ``depth``
    batches are kept in flight.  ``__next__`` makes the consumer's current stream wait for the copies' event and hands
    over a Batch nothing else references.  ``epochs=None``: endless (auto-reset, ``dataloader.py:104-113``).
    ``drop_last`` as the reference's ``dl_kwargs`` (True).  Per-sample ``preprocess`` hooks of a CondensedDataset (the
    protein crop) are Python per sample by nature: datasets that carry them go through ``DataLoader``."""

    def __init__(self, source, batch_size: int, device=None, shuffle: bool = True, seed: int = 0, drop_last: bool = True,
                 epochs: Optional[int] = 1, depth: int = 3):
        if isinstance(source, Batch):
            store = source
        else:
            samples = list(source)
            store = Batch.from_data_list([s.data if isinstance(s, Data) else s for s in samples],
                                         dict(samples[0].attrs) if isinstance(samples[0], Data) else None)
        if getattr(store, "preprocess", None):
            raise ValueError("PrefetchLoader gathers batches from the condensed tensors; per-sample preprocess hooks need "
                             "DataLoader (worker processes)")
        if store["_n_nodes"].device.type != "cpu":
            raise ValueError("the store lives on the host (batches are copied to the device as they are assembled)")
        self.store, self.batch_size = store, int(batch_size)
        self.device = torch.device(device) if device is not None else None
        self.shuffle, self.seed, self.drop_last, self.epochs, self.depth = bool(shuffle), int(seed), bool(drop_last), epochs, int(depth)
        self.n = len(store)
        # (a split that cannot fill one batch -- n_val = 0 -- fails when it is iterated, as the reference's DataLoader does, not here)

    def __len__(self) -> int:
        per = self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)
        return per if self.epochs is None else per * self.epochs

    def id_batches(self) -> Iterator[List[int]]:
        """The graph ids of every batch, in order (what the worker consumes; exposed for the parity test)."""
        gen = torch.Generator()
        gen.manual_seed(self.seed)
        epoch = 0
        while self.epochs is None or epoch < self.epochs:
            order = torch.randperm(self.n, generator=gen) if self.shuffle else torch.arange(self.n)
            stop = self.n - self.batch_size + 1 if self.drop_last else self.n
            for a in range(0, stop, self.batch_size):
                yield order[a:a + self.batch_size].tolist()
            epoch += 1

    def assemble(self, ids: Sequence[int]) -> Batch:
        """One batch on the host (the worker's first half; also the oracle-free reference point of the tests)."""
        return self.store.index_select(list(ids))

    def __iter__(self):
        return _PrefetchIter(self)


_LIVE_ITERS: "weakref.WeakSet" = weakref.WeakSet()      # iterators whose worker thread has been started


def _close_live_iters() -> None:
    for it in list(_LIVE_ITERS):
        try:
            it.close()
        except Exception:
            pass


atexit.register(_close_live_iters)      # a worker that still holds a HIP stream at interpreter shutdown aborts the process


class _PrefetchIter:
    """The worker thread (and its HIP copy stream) start with the first ``next()``: an iterator that is created and never
    used -- ``getDataIters`` hands out the eval iterator eagerly -- costs nothing and needs no ``close()``; live ones are closed
    at interpreter exit (``atexit``) if the caller did not."""

    def __init__(self, loader: PrefetchLoader):
        self.loader = loader
        self.q: "queue.Queue" = queue.Queue(maxsize=max(loader.depth, 1))
        self.stop = threading.Event()
        dev = loader.device
        self.cuda = dev is not None and dev.type == "cuda"
        self.copy_stream = None
        self.thread = None

    def _start(self) -> None:
        loader = self.loader
        if loader.n < loader.batch_size and loader.drop_last:
            raise ValueError(f"{loader.n} graphs cannot fill a batch of {loader.batch_size}")
        self.copy_stream = torch.cuda.Stream(device=loader.device) if self.cuda else None
        self.thread = threading.Thread(target=self._work, name="e3k-prefetch", daemon=True)
        _LIVE_ITERS.add(self)
        self.thread.start()

    def _stage(self, slot: dict, key: str, t: torch.Tensor) -> torch.Tensor:
        """``t`` copied into the slot's pinned buffer for ``key`` (grown geometrically; pinned allocations cost
        milliseconds each, so they happen a handful of times per run, not ten times per batch)."""
        buf = slot.get(key)
        if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
            buf = slot[key] = torch.empty(max(int(t.numel() * 1.25), 16), dtype=t.dtype).pin_memory()
        view = buf[:t.numel()].view(t.shape)
        np.copyto(view.numpy(), t.numpy())      # (numpy: a torch CPU copy of this size wakes the intra-op thread pool)
        return view

    def _work(self):
        loader = self.loader
        try:
            if self.cuda:
                torch.cuda.set_device(loader.device)
            # a ring of pinned staging sets: one more than the batches that can be in flight, each guarded by the event of
            # the copies last issued from it
            ring = [({}, None) for _ in range(max(loader.depth, 1) + 2)]
            turn = 0
            for ids in loader.id_batches():
                if self.stop.is_set():
                    return
                batch = loader.assemble(ids)
                ready = None
                if self.cuda:
                    slot, busy = ring[turn % len(ring)]
                    if busy is not None:
                        busy.synchronize()           # (this worker thread only: the copies from this slot have landed)
                    with torch.cuda.stream(self.copy_stream):
                        for key in list(batch.data.keys()):
                            batch.data[key] = self._stage(slot, key, batch.data[key]).to(loader.device, non_blocking=True)
                        batch.device = loader.device
                        ready = torch.cuda.Event()
                        ready.record(self.copy_stream)
                    ring[turn % len(ring)] = (slot, ready)
                    turn += 1
                elif loader.device is not None:
                    batch.to(loader.device)
                while not self.stop.is_set():
                    try:
                        self.q.put((batch, ready), timeout=0.1)
                        break
                    except queue.Full:
                        continue
            self.q.put((None, None))
        except BaseException as exc:      # surfaces in the consumer
            self.q.put((exc, None))

    def __iter__(self):
        return self

    def __next__(self) -> Batch:
        if self.thread is None:
            if self.stop.is_set():
                raise StopIteration
            self._start()
        batch, ready = self.q.get()
        if batch is None:
            raise StopIteration
        if isinstance(batch, BaseException):
            raise batch
        if ready is not None:
            cur = torch.cuda.current_stream(self.loader.device)
            cur.wait_event(ready)
            for t in batch.data.values():      # allocated on the copy stream, used on the consumer's
                t.record_stream(cur)
        return batch

    def close(self):
        """Stops the worker and waits for it (an endless loader must be closed before the interpreter exits: a thread
        that still holds a HIP stream at shutdown aborts the process)."""
        self.stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        if self.thread is not None and self.thread.is_alive() and threading.current_thread() is not self.thread:
            self.thread.join(timeout=10)
        _LIVE_ITERS.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------------
# getDataIters
# ---------------------------------------------------------------------------------------------------------------------
def _auto_reset(make_iter):
    it = make_iter()
    while True:
        try:
            batch = next(it)
        except StopIteration:
            it = make_iter()
            batch = next(it)
        yield batch


def getDataIters(config, rank: int = 0, world_size: int = 1, seed: int = 0, device=None, num_workers: int = 0):
    """``e3_layers/data/dataloader.py:47-118`` without absl FLAGS (rank / world size / seed / workers are arguments):
    ``config.data_config`` holds ``path`` (split among ranks by the gcd rule when it is a list), ``n_train`` / ``n_val``
    (counts or fractions), ``train_val_split`` ("random" | "sequential") and the CondensedDataset kwargs; returns the
    auto-resetting (train, eval) iterators.  Datasets without per-sample preprocess hooks are served by
    ``PrefetchLoader`` (the batches arrive on ``device``); the others by ``DataLoader`` with ``num_workers`` processes."""
    dc = dict(config.data_config)
    path = dc.get("path")
    if isinstance(path, (tuple, list)):
        gcd = math.gcd(world_size, len(path))
        per = len(path) // gcd
        dc["path"] = list(path[(rank % gcd) * per:(rank % gcd + 1) * per])
    n_train, n_val, mode = dc.pop("n_train"), dc.pop("n_val"), dc.pop("train_val_split", "random")
    dc.pop("std", None)
    dataset = CondensedDataset(**dc)
    total = len(dataset)
    if isinstance(n_train, float):
        n_train = int(n_train * total)
    if isinstance(n_val, float):
        n_val = int(n_val * total)
    if n_train + n_val > total:
        raise ValueError("too little data for training and validation. please reduce n_train and n_val")
    if mode == "random":
        idcs = torch.randperm(total, generator=torch.Generator().manual_seed(seed))
    elif mode == "sequential":
        idcs = torch.arange(total)
    else:
        raise NotImplementedError(f"splitting mode {mode} not implemented")
    train_ds, eval_ds = dataset.index_select(idcs[:n_train]), dataset.index_select(idcs[n_train:n_train + n_val])
    bs = int(config.batch_size)

    def iters(ds, shuffle):
        if ds.preprocess:
            gen = torch.Generator().manual_seed(seed + rank)
            kw = dict(batch_size=bs, num_workers=num_workers, pin_memory=device is not None, generator=gen, drop_last=True,
                      timeout=300 if num_workers > 0 else 0)
            return _auto_reset(lambda: iter(DataLoader(ds, shuffle=shuffle, **kw)))
        loader = PrefetchLoader(ds, bs, device=device, shuffle=shuffle, seed=seed + rank, drop_last=True, epochs=None)
        return iter(loader)

    return iters(train_ds, True), iters(eval_ds, False)
