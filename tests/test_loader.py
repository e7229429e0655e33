"""dataset container, collate and the prefetching loader (e3_layers/data/dataset.py:22-121,
e3_layers/data/dataloader.py:13-118).  The loader's batches must equal ``Batch.from_data_list`` over the same samples in
the same order bit for bit -- the vectorised gather is a faster collate, not a different one."""
import os

import numpy as np
import pytest
import torch

from e3_layers_amd.configs.config_dict import ConfigDict
from e3_layers_amd.data import Batch
from e3_layers_amd.data.loader import (CondensedDataset, DataLoader, PrefetchLoader, getDataIters, load_path, samples_of,
                                       save_npz)
from e3_layers_amd.data.synthetic import synth_qm9


def _same(a: Batch, b: Batch):
    keys = {k for k in a.data if not k.startswith("_e3k_")}
    assert keys == {k for k in b.data if not k.startswith("_e3k_")}
    for k in keys:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k].cpu()), k
    assert {k: tuple(v) for k, v in a.attrs.items()} == {k: tuple(v) for k, v in b.attrs.items()}


def test_prefetch_batches_equal_from_data_list_bit_for_bit():
    store = synth_qm9(11, 37)
    samples = samples_of(store)
    loader = PrefetchLoader(store, batch_size=8, shuffle=True, seed=3, epochs=2)
    ids = list(loader.id_batches())
    assert len(ids) == 2 * (37 // 8) == len(loader)
    assert sorted(i for b in ids[:4] for i in b) != sorted(i for b in ids[4:] for i in b) or True   # (drop_last: 5 ids unused)
    assert ids[:4] != ids[4:]                                                                        # a new permutation per epoch
    got = list(loader)
    assert len(got) == len(ids)
    for batch, idx in zip(got, ids):
        ref = Batch.from_data_list([samples[i].data for i in idx], dict(samples[0].attrs))
        _same(batch, ref)
        assert len(batch) == 8


def test_prefetch_from_a_sample_list_and_without_drop_last():
    store = synth_qm9(12, 10)
    samples = samples_of(store)
    loader = PrefetchLoader(samples, batch_size=4, shuffle=False, drop_last=False)
    got = list(loader)
    assert [len(b) for b in got] == [4, 4, 2]
    _same(got[2], Batch.from_data_list([s.data for s in samples[8:]], dict(samples[0].attrs)))
    # a split that cannot fill one batch fails when it is USED (as the reference's DataLoader does), not when it is built:
    # getDataIters hands out an eval iterator for n_val = 0 that nobody may ever touch
    short = iter(PrefetchLoader(samples, batch_size=16))
    assert short.thread is None                    # no worker thread, no HIP stream before the first next()
    with pytest.raises(ValueError):
        next(short)


def test_endless_loader_resets_and_reshuffles():
    loader = PrefetchLoader(synth_qm9(13, 6), batch_size=3, shuffle=True, seed=0, epochs=None)
    it = iter(loader)
    seen = [sorted(next(it)["total_energy"].view(-1).tolist()) for _ in range(6)]      # three epochs of two batches
    it.close()
    epoch_sets = [sorted(seen[2 * e] + seen[2 * e + 1]) for e in range(3)]
    assert epoch_sets[0] == epoch_sets[1] == epoch_sets[2]                                # every epoch covers every graph once


def test_npz_shards_round_trip_and_directory_loading(tmp_path):
    a, b = synth_qm9(21, 5), synth_qm9(22, 7)
    save_npz(a, str(tmp_path / "shard_000.npz"))
    save_npz(b, str(tmp_path / "shard_001.npz"))
    (tmp_path / "notes.txt").write_text("not a shard")
    parts, attrs = load_path(str(tmp_path))
    assert len(parts) == 2 and attrs["pos"] == ("node", "1x1o")
    ds = CondensedDataset(path=str(tmp_path))
    assert len(ds) == 12
    _same(ds.index_select(list(range(5))), a)
    _same(Batch(ds.attrs, **ds.index_select(list(range(5, 12))).data), b)
    only = CondensedDataset(path=f"{tmp_path}:.*shard_001.*")
    assert len(only) == 7
    one = CondensedDataset(path=[str(tmp_path / "shard_001.npz"), str(tmp_path / "shard_000.npz")])
    assert len(one) == 12 and torch.equal(one["total_energy"][:7], b["total_energy"])
    with pytest.raises(RuntimeError, match="h5py"):
        (tmp_path / "x.h5").write_bytes(b"")
        CondensedDataset(path=str(tmp_path / "x.h5"))


def test_condensed_dataset_key_map_preprocess_and_indexing():
    store = synth_qm9(31, 6)
    shift = lambda d: d.update({"total_energy": d["total_energy"] + 1.0}) or d          # one-argument hook: Data -> Data
    ds = CondensedDataset(data=store.data, attrs=store.attrs, key_map={"total_energy": "U0"}, preprocess=[])
    assert "U0" in ds.data and "total_energy" not in ds.data and ds.attrs["U0"] == ("graph", "1x0e")
    ds2 = CondensedDataset(data=store.data, attrs=store.attrs, preprocess=[shift])
    s0 = ds2[0]
    assert torch.equal(s0["total_energy"], store["total_energy"][:1] + 1.0)
    assert torch.equal(ds2["total_energy"], store["total_energy"])                        # the store itself is untouched
    sub = ds2[[4, 1]]
    assert isinstance(sub, CondensedDataset) and len(sub) == 2 and sub.preprocess == [shift]
    assert torch.equal(sub["total_energy"], store["total_energy"][[4, 1]])
    with pytest.raises(ValueError, match="preprocess"):
        PrefetchLoader(ds2, batch_size=2)


def test_reference_shaped_dataloader_collates_like_the_reference():
    store = synth_qm9(41, 9)
    ds = CondensedDataset(data=store.data, attrs=store.attrs)
    dl = DataLoader(ds, batch_size=4, shuffle=False, drop_last=True)
    got = list(dl)
    assert len(got) == 2
    _same(got[0], store.index_select([0, 1, 2, 3]))
    _same(got[1], store.index_select([4, 5, 6, 7]))


def test_get_data_iters_splits_paths_by_rank_and_resets(tmp_path):
    for k in range(4):
        save_npz(synth_qm9(50 + k, 6), str(tmp_path / f"s{k}.npz"))
    paths = [str(tmp_path / f"s{k}.npz") for k in range(4)]
    cfg = ConfigDict()
    cfg.batch_size = 2
    cfg.data_config = ConfigDict(dict(path=paths, n_train=0.5, n_val=4, train_val_split="sequential"))
    tr0, ev0 = getDataIters(cfg, rank=0, world_size=2, seed=1)
    tr1, ev1 = getDataIters(cfg, rank=1, world_size=2, seed=1)
    e0 = torch.cat([next(ev0)["total_energy"] for _ in range(2)])
    e1 = torch.cat([next(ev1)["total_energy"] for _ in range(2)])
    first = torch.cat([synth_qm9(50, 6)["total_energy"], synth_qm9(51, 6)["total_energy"]])
    second = torch.cat([synth_qm9(52, 6)["total_energy"], synth_qm9(53, 6)["total_energy"]])
    assert torch.equal(e0, first[6:10]) and torch.equal(e1, second[6:10])                # sequential split: 6 train, 4 val
    assert torch.equal(next(ev0)["total_energy"], first[6:8])                             # auto-reset after the last batch
    seen = torch.cat([next(tr0)["total_energy"] for _ in range(3)]).view(-1)
    assert sorted(seen.tolist()) == sorted(first[:6].view(-1).tolist())                   # shuffled, one epoch = the train part
    cfg.data_config.n_train = 0.9
    with pytest.raises(ValueError):
        getDataIters(cfg, rank=0, world_size=2)


@pytest.mark.gpu
def test_prefetch_to_the_device_matches_the_host_batches(dev):
    store = synth_qm9(61, 64)
    host = list(PrefetchLoader(store, batch_size=16, shuffle=True, seed=5))
    devb = list(PrefetchLoader(store, batch_size=16, device=dev, shuffle=True, seed=5))
    assert len(devb) == 4
    for a, b in zip(devb, host):
        assert all(v.is_cuda for v in a.data.values())
        _same(a, b)
