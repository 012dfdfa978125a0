"""Ragged batch container against the event-by-event oracle restatement of METDataset.process."""
import numpy as np
import torch

from deepmetv2_amd import data
from oracle import ref_ops


def _fake_file(n_evt=7, n_max=40, seed=0):
    rng = np.random.default_rng(seed)
    x = np.full((12, n_evt, n_max), -999.0, dtype=np.float32)
    sizes = rng.integers(0, n_max + 1, n_evt)
    sizes[0], sizes[1] = 0, n_max                       # empty and full events
    for e, n in enumerate(sizes):
        x[0, e, :n] = rng.exponential(2.0, n) * rng.choice([1, 4000], n, p=[0.97, 0.03])   # some beyond the clip
        x[1, e, :n] = rng.uniform(-5, 5, n)
        x[2, e, :n] = rng.uniform(-np.pi, np.pi, n)
        x[3:7, e, :n] = rng.normal(0, 1, (4, n))
        x[7, e, :n] = rng.choice([211, -211, 130, 22, 11, -13, 1, 2], n)
        x[8, e, :n] = rng.choice([-1, 0, 1], n)
        x[9, e, :n] = rng.integers(0, 4, n)
        x[10:, e, :n] = rng.integers(0, 3, (2, n))
    x[4, 2, 0] = np.nan                                  # nan_to_num path
    y = rng.normal(0, 30, (n_evt, 11)).astype(np.float32)
    return x, y, sizes


def test_decode_matches_event_by_event_restatement():
    x, y, sizes = _fake_file()
    got = data.events_from_padded(x, y)
    ref = ref_ops.decode_padded_events(x, y)
    assert len(got) == len(ref) == len(sizes)
    for (gx, gy), (rx, ry), n in zip(got, ref, sizes):
        assert gx.shape == (n, 11) and torch.equal(gx, rx) and torch.equal(gy, ry)
    assert float(torch.cat([g[0] for g in got]).abs().max()) <= 5000.0


def test_collate_and_loader():
    x, y, sizes = _fake_file(n_evt=9, seed=3)
    events = data.events_from_padded(x, y)
    b = data.collate(events[:4])
    assert b.num_graphs == 4 and b.num_nodes == int(sizes[:4].sum()) and b.max_nodes == int(sizes[:4].max())
    assert b.ptr.tolist() == [0] + np.cumsum(sizes[:4]).tolist()
    assert torch.equal(b.batch, torch.repeat_interleave(torch.arange(4), torch.tensor(sizes[:4])))
    assert torch.equal(b.x[b.ptr[1]:b.ptr[2]], events[1][0]) and torch.equal(b.y[2:3], events[2][1])
    loaders = data.EventLoader.split(events, batch_size=2, validation_split=0.2, seed=42)
    assert len(loaders["train"].indices) == 8 and len(loaders["test"].indices) == 1
    assert sorted(loaders["train"].indices + loaders["test"].indices) == list(range(9))
    seen = sum(bt.num_graphs for bt in loaders["train"])
    assert seen == 8 and len(loaders["train"]) == 4


def test_graph_registry_does_not_leak(monkeypatch):
    """radius_graph / knn_graph hand out a [2,E] tensor and remember its table so that EdgeConv can find it again: once
    the caller drops the tensor, the table and the registry entry must go too (no garbage collector involved)."""
    import gc
    import weakref
    import torch
    from tests import fake_native
    fake_native.install(monkeypatch)
    import deepmetv2_amd as dm
    from deepmetv2_amd import graph
    gc.disable()
    try:
        before = len(graph._graph_registry)
        x = torch.rand(40, 2)
        batch = torch.zeros(40, dtype=torch.int64)
        ei = dm.radius_graph(x, 0.3, batch, loop=True, max_num_neighbors=16)
        table, _flow = graph.lookup_graph(ei)
        wt = weakref.ref(table)
        del table
        assert len(graph._graph_registry) == before + 1 and wt() is not None
        del ei
        assert len(graph._graph_registry) == before and wt() is None
    finally:
        gc.enable()
