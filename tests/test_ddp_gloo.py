"""Multi-process data-parallel path on the CPU: world_size 2, gloo backend, HIP calls replaced by the oracle-backed
stand-in (tests/fake_native.py).  Checks the sharding helpers, the rank-aware loader, the parameter/buffer broadcast,
that the single flat all-reduce yields the gradient of the GLOBAL batch mean -- for equal shards (2 + 2 events) and for
unequal ones (1 + 3 events, shards balanced by cost) -- and that ranks stay bit-identical after the optimizer step."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [60, 35, 50, 44]      # 4 events, 2 per rank
K = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(rank_events):
    from deepmetv2_amd import synth
    x, y, batch, ptr = synth.make_events(SIZES, seed=5)
    keep = torch.isin(batch, torch.tensor(rank_events))
    xs, bs = x[keep], batch[keep]
    remap = {e: i for i, e in enumerate(rank_events)}
    bs = torch.tensor([remap[int(b)] for b in bs])
    return xs.contiguous(), y[rank_events].contiguous(), bs


UNEQUAL_SIZES = [60, 20, 25, 22]      # by cost (n^2): rank 0 takes event 0, rank 1 the other three


def _worker(rank, world, port, outdir, mode="equal"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fake_native
    fake_native.install()
    from deepmetv2_amd import data, synth
    from deepmetv2_amd.model import Net
    from deepmetv2_amd.parallel import FlatModule, GradSync, shard_range, train_step
    torch.manual_seed(100 + rank)                       # different initial weights per rank: broadcast must fix it
    model = Net(8, 3, graph="dynamic", k=K).train()
    flat = FlatModule(model)
    sync = GradSync(flat)
    sync.broadcast_state(0)
    p0 = flat.flat_param.detach().clone()
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3)
    if mode == "equal":
        events = list(shard_range(len(SIZES), rank, world))
        x, y, batch = _make(events)
        loss = train_step(model, flat, sync, opt, x, y, batch)
    else:
        # the rank-aware loader: one global batch of 4 events, shards balanced by cost -> 1 + 3 events
        evs = _unequal_events()
        loader = data.EventLoader(evs, batch_size=4, rank=rank, world=world, balance="cost")
        events = loader.shard(list(range(4)))
        (b,) = list(loader)
        assert b.global_graphs == 4 and b.num_graphs == len(events)
        loss = train_step(model, flat, sync, opt, b.x, b.y, b.batch, b.ptr, global_events=b.global_graphs)
    torch.save({"p0": p0, "grad": flat.flat_grad.clone(), "p1": flat.flat_param.detach().clone(), "loss": loss,
                "events": events}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _unequal_events():
    from deepmetv2_amd import synth
    x, y, batch, ptr = synth.make_events(UNEQUAL_SIZES, seed=9)
    return [(x[ptr[e]:ptr[e + 1]].contiguous(), y[e:e + 1].contiguous()) for e in range(len(UNEQUAL_SIZES))]


def test_rank_aware_loader_shards():
    from deepmetv2_amd import data
    evs = [(torch.zeros(n, 11), torch.zeros(1, 11)) for n in [8000, 500, 7000, 600, 4000, 4100, 30, 20, 10]]
    for balance in ("count", "cost"):
        per_rank = [data.EventLoader(evs, batch_size=6, rank=r, world=2, balance=balance) for r in range(2)]
        assert len(per_rank[0]) == len(per_rank[1]) == 2
        batches = [list(l) for l in per_rank]
        for step, ids in enumerate([[0, 1, 2, 3, 4, 5], [6, 7, 8]]):
            shards = [l.shard(ids) for l in per_rank]
            assert sorted(shards[0] + shards[1]) == ids                  # a partition of the global batch
            for r in range(2):
                b = batches[r][step]
                assert b.global_graphs == len(ids) and b.num_graphs == len(shards[r])
                assert b.num_nodes == sum(evs[i][0].shape[0] for i in shards[r])
    # by cost: the two heaviest events go to different ranks; by count: contiguous halves
    assert data.EventLoader(evs, 6, rank=0, world=2, balance="count").shard([0, 1, 2, 3, 4, 5]) == [0, 1, 2]
    by_cost = [data.EventLoader(evs, 6, rank=r, world=2, balance="cost").shard([0, 1, 2, 3, 4, 5]) for r in range(2)]
    assert 0 in by_cost[0] and 2 in by_cost[1]
    # a trailing global batch smaller than the world is dropped; a single process keeps everything and reports no share
    assert len(data.EventLoader(evs, batch_size=4, rank=0, world=4)) == 2
    single = list(data.EventLoader(evs, batch_size=4))
    assert len(single) == 3 and all(b.global_graphs is None for b in single)
    with pytest.raises(ValueError):
        data.EventLoader(evs, batch_size=2, rank=0, world=4)


def test_shard_helpers():
    from deepmetv2_amd.parallel import balanced_shards, shard_range
    assert [list(shard_range(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    assert list(shard_range(512, 7, 8)) == list(range(448, 512))
    sh = balanced_shards([n * n for n in [8000, 500, 7000, 600, 4000, 4100]], 2)
    assert sorted(sum(sh, [])) == list(range(6)) and 0 in sh[0] and 2 in sh[1]


@pytest.mark.timeout(600)
def test_two_rank_gloo_training_step(tmp_path):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    assert r[0]["events"] == [0, 1] and r[1]["events"] == [2, 3]
    assert torch.equal(r[0]["p0"], r[1]["p0"])                      # broadcast from rank 0
    assert torch.equal(r[0]["grad"], r[1]["grad"])                  # all-reduced gradient is identical everywhere
    assert torch.equal(r[0]["p1"], r[1]["p1"])                      # ranks stay in lock step after AdamW
    assert not torch.equal(r[0]["p0"], r[0]["p1"])

    # the all-reduced gradient is the mean of the two single-process shard gradients (per-rank BatchNorm statistics)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_native
    fake_native.install()
    from deepmetv2_amd.model import Net, loss_fn, split_features
    from deepmetv2_amd.parallel import FlatModule
    grads = []
    for events in ([0, 1], [2, 3]):
        torch.manual_seed(0)
        model = Net(8, 3, graph="dynamic", k=K).train()
        flat = FlatModule(model)
        with torch.no_grad():
            flat.flat_param.copy_(r[0]["p0"])
        x, y, batch = _make(events)
        loss_fn(model(*split_features(x), None, batch), x, y, batch).backward()
        flat.gather_grads()
        grads.append(flat.flat_grad.clone())
    expect = (grads[0] + grads[1]) / 2
    torch.testing.assert_close(r[0]["grad"], expect, rtol=1e-5, atol=1e-6 * float(expect.abs().max()))


@pytest.mark.timeout(600)
def test_two_rank_gloo_unequal_shares(tmp_path):
    """1 + 3 events (shards balanced by cost): the all-reduced gradient must be the gradient of the GLOBAL batch mean
    (model/net.py:60: mean over the 4 events), not the mean of the two rank means -- checked against a single process that
    forwards each shard on its own (per-rank BatchNorm statistics, as the ranks do) and differentiates the global mean."""
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), "unequal"), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    assert r[0]["events"] == [0] and r[1]["events"] == [1, 2, 3]
    assert torch.equal(r[0]["grad"], r[1]["grad"]) and torch.equal(r[0]["p1"], r[1]["p1"])

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_native
    fake_native.install()
    from deepmetv2_amd import data
    from deepmetv2_amd.model import Net, loss_fn, split_features
    from deepmetv2_amd.parallel import FlatModule
    evs = _unequal_events()
    torch.manual_seed(0)
    model = Net(8, 3, graph="dynamic", k=K).train()
    flat = FlatModule(model)
    with torch.no_grad():
        flat.flat_param.copy_(r[0]["p0"])
    total = 0.0
    for events in ([0], [1, 2, 3]):
        b = data.collate([evs[i] for i in events])
        # sum of the per-event losses of the shard = its mean x its event count
        total = total + loss_fn(model(*split_features(b.x), None, b.batch), b.x, b.y, b.batch) * len(events)
    (total / 4).backward()
    flat.gather_grads()
    expect = flat.flat_grad.clone()
    torch.testing.assert_close(r[0]["grad"], expect, rtol=1e-5, atol=1e-6 * float(expect.abs().max()))
    # and it is NOT the unweighted mean of the rank gradients (what dividing by the world size would give)
    assert float(r[0]["loss"]) != float(r[1]["loss"])
