"""Multi-process data-parallel path on the CPU: world_size 2, gloo backend, HIP calls replaced by the oracle-backed
stand-in (tests/fake_native.py).  Checks the sharding helpers, the parameter/buffer broadcast, that the single flat
all-reduce yields the mean of the per-rank gradients, and that ranks stay bit-identical after the optimizer step."""
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


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fake_native
    fake_native.install()
    from deepmetv2_amd.model import Net
    from deepmetv2_amd.parallel import FlatModule, GradSync, shard_range, train_step
    torch.manual_seed(100 + rank)                       # different initial weights per rank: broadcast must fix it
    model = Net(8, 3, graph="dynamic", k=K).train()
    flat = FlatModule(model)
    sync = GradSync(flat)
    sync.broadcast_state(0)
    p0 = flat.flat_param.detach().clone()
    events = list(shard_range(len(SIZES), rank, world))
    x, y, batch = _make(events)
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3)
    loss = train_step(model, flat, sync, opt, x, y, batch)
    torch.save({"p0": p0, "grad": flat.flat_grad.clone(), "p1": flat.flat_param.detach().clone(), "loss": loss,
                "events": events}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


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
