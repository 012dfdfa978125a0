"""Child process of tests/test_gpu_ddp.py: one of TWO ranks that share the box's one GPU.  The process group is gloo
(RCCL cannot put two ranks on one device), everything else is the product path: the HIP library, the rank-aware
loader with shards balanced by cost (1 + 3 events), the backward seeded with the rank's share, the SUM all-reduce of
the flat gradient, FlatAdamW, and the two-hipGraph step around the eager collective.  Each rank writes what it saw to
`$DMET_OUT/rank<r>.pt`; rank 0 also differentiates the global batch mean in one process for the expected gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = [1500, 500, 520, 480]     # by cost (n^2): one rank takes event 0, the other the remaining three
K = 16


def main():
    import torch
    import torch.distributed as dist

    from deepmetv2_amd import data, synth
    from deepmetv2_amd.model import Net, loss_fn, split_features
    from deepmetv2_amd.optim import FlatAdamW
    from deepmetv2_amd.parallel import FlatModule, GradSync, GraphedTrainStep, train_step

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    x, y, batch, ptr = synth.make_events(SIZES, seed=21)
    evs = [(x[ptr[e]:ptr[e + 1]].contiguous(), y[e:e + 1].contiguous()) for e in range(len(SIZES))]
    loader = data.EventLoader(evs, batch_size=len(SIZES), device=dev, rank=rank, world=world, balance="cost")
    mine = loader.shard(list(range(len(SIZES))))
    (b,) = list(loader)
    assert b.global_graphs == len(SIZES) and b.num_graphs == len(mine)

    torch.manual_seed(100 + rank)                # different initial weights per rank: the broadcast must fix it
    model = Net(8, 3, graph="dynamic", k=K).to(dev).train()
    flat = FlatModule(model)
    sync = GradSync(flat)
    sync.broadcast_state(0)
    p0 = flat.flat_param.detach().clone()
    buf0 = [t.detach().clone() for t in flat.buffers()]
    opt = FlatAdamW([flat.flat_param], lr=1e-3)
    loss = train_step(model, flat, sync, opt, b.x, b.y, b.batch, b.ptr, global_events=b.global_graphs)
    out = {"events": mine, "p0": p0.cpu(), "grad": flat.flat_grad.detach().cpu().clone(),
           "p1": flat.flat_param.detach().cpu().clone(), "loss": float(loss)}

    # the step as two hipGraphs around the eager all-reduce, continuing from p1 on both ranks
    sync.set_share(b.num_graphs, b.global_graphs)
    gstep = GraphedTrainStep(model, flat, sync, opt, b.x, b.y, b.batch, b.ptr, warmup=1)
    losses = [float(gstep()) for _ in range(3)]
    torch.cuda.synchronize(dev)
    out["graphed_losses"] = losses
    out["p_graphed"] = flat.flat_param.detach().cpu().clone()
    out["grad_graphed"] = flat.flat_grad.detach().cpu().clone()

    if rank == 0:
        # expected gradient of the first step: one process, each shard forwarded on its own (per-rank BatchNorm
        # statistics, as the ranks have them), d/dparams of the mean over the 4 events
        torch.manual_seed(0)
        ref = Net(8, 3, graph="dynamic", k=K).to(dev).train()
        rflat = FlatModule(ref)
        with torch.no_grad():
            rflat.flat_param.copy_(p0)
            for t, s in zip(rflat.buffers(), buf0):
                t.copy_(s)
        total = 0.0
        for events in ([0], [1, 2, 3]):
            sb = data.collate([evs[i] for i in events]).to(dev)
            total = total + loss_fn(ref(*split_features(sb.x), None, sb.batch), sb.x, sb.y, sb.batch, ptr=sb.ptr) * len(events)
        (total / len(SIZES)).backward()
        rflat.gather_grads()
        out["expect_grad"] = rflat.flat_grad.detach().cpu().clone()
    torch.save(out, os.path.join(os.environ["DMET_OUT"], f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    print(f"GLOO_GPU_CHILD rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
