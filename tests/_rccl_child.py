"""Child process of tests/test_gpu_rccl.py: started BEFORE anything touches the GPU, so the RCCL communicator is
created in a fresh process (WORLD_SIZE / RANK / MASTER_* come from the environment).  Runs the same seeded training
steps without a process group and with the `nccl` (= RCCL) group, eager and as hipGraphs, and prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from deepmetv2_amd import register_batch, synth
    from deepmetv2_amd.model import Net
    from deepmetv2_amd.parallel import FlatModule, GradSync, GraphedTrainStep, train_step

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    sizes = [700, 300, 1200, 64, 900, 450]
    x, y, batch, ptr = synth.make_events(sizes, seed=9, device=dev)
    register_batch(batch, ptr, len(sizes), max_nodes=max(sizes))

    def run(steps: int = 3):
        torch.manual_seed(0)
        model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
        flat = FlatModule(model)
        sync = GradSync(flat)
        sync.broadcast_state(0)
        opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        losses = [float(train_step(model, flat, sync, opt, x, y, batch, ptr)) for _ in range(steps)]
        torch.cuda.synchronize(dev)
        return {"losses": losses, "params": flat.flat_param.detach().cpu(), "active": sync.active,
                "ms": (time.perf_counter() - t0) * 1e3 / steps}

    out = {}
    base = run()
    out["nogroup_eager"] = {k: v for k, v in base.items() if k != "params"}
    dist.init_process_group("nccl", device_id=dev)
    out["backend"] = dist.get_backend()
    out["world"] = dist.get_world_size()
    eager = run()
    out["rccl_eager"] = {k: v for k, v in eager.items() if k != "params"}
    out["eager_bitwise_equal"] = bool(torch.equal(base["params"], eager["params"]))
    # a real collective result, not just "did not crash": all_reduce of a known vector
    t = torch.arange(6641, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    out["allreduce_ok"] = bool(torch.equal(t.cpu(), torch.arange(6641, dtype=torch.float32) * dist.get_world_size()))
    # hipGraph form: two graphs around the eager all-reduce; 3 replays must be finite, stay close to the eager losses
    # of the same steps (the graphed run starts from the state its capture warm-up left, so it is compared with its own
    # eager continuation rather than bit for bit) and run at a sane speed
    torch.manual_seed(0)
    model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
    flat = FlatModule(model)
    sync = GradSync(flat)
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True, capturable=True)
    gstep = GraphedTrainStep(model, flat, sync, opt, x, y, batch, ptr, warmup=1)
    snap = flat.flat_param.detach().clone()
    opt_state = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in opt.state.items()}
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    gl = [float(gstep()) for _ in range(3)]
    torch.cuda.synchronize(dev)
    g_ms = (time.perf_counter() - t0) * 1e3 / 3
    g_params = flat.flat_param.detach().cpu()
    # eager continuation from the same snapshot (same optimizer state)
    with torch.no_grad():
        flat.flat_param.copy_(snap)
    for k_, v in opt.state.items():
        for kk, vv in v.items():
            if torch.is_tensor(vv):
                vv.copy_(opt_state[k_][kk])
    el = [float(train_step(model, flat, sync, opt, x, y, batch, ptr)) for _ in range(3)]
    torch.cuda.synchronize(dev)
    out["rccl_graphed"] = {"losses": gl, "ms": g_ms}
    out["rccl_graphed_vs_eager_losses"] = el
    out["graphed_bitwise_equal"] = bool(torch.equal(g_params, flat.flat_param.detach().cpu()))
    dist.destroy_process_group()
    print("RCCL_CHILD " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
