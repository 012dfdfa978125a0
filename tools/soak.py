"""Soak: a few hundred training steps on fresh synthetic batches (new events every step, ragged sizes): finite loss,
steady memory, no kNN fallback storms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
flat = FlatModule(model); sync = GradSync(flat)
opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)
t0 = time.perf_counter()
peak0 = None
for it in range(steps):
    sizes = synth.ragged_sizes(32, 500, 6000, seed=it)
    x, y, batch, ptr = synth.make_events(sizes, seed=1000 + it, device=dev)
    dm.register_batch(batch, ptr, len(sizes), max_nodes=max(sizes))
    loss = train_step(model, flat, sync, opt, x, y, batch, ptr)
    if it % 50 == 0 or it == steps - 1:
        torch.cuda.synchronize()
        mem = torch.cuda.max_memory_allocated() / 2**20
        if it == 50:
            peak0 = mem
        print(f"step {it:4d} loss {float(loss):12.4f} finite={bool(torch.isfinite(loss))} peak_mem {mem:8.1f} MiB "
              f"elapsed {time.perf_counter() - t0:6.1f} s", flush=True)
        assert bool(torch.isfinite(loss))
print("ok")
