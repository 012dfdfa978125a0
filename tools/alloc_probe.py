"""Does the static flow call the device allocator every step?  (caching-allocator statistics across training steps)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
B, n = 64, 4500
x, y, batch, ptr = synth.make_events([n] * B, seed=0, device=dev)
dm.register_batch(batch, ptr, B, max_nodes=n)
torch.manual_seed(0)
model = Net(8, 3, graph="static").to(dev).train()
flat = FlatModule(model); sync = GradSync(flat)
opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)


def graph():
    phi = torch.atan2(x[:, 1], x[:, 0])
    return dm.radius_table(torch.cat([x[:, 3][:, None], phi[:, None]], 1), r=0.4, batch=batch, loop=True, max_num_neighbors=255)


def step():
    return train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=graph())


for _ in range(5):
    step()
torch.cuda.synchronize()
s0 = torch.cuda.memory_stats(dev)
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
t1 = time.perf_counter()
s1 = torch.cuda.memory_stats(dev)
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated"):
    print(k, s1.get(k, 0) - s0.get(k, 0))
print("reserved MiB", torch.cuda.memory_reserved(dev) / 2**20, "ms/step", (t1 - t0) / 20 * 1e3)
