"""Experiment: one hipGraph per training step (torch.cuda.CUDAGraph) vs eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, k = 4500, 16
x, y, batch, ptr = synth.make_events([n] * B, seed=1234, device=dev)
dm.register_batch(batch, ptr, B, max_nodes=n)
torch.manual_seed(0)
model = Net(8, 3, graph="dynamic", k=k).to(dev).train()
flat = FlatModule(model); sync = GradSync(flat)
opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, capturable=True)

def step():
    return train_step(model, flat, sync, opt, x, y, batch, ptr)

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step")

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step  loss={float(loss):.4f}")
