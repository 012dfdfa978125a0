"""Where does the launch thread spend its time?  cProfile over N eager training steps of the benchmark (the GPU is not
waited for inside the profiled region), top functions by own time and by cumulative time.
Usage: python tools/host_profile.py [steps] [fused|stock-knn-graph|stock-dynamic][+layers|+fuse]"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import deepmetv2_amd as dm
from deepmetv2_amd import stock_model, synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.optim import FlatAdamW
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
which = sys.argv[2] if len(sys.argv) > 2 else "fused"
dev = torch.device("cuda:0")
B, n = 64, 4500
x, y, batch, ptr = synth.make_events([n] * B, seed=1234, device=dev)
dm.register_batch(batch, ptr, B, max_nodes=n, min_nodes=n)
torch.manual_seed(0)
if which == "fused":
    model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
    flat = FlatModule(model); sync = GradSync(flat)
    opt = FlatAdamW([flat.flat_param], lr=1e-3)

    def step():
        return train_step(model, flat, sync, opt, x, y, batch, ptr)
else:
    acc = None
    if "+" in which:
        which, acc = which.split("+")
    model = stock_model.StockNet(dm, 8, 3, variant=which[len("stock-"):].replace("-", "_"), k=16).to(dev).train()
    if acc == "layers":
        model = dm.accelerate(model, fuse=False)
    elif acc == "fuse":
        model = dm.accelerate(model, graph="dynamic", k=16)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)

    def step():
        return stock_model.stock_train_step(dm, model, opt, x, y, batch)

for _ in range(30):
    step()
torch.cuda.synchronize()
import gc
gc.disable()
import time
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter(); step(); ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
ts.sort()
print(f"{which}: host enqueue time per step, GPU idle at the start: median {ts[10] * 1e3:.3f} ms, min {ts[0] * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step()
    if i % 10 == 9:
        torch.cuda.synchronize()      # keep the launch queue short: time spent blocked in a full queue is not host work
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(28)
    print(buf.getvalue()[:6000])
