"""Host-side (Python launch thread) time per training step vs GPU time: the step is GPU-bound while host < GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
flow = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
B, n = 64, 4500
x, y, batch, ptr = synth.make_events([n] * B, seed=0, device=dev)
dm.register_batch(batch, ptr, B, max_nodes=n)
torch.manual_seed(0)
model = Net(8, 3, graph="dynamic" if flow == "dynamic" else "static", k=16).to(dev).train()
flat = FlatModule(model); sync = GradSync(flat)
opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)


def graph():
    if flow == "dynamic":
        return None
    phi = torch.atan2(x[:, 1], x[:, 0])
    return dm.radius_table(torch.cat([x[:, 3][:, None], phi[:, None]], 1), r=0.4, batch=batch, loop=True, max_num_neighbors=255)


def step():
    return train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=graph())


for _ in range(10):
    step()
torch.cuda.synchronize()
import gc; gc.disable()
# host time: enqueue 30 steps while the GPU is kept far behind by a long sleep kernel? simpler: time the enqueue alone
t0 = time.perf_counter()
for _ in range(30):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{flow}: enqueue {1e3 * (t1 - t0) / 30:.3f} ms/step (lower bound of the host time when it exceeds the GPU's), "
      f"incl. drain {1e3 * (t2 - t0) / 30:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28); st.sort_stats("cumtime").print_stats(30)
