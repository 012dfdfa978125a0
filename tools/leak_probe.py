"""Device memory held after N training steps of each flow with the garbage collector OFF: anything that needs the cycle
collector to be released shows up as growth (found this way: a table <-> edge_index reference cycle through the graph
registry, 0.6 GB per step of the radius_graph flow)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmetv2_amd as dm
from deepmetv2_amd import synth
from deepmetv2_amd.model import Net
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
x, y, batch, ptr = synth.make_events([4500] * 64, seed=0, device=dev)
dm.register_batch(batch, ptr, 64, max_nodes=4500)
etaphi = torch.stack([x[:, 3], torch.atan2(x[:, 1], x[:, 0])], 1).contiguous()
bad = False
for flow in ("dynamic", "static", "static-table", "knn_graph edge_index"):
    torch.manual_seed(0)
    model = Net(8, 3, graph="dynamic" if flow == "dynamic" else "static", k=16).to(dev).train()
    flat = FlatModule(model); sync = GradSync(flat)
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)

    def graph():
        if flow == "dynamic":
            return None
        if flow == "static":
            return dm.radius_graph(etaphi, 0.4, batch, loop=True, max_num_neighbors=255)
        if flow == "static-table":
            return dm.radius_table(etaphi, 0.4, batch, loop=True, max_num_neighbors=255)
        return dm.knn_graph(x[:, :8].contiguous(), 16, batch, loop=True)

    for _ in range(3):
        train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=graph())
    torch.cuda.synchronize(); gc.collect(); gc.disable()
    m0 = torch.cuda.memory_allocated(dev)
    for _ in range(40):
        train_step(model, flat, sync, opt, x, y, batch, ptr, edge_index=graph())
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated(dev)
    gc.enable()
    grew = (m1 - m0) / 2**20
    print(f"{flow:22s} allocated after 40 steps without gc: {m1 / 2**20:9.1f} MiB ({grew:+.1f} MiB)")
    bad = bad or grew > 64
    del model, flat, sync, opt
print("LEAK" if bad else "ok")
sys.exit(1 if bad else 0)
