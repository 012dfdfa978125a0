import sys; sys.path.insert(0, "/root/repo")
import torch
import deepmetv2_amd as dm
from deepmetv2_amd.model import Net, loss_fn, split_features
dev = torch.device("cuda:0")
ok = True
def check(name, fn):
    global ok
    try:
        r = fn(); torch.cuda.synchronize(); print("ok  ", name, "->", r)
    except Exception as e:
        ok = False; print("FAIL", name, type(e).__name__, str(e)[:200])
x0 = torch.zeros(0, 32, device=dev); b0 = torch.zeros(0, dtype=torch.int64, device=dev)
check("knn_graph empty", lambda: tuple(dm.knn_graph(x0, 4, b0).shape))
check("radius_graph empty", lambda: tuple(dm.radius_graph(torch.zeros(0, 2, device=dev), 0.4, b0).shape))
x1 = torch.randn(1, 32, device=dev); b1 = torch.zeros(1, dtype=torch.int64, device=dev)
check("knn_graph single node loop=True", lambda: dm.knn_graph(x1, 4, b1, loop=True).tolist())
check("knn_graph single node loop=False", lambda: dm.knn_graph(x1, 4, b1, loop=False).tolist())
lin = torch.nn.Sequential(torch.nn.Linear(64, 32)).to(dev)
conv = dm.DynamicEdgeConv(nn=lin, k=16).to(dev)
check("DynamicEdgeConv single node", lambda: tuple(conv(x1, b1).shape))
check("DynamicEdgeConv empty", lambda: tuple(conv(x0, b0).shape))
xs = torch.randn(5, 32, device=dev, requires_grad=True); bs = torch.tensor([0, 0, 2, 2, 2], device=dev)   # event 1 empty
def f():
    out = conv(xs, bs); out.sum().backward(); return tuple(out.shape), bool(torch.isfinite(xs.grad).all())
check("DynamicEdgeConv with an empty event in the middle + backward", f)
from deepmetv2_amd import synth
x, y, batch, ptr = synth.make_events([1], seed=1, device=dev)
model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
def g():
    w = model(*split_features(x), None, batch); l = loss_fn(w, x, y, batch); l.backward(); return float(l)
check("Net train step on one single-node event", g)
x, y, batch, ptr = synth.make_events([3, 0, 2], seed=2, device=dev)
def h():
    model.zero_grad(); w = model(*split_features(x), None, batch); l = loss_fn(w, x, y, batch); l.backward(); return float(l)
check("Net train step with an empty event", h)
et = torch.rand(7, 2, device=dev); bt = torch.tensor([0, 0, 0, 1, 1, 1, 1], device=dev)
model_s = Net(8, 3, graph="static").to(dev).train()
x7, y7, batch7, _ = synth.make_events([3, 4], seed=3, device=dev)
def s():
    ei = dm.radius_graph(et, 0.4, batch7, loop=True, max_num_neighbors=255)
    w = model_s(*split_features(x7), ei, batch7); l = loss_fn(w, x7, y7, batch7); l.backward(); return tuple(ei.shape), float(l)
check("static flow tiny", s)
print("ALL OK" if ok else "SOME FAILED")
