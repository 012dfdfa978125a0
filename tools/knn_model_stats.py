"""Flag statistics / timing of the kNN paths on the embeddings the model actually feeds them (layer 1 and 2).
Usage: python tools/knn_model_stats.py [train_steps]   (train_steps AdamW steps of the benchmark first: the bench measures
its kNN on the embeddings the run ended with)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
from deepmetv2_amd.model import Net, split_features
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(0)
B, n = 64, 4500
x, y, batch, ptr = synth.make_events([n] * B, seed=1234, device=dev)
model = Net(8, 3, graph="dynamic", k=16).to(dev).train()
if steps:
    flat = FlatModule(model)
    sync = GradSync(flat)
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)
    for _ in range(steps):
        train_step(model, flat, sync, opt, x, y, batch, ptr)
g = model.graphnet
xc, xk = split_features(x)
with torch.no_grad():
    emb = g.embed(xc, xk)
    embs = [emb]
    conv, norm = g.conv_continuous[0]
    emb2 = emb + norm(conv(emb, batch))
    embs.append(emb2)
for li, e in enumerate(embs):
    e = e.contiguous()
    st = {}
    nbr, dist = _native.knn(e, ptr, 16, stats=st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.knn(e, ptr, 16); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    nrm = (e * e).sum(1)
    print(f"steps {steps} layer {li}: path={os.environ.get('DMET_KNN_PATH','filter')} stats={st} median {sorted(ts)[3]:.3f} ms; |x|^2 mean {float(nrm.mean()):.2f} max {float(nrm.max()):.2f}; "
          f"|x|max {float(e.abs().max()):.1f}; d16 mean {float(dist[:, 15].mean()):.4f} d2 mean {float(dist[:, 1].mean()):.5f} dup-ish (d2<1e-6): {int((dist[:,1] < 1e-6).sum())}")
