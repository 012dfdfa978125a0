"""Upper bound on tile pruning for the kNN sweep: with nodes of an event sorted by a locality key, what fraction of
(64-query wave, 32-candidate tile) pairs could be skipped by the ball bound (|x_q - c_t| - r_t)^2 > d_k(q) for all 64 queries?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
from deepmetv2_amd.model import Net, split_features

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, n, k = 4, 4500, 16
x, y, batch, ptr = synth.make_events([n] * B, seed=1)
x, batch, ptr = x.to(dev), batch.to(dev), ptr.to(dev)
model = Net(8, 3, graph="dynamic", k=k).to(dev).train()
g = model.graphnet
xc, xk = split_features(x)
with torch.no_grad():
    emb = g.embed(xc, xk)
    conv, norm = g.conv_continuous[0]
    emb2 = (emb + norm(conv(emb, batch))).contiguous()
for name, e in (("layer1", emb.contiguous()), ("layer2", emb2), ("gaussian", torch.randn_like(emb))):
    _, dist = _native.knn(e, ptr, k)
    for order_name in ("index", "category", "pc1"):
        skipped = total = 0
        for b in range(B):
            ev = e[b * n:(b + 1) * n]
            dk = dist[b * n:(b + 1) * n, k - 1]
            if order_name == "category":
                key = xk[b * n:(b + 1) * n, 0].abs() * 100 + (xk[b * n:(b + 1) * n, 1] + 1) * 10 + xk[b * n:(b + 1) * n, 2]
                perm = torch.argsort(key, stable=True)
            elif order_name == "pc1":
                u, s, v = torch.pca_lowrank(ev, q=1)
                perm = torch.argsort(ev @ v[:, 0])
            else:
                perm = torch.arange(n, device=dev)
            evp, dkp = ev[perm], dk[perm]
            nt = n // 32
            tiles = evp[:nt * 32].view(nt, 32, 32)
            cen = tiles.mean(1)
            rad = (tiles - cen[:, None]).norm(dim=2).max(1).values
            dq = torch.cdist(evp, cen)                       # [n, nt]
            lb = (dq - rad[None]).clamp(min=0) ** 2
            can_skip = lb > dkp[:, None]                     # per query
            nw = n // 64
            w = can_skip[:nw * 64].view(nw, 64, nt).all(1)   # all 64 queries of the wave agree
            skipped += int(w.sum()); total += w.numel()
        print(f"{name:9s} order={order_name:8s}: skippable (wave, tile) pairs {100.0 * skipped / total:5.1f} %")
