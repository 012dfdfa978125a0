"""How tight is 'k-th layer-2 distance within the layer-1 neighbour set' as an a-priori kNN threshold?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
from deepmetv2_amd.model import Net, split_features

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, n, k = 8, 4500, 16
x, y, batch, ptr = synth.make_events([n] * B, seed=1)
x, batch, ptr = x.to(dev), batch.to(dev), ptr.to(dev)
model = Net(8, 3, graph="dynamic", k=k).to(dev).train()
g = model.graphnet
with torch.no_grad():
    emb = g.embed(*split_features(x))
    conv, norm = g.conv_continuous[0]
    emb2 = (emb + norm(conv(emb, batch))).contiguous()
    nbr1, _ = _native.knn(emb.contiguous(), ptr, k)
    nbr2, d2 = _native.knn(emb2, ptr, k)
    # layer-2 distances to the layer-1 neighbours
    diff = emb2[nbr1.long()] - emb2[:, None, :]
    dseed = (diff * diff).sum(-1)                      # [N,k]
    tau = dseed.max(1).values                          # k-th (largest) distance within the seed set
    overlap = (nbr1[:, :, None] == nbr2[:, None, :]).any(-1).float().mean()
    cnts = []
    for b in range(B):
        e = emb2[b * n:(b + 1) * n]
        dd = torch.cdist(e, e) ** 2
        cnts.append((dd < tau[b * n:(b + 1) * n, None] * (1 + 1e-5)).sum(1))
    cnt = torch.cat(cnts).float()
print(f"overlap of layer-2 neighbours with the layer-1 set: {float(overlap):.3f}")
print(f"candidates below the seeded threshold: mean {float(cnt.mean()):.1f} median {float(cnt.median()):.0f} "
      f"p90 {float(cnt.quantile(0.9)):.0f} p99 {float(cnt.quantile(0.99)):.0f} max {float(cnt.max()):.0f}; "
      f"frac > 44: {float((cnt > 44).float().mean()):.4f}")
