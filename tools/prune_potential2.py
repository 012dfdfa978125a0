"""Round 3: how much of the kNN sweep could a locality order of an event's nodes remove?

For each embedding the model feeds the kNN build (layer 1 / layer 2, before and after some training steps) and for a few
orders of an event's nodes (queries and candidates both taken in that order: a wavefront = 64 consecutive queries, a tile =
32 consecutive candidates) it reports
  * the fraction of (wavefront, tile) pairs a ball bound could skip for ALL 64 queries given each query's true d_k
    (upper bound of what pruning can do), and
  * the fraction skippable given only the SEEDED bound of a query (k-th distance to its previous-layer neighbours).
Orders: index (today), pc1 (first principal component), kd (recursive median split on the widest coordinate down to
32-node leaves), kmeans (Lloyd, 128 centres, nodes sorted by centre, centres by pc1).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deepmetv2_amd import _native, synth
from deepmetv2_amd.model import Net, split_features
from deepmetv2_amd.parallel import FlatModule, GradSync, train_step

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
torch.manual_seed(0)
B, n, k = 8, 4500, 16
x, y, batch, ptr = synth.make_events([n] * B, seed=1234, device=dev)
model = Net(8, 3, graph="dynamic", k=k).to(dev).train()
if steps:
    flat = FlatModule(model)
    sync = GradSync(flat)
    opt = torch.optim.AdamW([flat.flat_param], lr=1e-3, fused=True)
    for _ in range(steps):
        train_step(model, flat, sync, opt, x, y, batch, ptr)
g = model.graphnet
xc, xk = split_features(x)
with torch.no_grad():
    emb = g.embed(xc, xk).contiguous()
    conv, norm = g.conv_continuous[0]
    emb2 = (emb + norm(conv(emb, batch))).contiguous()


def kd_order(ev):
    n = ev.shape[0]
    perm = torch.arange(n, device=ev.device)
    segs = [(0, n)]
    while segs:
        nxt = []
        for lo, hi in segs:
            if hi - lo <= 32:
                continue
            sub = ev[perm[lo:hi]]
            dim = int((sub.max(0).values - sub.min(0).values).argmax())
            o = torch.argsort(sub[:, dim])
            perm[lo:hi] = perm[lo:hi][o]
            mid = lo + (((hi - lo) // 2 + 31) // 32) * 32
            nxt += [(lo, mid), (mid, hi)]
        segs = nxt
    return perm


def kmeans_order(ev, K=128, iters=6):
    n = ev.shape[0]
    cen = ev[torch.randperm(n, device=ev.device)[:K]].clone()
    for _ in range(iters):
        a = torch.cdist(ev, cen).argmin(1)
        for c in range(K):
            m = a == c
            if m.any():
                cen[c] = ev[m].mean(0)
    a = torch.cdist(ev, cen).argmin(1)
    u, s, v = torch.pca_lowrank(cen, q=1)
    crank = torch.argsort(torch.argsort(cen @ v[:, 0]))
    return torch.argsort(crank[a] * n + torch.arange(n, device=ev.device))


def report(name, e, seed_tau=None):
    _, dist = _native.knn(e, ptr, k)
    for order_name in ("index", "pc1", "kd", "kmeans"):
        sk = sk_seed = total = 0
        near = 0
        for b in range(B):
            ev = e[b * n:(b + 1) * n]
            dk = dist[b * n:(b + 1) * n, k - 1]
            if order_name == "pc1":
                u, s, v = torch.pca_lowrank(ev, q=1)
                perm = torch.argsort(ev @ v[:, 0])
            elif order_name == "kd":
                perm = kd_order(ev)
            elif order_name == "kmeans":
                perm = kmeans_order(ev)
            else:
                perm = torch.arange(n, device=dev)
            evp, dkp = ev[perm], dk[perm]
            nt = n // 32
            tiles = evp[:nt * 32].view(nt, 32, 32)
            cen = tiles.mean(1)
            rad = (tiles - cen[:, None]).norm(dim=2).max(1).values
            dq = torch.cdist(evp, cen)
            lb = (dq - rad[None]).clamp(min=0) ** 2
            nw = n // 64
            can = lb > dkp[:, None]
            sk += int(can[:nw * 64].view(nw, 64, nt).all(1).sum())
            total += nw * nt
            if seed_tau is not None:
                sp = seed_tau[b * n:(b + 1) * n][perm]
                can2 = lb > sp[:, None]
                sk_seed += int(can2[:nw * 64].view(nw, 64, nt).all(1).sum())
            # tiles that hold at least one true neighbour of at least one query of the wavefront
            dd = torch.cdist(evp, evp) ** 2
            hit = (dd <= dkp[:, None] * (1 + 1e-6))[:nw * 64, :nt * 32].view(nw, 64, nt, 32).any(3).any(1)
            near += int(hit.sum())
        msg = f"{name:14s} order={order_name:7s}: skippable by ball bound with true d_k {100.0 * sk / total:5.1f} %"
        if seed_tau is not None:
            msg += f", with the seeded bound {100.0 * sk_seed / total:5.1f} %"
        msg += f"; (wave, tile) pairs that hold a true neighbour {100.0 * near / total:5.1f} %"
        print(msg, flush=True)


with torch.no_grad():
    nbr1, _ = _native.knn(emb, ptr, k)
    diff = emb2[nbr1.long()] - emb2[:, None, :]
    tau2 = (diff * diff).sum(-1).max(1).values * (1 + 1e-5)
    nbr2, d2 = _native.knn(emb2, ptr, k)
    cnts = []
    for b in range(B):
        e = emb2[b * n:(b + 1) * n]
        dd = torch.cdist(e, e) ** 2
        cnts.append((dd < tau2[b * n:(b + 1) * n, None]).sum(1))
    cnt = torch.cat(cnts).float()
    print(f"steps {steps}: candidates below the seeded threshold (layer 2 from layer-1 graph): mean {float(cnt.mean()):.1f} "
          f"median {float(cnt.median()):.0f} p90 {float(cnt.quantile(0.9)):.0f} p99 {float(cnt.quantile(0.99)):.0f} "
          f"max {float(cnt.max()):.0f}; seeded / true d_k ratio median {float((tau2 / d2[:, k - 1]).median()):.2f}", flush=True)
    report(f"s{steps} layer1", emb)
    report(f"s{steps} layer2", emb2, tau2)
    if steps == 0:
        report("gaussian", torch.randn_like(emb))
