"""Randomised cross-check of the two kNN paths (matrix-core filter + certified re-rank vs the exact VALU kernel):
every output bit must agree on ragged batches, tiny / empty events, duplicated rows, clustered and heavy-tailed data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
D = int(sys.argv[2]) if len(sys.argv) > 2 else 32      # 32, or 64 (the DRN width: second filter form only)
g = torch.Generator().manual_seed(2024)
bad = 0
for it in range(rounds):
    B = int(torch.randint(1, 40, (1,), generator=g))
    kind = it % 5
    hi = [3000, 200, 6000, 40, 1200][kind]
    sizes = [int(v) for v in torch.randint(0, hi, (B,), generator=g)]
    N = sum(sizes)
    if N == 0:
        continue
    k = int([16, 8, 20, 13, 1][it % 5])
    x = torch.randn(N, D, generator=g)
    mode = it % 4
    if mode == 1:      # clusters + exact duplicates
        c = torch.randn(7, D, generator=g) * 3
        x = c[torch.randint(0, 7, (N,), generator=g)] + 1e-2 * torch.randn(N, D, generator=g)
        x[N // 2:N // 2 + N // 10] = x[:N // 10]
    elif mode == 2:    # heavy tails
        x = x * torch.exp(2.0 * torch.randn(N, 1, generator=g))
    elif mode == 3:    # low-dimensional manifold (many near ties)
        x = torch.randn(N, 2, generator=g) @ torch.randn(2, D, generator=g)
    hostile = (it // 4) % 3     # every third block of four rounds: far rows / non-finite rows on top of the mode
    if hostile == 1 and N >= 8:     # rows 1e5..1e6 from the origin: distances >= 1e10 must come back as -1
        idx = torch.randperm(N, generator=g)[: max(2, N // 40)]
        x[idx] = x[idx] * torch.empty(idx.numel(), 1).uniform_(1e5, 1e6, generator=g)
        x[idx[:2]] = x[idx[0]].clone()
    elif hostile == 2 and N >= 8:   # NaN / +-inf entries
        idx = torch.randperm(N, generator=g)[: max(3, N // 60)]
        x[idx[0::3], it % D] = float("nan")
        x[idx[1::3], (it + 1) % D] = float("inf")
        x[idx[2::3]] = float("-inf")
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
    xd = x.to(dev)
    os.environ["DMET_KNN_PATH"] = "exact"
    n0, d0 = _native.knn(xd, ptr, k)
    os.environ.pop("DMET_KNN_PATH")
    st = {}
    n1, d1 = _native.knn(xd, ptr, k, stats=st)
    ok = torch.equal(n0, n1) and torch.equal(d0, d1)
    bad += 0 if ok else 1
    print(f"round {it:2d}: B={B:2d} N={N:6d} k={k:2d} mode={mode} hostile={hostile} flagged_queries={st['flagged_queries']:6d} "
          f"{'ok' if ok else 'MISMATCH rows=' + str(int((n0 != n1).any(1).sum()))}")
print("mismatching rounds:", bad)
sys.exit(1 if bad else 0)
