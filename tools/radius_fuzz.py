"""Randomised cross-check of the windowed radius graph against the all-pairs sweep: ids, order and counts must agree."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native

dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(77)
bad = 0
for it in range(rounds):
    B = int(torch.randint(1, 50, (1,), generator=g))
    hi = [3000, 100, 6000, 20, 800][it % 5]
    sizes = [int(v) for v in torch.randint(0, hi, (B,), generator=g)]
    N = sum(sizes)
    if N == 0:
        continue
    D = [2, 1, 3, 8, 2][it % 5]
    x = torch.randn(N, D, generator=g) * torch.tensor([3.0, 2.0, 1.0, 0.5, 0.5, 0.2, 0.2, 0.1][:D])
    mode = it % 4
    if mode == 1:      # dense blobs: rows overflow max_nbr
        c = torch.randn(5, D, generator=g) * 2
        x = c[torch.randint(0, 5, (N,), generator=g)] + 0.05 * torch.randn(N, D, generator=g)
    elif mode == 2:    # lattice: exact ties at the radius
        x = torch.round(x * 4) / 4
    elif mode == 3:    # constant first coordinate: the window keeps everything
        x[:, 0] = 0.25
    r = [0.4, 0.25, 1.0, 0.05][it % 4]
    mx = [255, 32, 7, 64][(it // 4) % 4]
    skip = bool(it % 2)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
    xd = x.to(dev)
    _native.RADIUS_FORM = "sweep"
    n0, c0 = _native.radius(xd, ptr, r, mx, skip_self=skip, pad=True)
    _native.RADIUS_FORM = "windowed"
    n1, c1 = _native.radius(xd, ptr, r, mx, skip_self=skip, pad=True)
    ok = torch.equal(n0, n1) and torch.equal(c0, c1)
    bad += 0 if ok else 1
    print(f"round {it:3d}: B={B:2d} N={N:6d} D={D} r={r} max={mx:3d} mode={mode} mean_cnt={float(c0.float().mean()):6.1f} {'ok' if ok else 'MISMATCH'}")
print("mismatching rounds:", bad)
sys.exit(1 if bad else 0)
