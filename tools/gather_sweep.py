"""B-sweep of the graded gather + max kernel and its backward scatter (standalone, warm caches, HIP events):
B x 4500-node events for B in {32, 48, 64, 65, 96, 128} and the 8 `balanced_shards` of a 512-event ragged batch
(500..8000 nodes), with the roofline fraction on the same algorithmic-byte definition as bench.py (352 B/node training
form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
from deepmetv2_amd.parallel import balanced_shards

dev = torch.device("cuda:0")
H, k = 32, 16


def med(fn, reps=15):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def run(label, sizes):
    N = sum(sizes)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=g).to(dev)
    W = (torch.randn(H, 2 * H, generator=g) / 8).to(dev)
    b = torch.randn(H, generator=g).to(dev)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
    nbr, _d, loc = _native.knn_local(x, ptr, k)
    fits = max(sizes) <= 5119
    P, Q = _native.node_linear_split(x, W, b, sliced=fits)
    fwd = lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=fits, nbr_local=loc, sliced=fits, mixed=not fits)
    if not fits:   # the alternative for batches with oversized events: L2 gathers for every event
        t_l2 = med(lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=False))
        print(f"{label:34s} N={N:7d} max={max(sizes):5d} fwd {t_l2 * 1e3:7.1f} us  frac {N * 352.0 / (t_l2 * 1e-3) / 8e12:5.3f}   [L2 gathers, whole batch]")
    out, arg = fwd()
    gout = torch.randn(N, H, device=dev)
    bwd = lambda: _native.gather_max_bwd_lds(gout, arg, nbr, ptr, nbr_local=loc)
    t_f, t_b = med(fwd), med(bwd)
    alg = N * 352.0
    print(f"{label:34s} N={N:7d} max={max(sizes):5d} fwd {t_f * 1e3:7.1f} us  frac {alg / (t_f * 1e-3) / 8e12:5.3f}   "
          f"bwd {t_b * 1e3:7.1f} us   [{_native.last_gather_kernel[:60]}]")


for B in (16, 32, 40, 48, 64, 65, 72, 96, 128):
    run(f"B={B} x 4500", [4500] * B)
run("ragged 64 x U[500,5000] (every event LDS-resident)", synth.ragged_sizes(64, 500, 5000, seed=7))
sizes = synth.ragged_sizes(512, 500, 8000, seed=1234)
for r, shard in enumerate(balanced_shards([s * s for s in sizes], 8)):
    run(f"ragged 512-event batch, shard {r} ({len(shard)} ev)", [sizes[i] for i in shard])
run("ragged 64 x U[500,8000] (configs[4])", synth.ragged_sizes(64, 500, 8000, seed=1234))
