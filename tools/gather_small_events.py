"""The gather on batches of SMALL events (what real data looks like): L2 form against the LDS form, us per launch, standalone,
warm.  (Round 3, second session: an 80 KB image variant of the LDS kernel -- two workgroups per CU, chosen by a max_nodes
hint -- was built, bit-identical, and timed with this script: no faster than the 160 KB image at any of these shapes; removed.
Run against that build the third column was its time.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, synth
dev = torch.device("cuda:0"); H, k = 32, 16
def med(fn, reps=21):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for label, sizes in (("128 x 1000", [1000] * 128), ("128 x 2000", [2000] * 128), ("256 x 1000", [1000] * 256), ("110 x 2500", [2500] * 110),
                     ("128 x U[1000,2500]", synth.ragged_sizes(128, 1000, 2500, seed=5)), ("256 x U[500,2000]", synth.ragged_sizes(256, 500, 2000, seed=6))):
    N = sum(sizes); g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=g).to(dev); W = (torch.randn(H, 2 * H, generator=g) / 8).to(dev); b = torch.randn(H, generator=g).to(dev)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
    nbr, _d, loc = _native.knn_local(x, ptr, k)
    P, Q = _native.node_linear_split(x, W, b, sliced=True); Pr, Qr = _native.node_linear_split(x, W, b, sliced=False)
    t_l2 = med(lambda: _native.gather_max(Pr, Qr, nbr, ptr, want_arg=True, lds=False))
    t_full = med(lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=True, nbr_local=loc, sliced=True))
    t_half = med(lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=True, nbr_local=loc, sliced=True, max_nodes=max(sizes)))
    print(f"{label:22s} N={N:7d}  L2 form {t_l2:6.1f}   LDS form (160 KB image, 1024 threads) {t_full:6.1f}   half form (80 KB, 512 threads) {t_half:6.1f}", flush=True)
