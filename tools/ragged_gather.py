import os, sys
sys.path.insert(0, "/root/repo")
import torch
from deepmetv2_amd import _native, synth
dev = torch.device("cuda:0"); H, k = 32, 16
def med(fn, reps=21):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for label, sizes in (("U[500,5000]", synth.ragged_sizes(64, 500, 5000, seed=7)), ("U[500,5100]x96", synth.ragged_sizes(96, 500, 5100, seed=3)),
                     ("64x4500", [4500] * 64), ("128x2000", [2000] * 128), ("U[1000,2500]x128", synth.ragged_sizes(128, 1000, 2500, seed=5))):
    N = sum(sizes); g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=g).to(dev); W = (torch.randn(H, 2 * H, generator=g) / 8).to(dev); b = torch.randn(H, generator=g).to(dev)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.tensor(sizes).cumsum(0)]).to(dev)
    nbr, _d, loc = _native.knn_local(x, ptr, k)
    P, Q = _native.node_linear_split(x, W, b, sliced=True); Pr, Qr = _native.node_linear_split(x, W, b, sliced=False)
    t_l2 = med(lambda: _native.gather_max(Pr, Qr, nbr, ptr, want_arg=True, lds=False))
    res = [f"L2 {t_l2:6.1f}"]
    for bal in ("0", "1"):
        os.environ["DMET_GATHER_BALANCED"] = bal
        t = med(lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=True, nbr_local=loc, sliced=True))
        res.append(f"LDS bal={bal} {t:6.1f}")
    del os.environ["DMET_GATHER_BALANCED"]
    t = med(lambda: _native.gather_max(P, Q, nbr, ptr, want_arg=True, lds=True, nbr_local=loc, sliced=True))
    res.append(f"LDS auto {t:6.1f}")
    print(f"{label:20s} N={N:7d} " + "  ".join(res) + f"   (roofline 0.40 = {N*352/0.4/8e12*1e6:5.1f} us)", flush=True)
