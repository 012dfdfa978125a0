"""Micro-benchmark of the fused EdgeConv pieces on BASELINE configs[1] shapes (real kNN table)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native, _lib
B, n, H, k = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 4500, 32, 16
dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(B * n, H, device=dev)
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
nbr, _, loc = _native.knn_local(x, ptr, k)
W = torch.randn(H, 2 * H, device=dev) * 0.1; b = torch.randn(H, device=dev)
P, Q = _native.node_linear_split(x, W, b)
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
res = {}
for form in ("l2-only", "lds"):
    _native.GATHER_MAX_FORM = form
    for arg in (False, True):
        us = timeit(lambda: _native.gather_max(P, Q, nbr, ptr, arg))
        byts = B * n * (128 + 64 + 128 + (32 if arg else 0))
        print(f"gather_max[{form}] arg={arg}: {us:7.2f} us  -> {byts/us/1e3:7.1f} GB/s algorithmic = {byts/us/1e3/8000*100:5.1f}% of 8 TB/s")
    res[form] = _native.gather_max(P, Q, nbr, ptr, True)
for arg in (False, True):
    us = timeit(lambda: _native.gather_max(P, Q, nbr, ptr, arg, lds=True, nbr_local=loc))
    byts = B * n * (128 + 64 + 128 + (32 if arg else 0))
    print(f"gather_max[lds, uint16 local ids] arg={arg}: {us:7.2f} us  -> {byts/us/1e3:7.1f} GB/s algorithmic = {byts/us/1e3/8000*100:5.1f}% of 8 TB/s")
r16 = _native.gather_max(P, Q, nbr, ptr, True, lds=True, nbr_local=loc)
print("uint16 ids agree:", torch.equal(r16[0], res["lds"][0]), torch.equal(r16[1], res["lds"][1]))
print("forms agree:", torch.equal(res["l2-only"][0], res["lds"][0]), torch.equal(res["l2-only"][1], res["lds"][1]))
for arg in (False, True):
    us = timeit(lambda: _native.edgeconv_fused_lds(x, W, b, nbr, ptr, arg))
    byts = B * n * (128 + 64 + 128 + (32 if arg else 0))
    print(f"edgeconv_fused_lds arg={arg}: {us:7.2f} us -> {byts/us/1e3:7.1f} GB/s algorithmic = {byts/us/1e3/8000*100:5.1f}% of 8 TB/s")
ref = _native.gather_max(P, Q, nbr, ptr, True)
fo = _native.edgeconv_fused_lds(x, W, b, nbr, ptr, True)
print("fused vs split max|d|:", float((fo[0]-ref[0]).abs().max()), "arg mismatches:", int((fo[1]!=ref[1]).sum()))
print(f"node_linear_split: {timeit(lambda: _native.node_linear_split(x, W, b)):.2f} us")
