"""Micro-benchmark of gather_max_bwd_lds_kernel: kNN table (int32 / uint16 ids), 255-wide radius-like table, and (experiment
build, k == 1) the same scatter without any id fetch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
B, n, H = 64, 4500, 32
dev = torch.device("cuda:0"); torch.manual_seed(0)
N = B * n
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
x = torch.randn(N, H, device=dev)
g = torch.randn(N, H, device=dev)
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
nbr, _, loc = _native.knn_local(x, ptr, 16)
arg = torch.randint(0, 16, (N, H), device=dev, dtype=torch.uint8)
print("k=16 int32 ids :", round(timeit(lambda: _native.gather_max_bwd_lds(g, arg, nbr, ptr)), 1), "us")
print("k=16 uint16 ids:", round(timeit(lambda: _native.gather_max_bwd_lds(g, arg, nbr, ptr, nbr_local=loc)), 1), "us")
wide = torch.zeros(N, 255, dtype=torch.int32, device=dev)
wide[:, :16] = nbr
arg36 = torch.randint(0, 16, (N, H), device=dev, dtype=torch.uint8)
print("255-wide rows  :", round(timeit(lambda: _native.gather_max_bwd_lds(g, arg36, wide, ptr)), 1), "us")
self1 = torch.arange(N, dtype=torch.int32, device=dev).view(-1, 1).contiguous()
arg0 = torch.zeros(N, H, dtype=torch.uint8, device=dev)
print("k=1 (no id fetch in the experiment build; one 4-byte id per row otherwise):",
      round(timeit(lambda: _native.gather_max_bwd_lds(g, arg0, self1, ptr)), 1), "us")
