import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deepmetv2_amd import _native
B, n, H = 64, 4500, 32
dev = torch.device("cuda:0"); torch.manual_seed(0)
N = B * n
ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
x = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev)
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
print("row-major gQ:", round(timeit(lambda: _native.gather_max_bwd_lds(g, arg, nbr, ptr, nbr_local=loc)), 1), "us",
      " slice-major gQ:", round(timeit(lambda: _native.gather_max_bwd_lds(g, arg, nbr, ptr, nbr_local=loc, sliced=True)), 1), "us")
