"""kNN build time against the number of events (4500 nodes, 32 features, k = 16): the filter's items are 64-query tiles
on 2048 wavefront slots, so B = 57 is two full rounds (4047 tiles) and B = 64 (4544) adds a tail of 448 split tiles --
what does that tail cost beyond its share of the work?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmetv2_amd import _native
n, D, k = 4500, 32, 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
Bs = [int(v) for v in sys.argv[1:]] or [28, 29, 32, 40, 48, 56, 57, 58, 60, 62, 64, 66, 72, 80, 86, 87, 96]
xall = torch.randn(max(Bs) * n, D, device=dev)
for B in Bs:
    x = xall[: B * n]
    ptr = torch.arange(0, (B + 1) * n, n, dtype=torch.int64, device=dev)
    for _ in range(3): _native.knn_local(x, ptr, k)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _native.knn_local(x, ptr, k); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    tiles = B * ((n + 63) // 64)
    print(f"B={B:3d} tiles={tiles:5d} rounds={tiles / 2048:5.2f}  median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us  "
          f"us/tile-round {ts[len(ts) // 2] / (tiles / 2048):6.1f}", flush=True)
